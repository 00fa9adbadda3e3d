#!/bin/bash
# round 6: the Ed25519 whole-batch call writes encodings and marks straight into the batch-wide arrays -- tests, then the typed call
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zj
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -k "test_gpu_msm or whole_batch or typed_boundary" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for i in 1 2 3; do
timeout 600 libecc_amd/lib/compat_check benchv 20 ed25519 2> /dev/null | grep -o '"call": "ec_verify_batch EDDSA25519", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*'
done
