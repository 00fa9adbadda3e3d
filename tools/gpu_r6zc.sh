#!/bin/bash
# round 6: fold 8 as the default of the bucket reduction -- the whole-batch tests (with the fold variants), then the three workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zc
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "test_gpu_schnorr_msm or test_gpu_msm or test_gpu_ed448_msm or typed_boundary" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for w in bip0340_msm ed25519_msm ed448_msm; do
  timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 2> /dev/null | tail -1 | python -c "
import json, sys
j = json.loads(sys.stdin.read())
print('$w: %.3f ms, %.1f M/s' % (j.get('ms_per_step', 0), j.get('value', 0) / 1e6))"
done
for i in 1 2; do
timeout 600 libecc_amd/lib/compat_check benchv 20 bip0340 2> /dev/null | grep -o '"call": "ec_verify_batch BIP0340[^,]*", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*'
timeout 600 libecc_amd/lib/compat_check benchv 20 ed25519 2> /dev/null | grep -o '"call": "ec_verify_batch EDDSA25519", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*'
done
