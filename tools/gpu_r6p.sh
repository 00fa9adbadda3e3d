#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6p
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -x -q -k "test_gpu_msm or whole_batch or typed_boundary or schnorr_msm" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
( time timeout 400 $R/libecc_amd/lib/compat_check benchj 20 ) > $O/benchj.json 2> $O/benchj.err
cat $O/benchj.json | cut -c1-200; tail -n 3 $O/benchj.err
