#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6e
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -k "schnorr_msm or typed_boundary" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
( time timeout 400 libecc_amd/lib/compat_check benchj 20 ) > $O/benchj.json 2> $O/benchj.err
cat $O/benchj.json | cut -c1-300; tail -n 3 $O/benchj.err
ECAMD_COMPAT_TIMING=1 timeout 300 libecc_amd/lib/compat_check bench_schnorr 20 > $O/typed_schnorr.txt 2>&1
grep "^bench" $O/typed_schnorr.txt
grep "compat timing" $O/typed_schnorr.txt | tail -8
