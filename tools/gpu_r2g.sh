#!/bin/bash
# round 2, GPU pass g: multi-scalar multiplication (tests + timing), secret mode on the masked secp256r1 loop, ubench lane-spread A/B
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm.py -x -q 2>&1 | tail -40 > $O/pytest_msm.log
tail -5 $O/pytest_msm.log
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_wycheproof.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_other.log
tail -5 $O/pytest_other.log
MSM_LOG2=${MSM_LOG2:-16,18,20} timeout 600 python tools/bench_msm.py > $O/bench_msm.json 2> $O/bench_msm.err
cat $O/bench_msm.json | head -60
tail -3 $O/bench_msm.err
timeout 300 python tools/bench_secret_mode.py > $O/secret_mode.json 2> $O/secret_mode.err
cat $O/secret_mode.json
timeout 300 $R/libecc_amd/lib/ubench > $O/ubench.json 2> $O/ubench.err
grep -A12 mul256_column_sums $O/ubench.json | head -30
