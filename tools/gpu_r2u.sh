#!/bin/bash
# round 2, GPU pass u: end-to-end rate of ec_verify_batch through libsign_amd.so (libecc structures in), and the compat tests
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2u
mkdir -p $O
cd $R
nproc > $O/nproc.txt
timeout 600 libecc_amd/lib/compat_check bench 18 > $O/compat_bench_18.txt 2>&1; cat $O/compat_bench_18.txt
timeout 600 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20.txt 2>&1; cat $O/compat_bench_20.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "typed_boundary or self_tests" 2>&1 | tail -n 5
