#!/bin/bash
# pipeline v2 of the secp256r1 path (coalesced staging, 64-byte table records): tests that touch it, bench, kernel trace
set -x
mkdir -p gpurun_out/r2d
cd $GRAFT_REPO_ROOT
ECAMD_TEST_FULL_LOG2=18 ECAMD_TEST_PARITY_ITEMS=16384 ECAMD_TEST_REF_ITEMS=1024 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py -m gpu -x -q --durations=8 -k "SECP256R1 or secp256r1 or device_pointer or linearity or exceptional or chunk or multi or golden or crafted or comb or pipeline or blind" > gpurun_out/r2d/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2d/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --parity-items 1024 > $GRAFT_REPO_ROOT/gpurun_out/r2d/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2d/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r2d/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/r2d/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" > gpurun_out/r2d/kernel_stats_head.csv
find gpurun_out/r2d/prof -name "*.db" -delete; find gpurun_out/r2d/prof -size +2M -delete
tail -4 gpurun_out/r2d/pytest.log; cat gpurun_out/r2d/bench.json | head -c 600
