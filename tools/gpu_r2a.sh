#!/bin/bash
# first GPU pass of round 2: micro-benchmark (instruction rates + sustained clock), GPU test suite at reduced full-size
# settings, bench line, A/B of the 64-bit-shift variant
set -x
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
libecc_amd/lib/ubench 2000 > gpurun_out/r2a/ubench.json 2> gpurun_out/r2a/ubench.err
rocm-smi --showclocks > gpurun_out/r2a/smi_idle.txt 2>&1
ECAMD_TEST_FULL_LOG2=17 ECAMD_TEST_PARITY_ITEMS=16384 ECAMD_TEST_REF_ITEMS=1024 timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --ubench-json gpurun_out/r2a/ubench_bench.json > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
ECAMD_LIB_PATH=$PWD/libecc_amd/lib/variants/libecc_amd_shiftpair.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --parity-items 4096 > gpurun_out/r2a/bench_shiftpair.json 2> gpurun_out/r2a/bench_shiftpair.err
tail -3 gpurun_out/r2a/pytest.log
