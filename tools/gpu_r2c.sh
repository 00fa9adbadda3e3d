#!/bin/bash
set -x
mkdir -p gpurun_out/r2c
cd $GRAFT_REPO_ROOT
ECAMD_TEST_FULL_LOG2=17 ECAMD_TEST_PARITY_ITEMS=4096 ECAMD_TEST_REF_ITEMS=512 timeout 1500 python -m pytest tests/test_gpu_formats.py tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -q --durations=10 -k "formats or multi or blind or secret or eddsa or ed448 or edge_fixtures or boundary or sign" > gpurun_out/r2c/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
timeout 600 python tools/bench_secret_mode.py > gpurun_out/r2c/secret_mode.json 2> gpurun_out/r2c/secret_mode.err
tail -5 gpurun_out/r2c/pytest.log
