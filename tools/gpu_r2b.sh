#!/bin/bash
# second GPU pass of round 2: the new format / blinding / secret-mode / neutral-R tests, secret-mode cost, clocks under load
set -x
mkdir -p gpurun_out/r2b
cd $GRAFT_REPO_ROOT
ECAMD_TEST_FULL_LOG2=17 ECAMD_TEST_PARITY_ITEMS=4096 ECAMD_TEST_REF_ITEMS=512 timeout 1500 python -m pytest tests/test_gpu_formats.py tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -x -q --durations=10 -k "formats or multi or blind or secret or eddsa or ed448 or edge_fixtures or boundary or sign" > gpurun_out/r2b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
# sustained clocks while the MAD stream runs (rocm-smi sampled from the side)
( libecc_amd/lib/ubench 30000 > gpurun_out/r2b/ubench_long.json 2>/dev/null & )
sleep 2
for k in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --json >> gpurun_out/r2b/smi_under_load.jsonl 2>/dev/null; echo >> gpurun_out/r2b/smi_under_load.jsonl; sleep 1.5; done
wait
sleep 20
timeout 600 python tools/bench_secret_mode.py > gpurun_out/r2b/secret_mode.json 2> gpurun_out/r2b/secret_mode.err
tail -3 gpurun_out/r2b/pytest.log
