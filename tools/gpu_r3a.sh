#!/bin/bash
# Round 3, GPU pass a: the libecc-typed boundary (new signing / key / X25519 forms, pipeline) against libecc's scalar
# functions, its end-to-end rates, the whole GPU suite, and the effective shader clock (GRBM_GUI_ACTIVE / wall) under the
# MAD stream of ubench and under the headline loop.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r3a.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
nproc > $O/host.txt; lscpu | grep -i "model name" >> $O/host.txt
( time timeout 900 libecc_amd/lib/compat_check 640 ) > $O/compat_check.txt 2>&1; echo "rc=$?" >> $O/compat_check.txt
( time ECAMD_COMPAT_CHUNK=2048 timeout 300 libecc_amd/lib/compat_check quick 6000 ) > $O/compat_quick_6000.txt 2>&1; echo "rc=$?" >> $O/compat_quick_6000.txt
timeout 300 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20.txt 2>&1
timeout 200 libecc_amd/lib/compat_check bench 18 > $O/compat_bench_18.txt 2>&1
ECAMD_COMPAT_THREADS=1 timeout 200 libecc_amd/lib/compat_check bench 18 > $O/compat_bench_18_1thread.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_ubench -- $R/libecc_amd/lib/ubench 2000 > $O/clk_ubench.json 2> $O/clk_ubench.err
db=$(ls -S $(find $O/clk_ubench -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py clock $db > $O/clk_ubench.md 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/clk_bench -- python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 6 --warmup 3 > $O/clk_bench.json 2> $O/clk_bench.err
db=$(ls -S $(find $O/clk_bench -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py clock $db > $O/clk_bench.md 2>&1
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 5 $O/compat_check.txt; cat $O/compat_bench_20.txt; tail -n 4 $O/pytest.log; head -20 $O/clk_ubench.md; head -12 $O/clk_bench.md
