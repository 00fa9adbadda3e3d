#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats of the bench command on the final build (profiles/r6_final_bench_kernels.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_kernels
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-secondary --no-traffic --no-cpu-baseline --parity-items 256 --steps 10 --warmup 3 > $O/prof_bench.json 2> $O/prof_bench.err
db=$(ls -S $(find $O/prof -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py dispatches $db k_p256_loop > $O/bench_loop_dispatches.md && python $R/tools/rocpd_summary.py kernels $db > $O/bench_kernels.md
rm -rf $O/prof
tail -n 4 $O/bench_loop_dispatches.md
head -8 $O/bench_kernels.md | cut -c1-160
tail -1 $O/prof_bench.json | cut -c1-300
