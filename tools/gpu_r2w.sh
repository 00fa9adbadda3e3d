#!/bin/bash
# rocprofv3 kernel trace of the bench command (final state of round 2)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2w
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
db=$(ls -S $(find $O/prof -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/bench_kernels.md
[ -n "$db" ] && python $R/tools/rocpd_summary.py dispatches $db k_p256_loop > $O/bench_loop_dispatches.md
tail -n 3 $O/bench_loop_dispatches.md
head -n 10 $O/bench_kernels.md | cut -c1-160
find $O -name '*.db' -delete; find $O -size +1M -delete
