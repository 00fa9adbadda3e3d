#!/bin/bash
# Round 5, after the final pass (measurements only): the Schnorr multi-scalar kernels with the adopted items-per-lane rule under rocprofv3, and
# the typed boundary's end-to-end rates on the final build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PASS:-r5k}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/tools/bench_schnorr.py --curves SECP256K1,SECP256R1 --log2 18,20 --reps 3 > $O/bench_schnorr.md 2> $O/prof.err
cat $O/bench_schnorr.md
db=$(find $O/prof -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/msm_kernels.md
rm -rf $O/prof
head -24 $O/msm_kernels.md
( $R/libecc_amd/lib/compat_check bench 20 ) > $O/typed_bench.txt 2>&1
grep bench $O/typed_bench.txt
