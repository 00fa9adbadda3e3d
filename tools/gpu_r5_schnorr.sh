#!/bin/bash
# Round 5: the Schnorr-type multi-scalar multiplication at the typed boundary (A/B through the threshold) and its kernels under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PASS:-r5g}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LG=${LG:-19}
( $R/libecc_amd/lib/compat_check bench_schnorr $LG ) > $O/typed_msm.txt 2>&1
( ECAMD_COMPAT_SCHNORR_MSM_MIN=0 $R/libecc_amd/lib/compat_check bench_schnorr $LG ) > $O/typed_items.txt 2>&1
grep "bench" $O/typed_msm.txt $O/typed_items.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/tools/bench_schnorr.py --curves SECP256K1 --log2 20 --reps 3 > $O/bench_schnorr_prof.md 2> $O/prof.err
cat $O/bench_schnorr_prof.md
db=$(find $O/prof -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/msm_kernels.md || ls -R $O/prof | head -20
rm -rf $O/prof
head -40 $O/msm_kernels.md
