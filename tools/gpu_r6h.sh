#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6h
mkdir -p $O
cd $R
export TMPDIR=/tmp
for algo in straus bucket; do
  ECAMD_SCHNORR_MSM_ALGO=$algo timeout 600 python tools/bench_schnorr.py --curves SECP256K1,SECP256R1,SECP384R1 --log2 15,16,17,18,19,20 > $O/sweep_$algo.md 2> $O/sweep_$algo.err
  cat $O/sweep_$algo.md | tail -n 20
done
for c in 13 14 15 16; do
  ECAMD_SCHNORR_MSM_ALGO=bucket ECAMD_SCHNORR_BKT_C=$c timeout 600 python tools/bench_schnorr.py --curves SECP256K1 --log2 17,18,19,20 > $O/sweep_c$c.md 2> $O/sweep_c$c.err
  tail -n 4 $O/sweep_c$c.md
done
cd /tmp
ECAMD_SCHNORR_MSM_ALGO=bucket timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bkt -o bkt -- python $R/tools/bench_protocols.py --workload bip0340_msm --ref-items 0 --no-cpu-baseline --steps 5 --warmup 1 > $O/prof.log 2>&1
python $R/tools/rocpd_summary.py kernels $(find /tmp/prof_bkt -name "*.db" | head -1) > $O/bkt_kernels.md 2>&1
head -40 $O/bkt_kernels.md
