#!/bin/bash
# round 6: per-kernel profile of the bucket evaluation, and the typed boundary (benchj) on the final build
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6m
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
for algo in bucket straus; do
  rm -rf /tmp/prof_$algo
  ECAMD_SCHNORR_MSM_ALGO=$algo timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$algo -o p -- python $R/tools/bench_protocols.py --workload bip0340_msm --ref-items 0 --no-cpu-baseline --steps 8 --warmup 2 > $O/prof_$algo.log 2>&1
  python $R/tools/rocpd_summary.py kernels $(find /tmp/prof_$algo -name "*.db" | head -1) > $O/kernels_$algo.md 2>&1
  head -22 $O/kernels_$algo.md
done
cd $R
( time timeout 400 $R/libecc_amd/lib/compat_check benchj 20 ) > $O/benchj.json 2> $O/benchj.err
cat $O/benchj.json | cut -c1-260
timeout 300 libecc_amd/lib/compat_check bench_schnorr 20 > $O/typed_schnorr.txt 2>&1
grep "^bench ec_verify" $O/typed_schnorr.txt
