#!/bin/bash
# round 6: kernel traces of the three whole-batch forms on the final build (profiles/r6_f4_kernels.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6u
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
for w in ed25519_msm bip0340_msm ed448_msm; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o $w -- python $R/tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 > $O/$w.log 2>&1
  python $R/tools/rocpd_summary.py kernels $O/prof_$w/${w}_results.db > $O/$w.kernels.md 2>&1
  head -25 $O/$w.kernels.md | cut -c1-160
  rm -rf $O/prof_$w
done
