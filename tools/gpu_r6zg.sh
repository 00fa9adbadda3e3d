#!/bin/bash
# round 6: what bounds the kernels of the two bucket evaluations (tools/kernel_bound.py: duration, HBM bytes, VALU busy per kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zg
mkdir -p $O
cd $R
export TMPDIR=/tmp
for w in bip0340_msm ed25519_msm; do
  timeout 900 python tools/kernel_bound.py --workload $w > $O/kernel_bound_$w.md 2> $O/kernel_bound_$w.err
  cat $O/kernel_bound_$w.md | cut -c1-200
done
