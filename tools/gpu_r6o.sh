#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for lg in 17 20; do
ECAMD_ED_MSM_ALGO=bucket timeout 400 python tools/bench_protocols.py --workload ed25519_msm --ref-items 0 --no-cpu-baseline --steps 5 --warmup 1 --batch-log2 $lg 2>&1 | tail -5 | cut -c1-600
done
