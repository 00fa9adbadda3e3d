#!/bin/bash
# round 2, GPU pass n: kernel traces of the protocol workloads of BASELINE configs 4 and 5
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2n
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in ed25519_verify ecdsa_verify x25519; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python $R/tools/bench_protocols.py --workload $w --no-cpu-baseline > $O/prof_$w.json 2> $O/prof_$w.err
  db=$(ls -S $(find $O/prof_$w -name '*.db') | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_$w.md
  head -n 12 $O/kernels_$w.md | cut -c1-170
done
find $O -name '*.db' -delete; find $O -size +1M -delete
