#!/bin/bash
# round 6: timelines (kernels + copies) of the typed boundary's EDDSA25519 and BIP0340 ec_verify_batch calls, 2^20 items
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6x
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
for w in ed25519 bip0340; do
  rm -rf /tmp/prof_$w
  ECAMD_COMPAT_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_$w -o typed -- $R/libecc_amd/lib/compat_check benchv 20 $w > $O/prof_$w.log 2>&1
  DB=$(find /tmp/prof_$w -name "*.db" | head -1)
  python $R/tools/timeline.py $DB 40 > $O/timeline_$w.md 2>&1
  grep "rate\|timing" $O/prof_$w.log | tail -12 | cut -c1-300
  tail -n 70 $O/timeline_$w.md
done
