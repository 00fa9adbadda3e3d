#!/bin/bash
# round 6: timeline of the typed boundary's ECDSA ec_verify_batch call on the final build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6ze
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/prof_ec
ECAMD_COMPAT_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_ec -o typed -- $R/libecc_amd/lib/compat_check benchv 20 > $O/prof_ecdsa.log 2>&1
DB=$(find /tmp/prof_ec -name "*.db" | head -1)
python $R/tools/timeline.py $DB 26 > $O/timeline_ecdsa.md 2>&1
grep "timing\|bench ec" $O/prof_ecdsa.log | tail -6 | cut -c1-220
tail -n 60 $O/timeline_ecdsa.md
