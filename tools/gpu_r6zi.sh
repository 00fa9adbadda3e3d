#!/bin/bash
# round 6: the filing split by window range ($ECAMD_BKT_FILE_SPLIT) measured again behind the faster accumulation
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for v in off on off on; do
  if [ $v = on ]; then export ECAMD_BKT_FILE_SPLIT=1; else unset ECAMD_BKT_FILE_SPLIT; fi
  for w in bip0340_msm ed448_msm; do
  timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 2> /dev/null | tail -1 | python -c "
import json, sys
j = json.loads(sys.stdin.read())
print('split $v $w: %.3f ms, %.1f M/s' % (j.get('ms_per_step', 0), j.get('value', 0) / 1e6))"
  done
done
