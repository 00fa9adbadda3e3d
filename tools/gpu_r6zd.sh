#!/bin/bash
# round 6: k_bkt_accum_g variants (tools/build_variant.py: two point records in flight; not pinned at the window loop's occupancy) on bip0340_msm
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zd
mkdir -p $O
cd $R
export TMPDIR=/tmp
for v in base p2 free p2free base p2 free p2free; do
  if [ $v = base ]; then unset ECAMD_LIB_PATH; else export ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so; fi
  timeout 300 python tools/bench_protocols.py --workload bip0340_msm --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 2> /dev/null | tail -1 | python -c "
import json, sys
j = json.loads(sys.stdin.read())
r = j.get('roofline') or {}
print('$v: %.3f ms, %.1f M/s, %s %.3f ms' % (j.get('ms_per_step', 0), j.get('value', 0) / 1e6, r.get('kernel'), r.get('kernel_ms') or 0))"
done
