#!/bin/bash
# Round 3, GPU pass i: occupancy A/B of the two hot 2^255 - 19 kernels after the new multiplication (ladder 103 VGPRs: 4 waves per SIMD,
# Edwards window loop 132: 3) against variants pinned to 5 / 4 waves.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3i
V=$R/libecc_amd/lib/variants
mkdir -p $O
cd $R
for rep in 1 2; do
  for v in prod occ54; do
    if [ $v = prod ]; then unset ECAMD_LIB_PATH; else export ECAMD_LIB_PATH=$V/libecc_amd_occ54.so; fi
    timeout 200 python tools/bench_protocols.py --workload x25519 --no-cpu-baseline --ref-items 0 --steps 8 --warmup 3 > $O/x25519_${v}_$rep.json 2> $O/x25519_${v}_$rep.err
    timeout 200 python tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline --ref-items 0 --steps 8 --warmup 3 > $O/ed25519_${v}_$rep.json 2> $O/ed25519_${v}_$rep.err
  done
done
unset ECAMD_LIB_PATH
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("kernel_ms"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
