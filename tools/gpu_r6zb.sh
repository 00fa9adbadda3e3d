#!/bin/bash
# round 6: the bucket reduction's fold (ecamd_bkt_fold: $ECAMD_BKT_FOLD = 16 | 8 | 4 | 2) -- tests on the default, then the three whole-batch workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zb
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "test_gpu_schnorr_msm or test_gpu_msm or test_gpu_ed448_msm" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for f in 16 4 8 2 16 4; do
  export ECAMD_BKT_FOLD=$f
  for w in bip0340_msm ed25519_msm ed448_msm; do
    timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 > $O/${w}_$f.log 2>&1
    python - $O/${w}_$f.log $w $f <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("fold %s %s: %.3f ms, %.1f M/s, gate %s" % (sys.argv[3], sys.argv[2], j.get("ms_per_step", 0), j.get("value", 0) / 1e6, (j.get("config") or {}).get("parity_gate", j.get("gate"))))
except Exception as e:
    print("fold", sys.argv[3], sys.argv[2], "FAILED", e)
PY
  done
done
