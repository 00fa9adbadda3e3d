#!/bin/bash
# round 6: the bucket evaluation's filing split by window range (A/B by $ECAMD_NO_BKT_FILE_SPLIT), tests of every whole-batch form first
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6v
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "ed448_msm or schnorr_msm or test_gpu_msm or whole_batch" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
show() {
python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "value %.4e" % j.get("value"), "ms %.3f" % j.get("ms_per_step"), "kernel", r.get("kernel"), "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"),
          "pipeline", r.get("pipeline_frac"), str((j.get("config") or {}).get("parity_gate"))[:60])
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
}
for w in bip0340_msm ed448_msm; do
  for ab in on off; do
    if [ $ab = off ]; then export ECAMD_NO_BKT_FILE_SPLIT=1; else unset ECAMD_NO_BKT_FILE_SPLIT; fi
    timeout 400 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 --ref-items 1024 > $O/${w}_$ab.json 2> $O/${w}_$ab.err
    show $O/${w}_$ab.json; tail -n 2 $O/${w}_$ab.err | grep -v amdgpu.ids
  done
done
unset ECAMD_NO_BKT_FILE_SPLIT
for lg in 17 18 19; do
  timeout 400 python tools/bench_protocols.py --workload bip0340_msm --batch-log2 $lg --no-cpu-baseline --steps 10 --warmup 2 --ref-items 0 > $O/bip0340_msm_$lg.json 2> $O/bip0340_msm_$lg.err
  show $O/bip0340_msm_$lg.json
done
