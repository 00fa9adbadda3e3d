#!/bin/bash
# round 6: Ed448 whole-batch verification against the item form, batch sizes and both evaluations (profiles/r6_ed448_msm.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6r
mkdir -p $O
cd $R
export TMPDIR=/tmp
show() {
python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "value %.3e" % j.get("value"), "ms %.3f" % j.get("ms_per_step"), "kernel", r.get("kernel"), "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"),
          "pipeline", r.get("pipeline_frac"), (j.get("config") or {}).get("parity_gate"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
}
timeout 300 python tools/bench_protocols.py --workload ed448_verify --no-cpu-baseline --steps 4 --warmup 1 > $O/ed448_verify.json 2> $O/ed448_verify.err
show $O/ed448_verify.json; tail -n 2 $O/ed448_verify.err
for lg in 20 18 17 16 14; do
  for algo in bucket straus; do
    ECAMD_SCHNORR_MSM_ALGO=$algo timeout 400 python tools/bench_protocols.py --workload ed448_msm --batch-log2 $lg --no-cpu-baseline --steps 6 --warmup 2 --ref-items 2048 > $O/ed448_msm_${lg}_$algo.json 2> $O/ed448_msm_${lg}_$algo.err
    show $O/ed448_msm_${lg}_$algo.json; tail -n 2 $O/ed448_msm_${lg}_$algo.err
  done
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o ed448 -- python $R/tools/bench_protocols.py --workload ed448_msm --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 > $O/prof.log 2>&1
ls $O/prof | head
