#!/bin/bash
# round 6: the single-inversion Ed448 decoding -- every Ed448 / EdDSA test, then the item form and the whole-batch form timed
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6t
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "448 or eddsa or ed448 or edge_fixtures or wycheproof or typed_boundary" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
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
for w in ed448_verify ed448_msm; do
  timeout 400 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 8 --warmup 2 --ref-items 1024 > $O/$w.json 2> $O/$w.err
  show $O/$w.json; tail -n 2 $O/$w.err | grep -v amdgpu.ids
done
for lg in 16 17 18; do
  timeout 400 python tools/bench_protocols.py --workload ed448_verify --batch-log2 $lg --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 > $O/ed448_verify_$lg.json 2> $O/ed448_verify_$lg.err
  show $O/ed448_verify_$lg.json
  timeout 400 python tools/bench_protocols.py --workload ed448_msm --batch-log2 $lg --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 > $O/ed448_msm_$lg.json 2> $O/ed448_msm_$lg.err
  show $O/ed448_msm_$lg.json
done
