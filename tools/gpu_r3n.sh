#!/bin/bash
# Round 3, GPU pass n: secp224r1 / secp192r1 on the signed sparse Montgomery reduction (units 224s / 192s).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
( time timeout 200 python -m pytest tests -m gpu -x -q -k "SECP192R1 or SECP224R1 or every_builtin or user_curve or fused" --durations=5 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
for c in SECP224R1 SECP192R1; do
  timeout 120 python bench.py --curve $c --no-cpu-baseline --no-traffic --no-secondary --parity-items 16384 --steps 6 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err
done
tail -n 9 $O/pytest_subset.log
for c in SECP224R1 SECP192R1; do python - "$O/bench_$c.json" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["parity_gate"])
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
tail -n 3 $O/bench_SECP224R1.err $O/bench_SECP192R1.err
