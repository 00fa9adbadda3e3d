#!/bin/bash
# bench.py on the given curves (one GPU, no secondary records): rate, the four pipeline kernels, the loop's fraction of the MAD stream
# usage: tools/curve_rates.sh OUTDIR CURVE...
O=$1; shift
mkdir -p $O
for c in "$@"; do
  timeout 300 python bench.py --curve $c --no-secondary --steps ${STEPS:-8} --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err
  python - $O/bench_$c.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(sys.argv[1].split("/")[-1], "M/s %.3f" % (j["value"] / 1e6), "ms %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in r.get("pipeline_ms", {}).items()},
          "frac %.3f pipeline %.3f" % (r["frac"], r.get("pipeline_frac", 0)), "traffic/alg %s" % r.get("traffic_over_algorithmic"), (j["config"].get("parity_gate") or "")[:60])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
