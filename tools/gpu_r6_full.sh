#!/bin/bash
# round 6: the driver's two commands on the current tree -- the whole GPU suite, then the default bench line
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PASS:-r6_full}
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 8 $O/pytest_gpu.log
( time timeout 1200 python bench.py ${BENCH_ARGS} ) > $O/bench_line.json 2> $O/bench_line.err
tail -n 4 $O/bench_line.err
python - $O/bench_line.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("headline %.2f M/s, %.3f ms, loop %.3f ms frac %.3f pipeline %.3f analytic %s" % (j["value"] / 1e6, j["ms_per_step"], r["kernel_ms"], r["frac"], r["pipeline_frac"], r.get("frac_of_analytic_peak")))
for s in j.get("secondary", []):
    if "calls" in s:
        for c in s["calls"]:
            print("  typed:", c.get("call"), c.get("rate"), c.get("device_ratio"), (c.get("cpu_baseline") or {}).get("value"))
        print("  typed wall_s", s.get("wall_s"))
    else:
        print(" ", s.get("config"), s.get("value"), s.get("kernel"), s.get("frac"), s.get("pipeline_frac"), s.get("error"), "wall", round(s.get("wall_s", 0), 1))
PY
