#!/bin/bash
# the bench half of tools/final_pass.sh: the default bench line and the rocprofv3 kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PASS:-final_bench}
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 600 python bench.py ) > $O/bench_line.json 2> $O/bench_line.err; tail -n 3 $O/bench_line.err
python - $O/bench_line.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("headline", round(j["value"] / 1e6, 2), "M/s", round(j["ms_per_step"], 3), "ms; loop frac", round(r["frac"], 4), "pipeline", round(r.get("pipeline_frac", 0), 4),
      "traffic", r.get("traffic"), "cpu", j["cpu_baseline"]["value"], "on", j["cpu_baseline"]["cores"])
for s in j.get("secondary", []):
    if isinstance(s, dict):
        print(" ", s.get("config"), round(s["value"] / 1e6, 2), "M/s", "frac", s.get("frac"), "pipeline", s.get("pipeline_frac"), "traffic/alg", s.get("traffic_over_algorithmic"))
PY
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-secondary --no-traffic --no-cpu-baseline --parity-items 256 --steps 10 --warmup 3 > $O/prof_bench.json 2> $O/prof_bench.err
db=$(ls -S $(find $O/prof -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py dispatches $db k_p256_loop > $O/bench_loop_dispatches.md && python $R/tools/rocpd_summary.py kernels $db > $O/bench_kernels.md
rm -rf $O/prof
tail -n 3 $O/bench_loop_dispatches.md
