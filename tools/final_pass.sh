#!/bin/bash
# End-of-round GPU pass on the tree as committed (about 14 GPU-minutes): the whole GPU suite as the driver runs it, smoke(), the default
# bench line, and the rocprofv3 kernel trace of the bench command with every dispatch of the dominant kernel listed.
#   gpurun --timeout 1500 -- 'PASS=r4w bash tools/final_pass.sh'        outputs: gpurun_out/$PASS/ (copy what is kept into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
PASS=${PASS:-final}
O=$R/gpurun_out/$PASS
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout ${PYTEST_T:-1100} python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 14 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
( time timeout 600 python bench.py ) > $O/bench_line.json 2> $O/bench_line.err; tail -n 3 $O/bench_line.err
python - $O/bench_line.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("headline", round(j["value"] / 1e6, 2), "M/s", round(j["ms_per_step"], 3), "ms; loop frac", round(r["frac"], 4), "pipeline", round(r.get("pipeline_frac", 0), 4),
      "traffic", r.get("traffic"), "cpu", j["cpu_baseline"]["value"], "on", j["cpu_baseline"]["cores"])
for s in j.get("secondary", []):
    if isinstance(s, dict):
        rr = s.get("roofline") or {}
        print(" ", s.get("config"), round(s["value"] / 1e6, 2), "M/s", "frac", s.get("frac", rr.get("frac")), "pipeline", s.get("pipeline_frac", rr.get("pipeline_frac")))
PY
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-secondary --no-traffic --no-cpu-baseline --parity-items 256 --steps 10 --warmup 3 > $O/prof_bench.json 2> $O/prof_bench.err
db=$(ls -S $(find $O/prof -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py dispatches $db k_p256_loop > $O/bench_loop_dispatches.md && python $R/tools/rocpd_summary.py kernels $db > $O/bench_kernels.md
rm -rf $O/prof
tail -n 6 $O/bench_loop_dispatches.md
