#!/bin/bash
# Round 3, GPU pass j: the state at the end of the round -- the whole GPU suite, smoke, the default bench line (with its secondary
# records and live traffic), the curves and protocols whose fields changed this round, the typed boundary end to end.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r3j.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
( time timeout 400 python bench.py --ubench-json $O/ubench.json ) > $O/bench.json 2> $O/bench.err
B="python $R/bench.py --no-cpu-baseline --no-traffic --no-secondary --parity-items 4096 --steps 6 --warmup 2"
for c in SECP256K1 WEI448 WEI25519 BRAINPOOLP256R1 SECP224R1; do
  timeout 200 $B --curve $c > $O/bench_$c.json 2> $O/bench_$c.err
done
for w in x448 ed448_verify ecdsa_sign ecccdh; do
  timeout 200 python tools/bench_protocols.py --workload $w --no-cpu-baseline > $O/proto_$w.json 2> $O/proto_$w.err
done
for c in SECP384R1 SECP521R1; do
  timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 1024 > $O/proto_ecdsa_verify_$c.json 2> $O/proto_ecdsa_verify_$c.err
done
timeout 200 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --no-traffic --no-secondary --parity-items 1024 --steps 10 --warmup 3 > $O/prof_bench.json 2> $O/prof_bench.err
db=$(ls -S $(find $O/prof -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/bench_kernels.md
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 14 $O/pytest.log; tail -n 2 $O/smoke.log; tail -n 4 $O/bench.err; python - "$O/bench.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("HEADLINE", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("pipeline_frac"), j["roofline"].get("traffic"))
for s in j.get("secondary", []):
    print("  secondary", s.get("metric", "")[:60], s.get("value"), (s.get("roofline") or {}).get("frac"))
PY
for f in $O/bench_*.json $O/proto_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
cat $O/compat_bench_20.txt | cut -c1-200; head -9 $O/bench_kernels.md | cut -c1-150
