#!/bin/bash
# Round 3, GPU pass l: the final library state (k_ecdsa_prep beside the table kernels on secp256r1): whole GPU suite, smoke, the default
# bench line, the A/B of the side stream, the typed boundary end to end.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
( time timeout 400 python bench.py ) > $O/bench.json 2> $O/bench.err
for v in side noside; do
  if [ $v = side ]; then unset ECAMD_NO_SIDE_STREAM; else export ECAMD_NO_SIDE_STREAM=1; fi
  timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve SECP256R1 --no-cpu-baseline --ref-items 4096 --steps 8 --warmup 3 > $O/verify_p256_$v.json 2> $O/verify_p256_$v.err
done
unset ECAMD_NO_SIDE_STREAM
timeout 200 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20.txt 2>&1
tail -n 12 $O/pytest.log; tail -n 2 $O/smoke.log; tail -n 3 $O/bench.err
python - "$O/bench.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("HEADLINE", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("pipeline_frac"), j["roofline"].get("traffic"))
for s in j.get("secondary", []):
    print("  secondary", s.get("config"), s.get("value"), s.get("frac"))
PY
for f in $O/verify_p256_*.json; do python - "$f" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"))
PY
done
cat $O/compat_bench_20.txt | cut -c1-170
