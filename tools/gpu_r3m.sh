#!/bin/bash
# Round 3, GPU pass m: the v_mad_i64_i32 stream in ubench (secp384r1's signed reduction MADs) and the bench lines that use it.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
timeout 60 libecc_amd/lib/ubench > $O/ubench.json 2> $O/ubench.err; echo "rc=$?" >> $O/ubench.err
timeout 200 python bench.py --curve SECP384R1 --no-cpu-baseline --no-traffic --no-secondary --parity-items 4096 --steps 6 --warmup 2 > $O/bench_secp384r1.json 2> $O/bench_secp384r1.err
( time timeout 300 python bench.py ) > $O/bench.json 2> $O/bench.err
python - "$O/ubench.json" <<'PY'
import json, sys
u = json.loads(open(sys.argv[1]).read())
for k in ("v_mad_u64_u32", "v_mad_u64_u32_sgpr", "v_mad_i64_i32_sgpr"):
    print(k, u[k])
PY
for f in $O/bench_secp384r1.json $O/bench.json; do python - "$f" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], j["value"], j["roofline"]["frac"], j["roofline"]["peak"], [ (s["config"], s["value"], s["frac"]) for s in j.get("secondary", [])])
PY
done
tail -n 3 $O/bench.err
