#!/bin/bash
# Round 3, GPU pass e: the Goldilocks unit on 28-bit limbs (reduction folded into the product columns).
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/gpu_r3e.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
( timeout 120 python tools/x448_time.py ) > $O/x448_digest.json 2> $O/x448_digest.err; echo "rc=$?" >> $O/x448_digest.json
( time timeout 700 python -m pytest tests -m gpu -x -q -k "448 or xdh or eddsa or fallback or every_builtin or libecc_typed or fused or SECP384R1 or SECP224R1 or cofactor or rfc" --durations=6 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
timeout 300 python bench.py --curve WEI448 --no-traffic --no-secondary --steps 8 --warmup 3 > $O/bench_wei448.json 2> $O/bench_wei448.err
timeout 300 python tools/bench_protocols.py --workload ed448_verify --steps 5 --warmup 2 > $O/ed448_verify.json 2> $O/ed448_verify.err
timeout 300 python tools/bench_protocols.py --workload x448 --steps 5 --warmup 2 > $O/x448.json 2> $O/x448.err
for c in SECP384R1 SECP521R1; do
  timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 1024 --steps 6 --warmup 2 > $O/ecdsa_verify_$c.json 2> $O/ecdsa_verify_$c.err
done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_w448 -- python $R/bench.py --curve WEI448 --no-cpu-baseline --no-traffic --no-secondary --parity-items 1024 --steps 5 --warmup 2 > $O/prof_w448.json 2> $O/prof_w448.err
db=$(ls -S $(find $O/prof_w448 -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_wei448.md
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_ed448 -- python $R/tools/bench_protocols.py --workload ed448_verify --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_ed448.json 2> $O/prof_ed448.err
db=$(ls -S $(find $O/prof_ed448 -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ed448_verify.md
find $O -name '*.db' -delete; find $O -size +1M -delete
cat $O/x448_digest.json; tail -n 12 $O/pytest_subset.log
for f in bench_wei448 ed448_verify x448 ecdsa_verify_SECP384R1 ecdsa_verify_SECP521R1; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("frac"), (j.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
tail -3 $O/bench_wei448.err $O/ed448_verify.err $O/x448.err
head -12 $O/kernels_wei448.md | cut -c1-150; head -14 $O/kernels_ed448_verify.md | cut -c1-150
