#!/bin/bash
# Round 3, GPU pass f: secp521r1 on plain residues (18 limbs, 2^522 = 2 folded inside the product columns).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r3f.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
( time timeout 500 python -m pytest tests -m gpu -x -q -k "SECP521R1 or every_builtin or fallback or fused or user_curve or linearity" --durations=6 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
timeout 300 python bench.py --curve SECP521R1 --no-traffic --no-secondary --steps 8 --warmup 3 > $O/bench_secp521r1.json 2> $O/bench_secp521r1.err
timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve SECP521R1 --no-cpu-baseline --ref-items 4096 --steps 6 --warmup 2 > $O/ecdsa_verify_SECP521R1.json 2> $O/ecdsa_verify_SECP521R1.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_521 -- python $R/bench.py --curve SECP521R1 --no-cpu-baseline --no-traffic --no-secondary --parity-items 1024 --steps 5 --warmup 2 > $O/prof_521.json 2> $O/prof_521.err
db=$(ls -S $(find $O/prof_521 -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_secp521r1.md
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_v521 -- python $R/tools/bench_protocols.py --workload ecdsa_verify --curve SECP521R1 --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_v521.json 2> $O/prof_v521.err
db=$(ls -S $(find $O/prof_v521 -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ecdsa_verify_secp521r1.md
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 12 $O/pytest_subset.log
for f in bench_secp521r1 ecdsa_verify_SECP521R1; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("frac"), (j.get("config") or {}).get("parity_gate") if isinstance(j.get("config"), dict) else "")
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
tail -n 3 $O/bench_secp521r1.err; tail -n 3 $O/ecdsa_verify_SECP521R1.err
head -10 $O/kernels_secp521r1.md | cut -c1-150; head -12 $O/kernels_ecdsa_verify_secp521r1.md | cut -c1-150
