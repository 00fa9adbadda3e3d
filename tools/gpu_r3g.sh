#!/bin/bash
# Round 3, GPU pass g: secp384r1 with the Montgomery reduction on the four signed digits of p + 1.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r3g.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
( time timeout 500 python -m pytest tests -m gpu -x -q -k "SECP384R1 or every_builtin or fallback or fused or user_curve or linearity" --durations=6 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
timeout 300 python bench.py --curve SECP384R1 --no-traffic --no-secondary --steps 8 --warmup 3 > $O/bench_secp384r1.json 2> $O/bench_secp384r1.err
timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve SECP384R1 --no-cpu-baseline --ref-items 4096 --steps 6 --warmup 2 > $O/ecdsa_verify_SECP384R1.json 2> $O/ecdsa_verify_SECP384R1.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_384 -- python $R/bench.py --curve SECP384R1 --no-cpu-baseline --no-traffic --no-secondary --parity-items 1024 --steps 5 --warmup 2 > $O/prof_384.json 2> $O/prof_384.err
db=$(ls -S $(find $O/prof_384 -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_secp384r1.md
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_v384 -- python $R/tools/bench_protocols.py --workload ecdsa_verify --curve SECP384R1 --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_v384.json 2> $O/prof_v384.err
db=$(ls -S $(find $O/prof_v384 -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ecdsa_verify_secp384r1.md
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 12 $O/pytest_subset.log
for f in bench_secp384r1 ecdsa_verify_SECP384R1; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("frac"), (j.get("config") or {}).get("parity_gate") if isinstance(j.get("config"), dict) else "")
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
tail -n 3 $O/bench_secp384r1.err; tail -n 3 $O/ecdsa_verify_SECP384R1.err
head -10 $O/kernels_secp384r1.md | cut -c1-150; head -12 $O/kernels_ecdsa_verify_secp384r1.md | cut -c1-150
