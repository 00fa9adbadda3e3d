#!/bin/bash
# Round 3, GPU pass h: 2^255 - 19 multiplication with the high columns first (the fold rides in the low columns).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r3h.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests -m gpu -x -q -k "25519 or xdh or eddsa or msm or rfc or edge_fixtures or every_builtin or fallback or zero_challenge or encode_point or combination" --durations=6 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
timeout 300 python tools/bench_protocols.py --workload x25519 --steps 8 --warmup 3 > $O/x25519.json 2> $O/x25519.err
timeout 300 python tools/bench_protocols.py --workload ed25519_verify --steps 8 --warmup 3 > $O/ed25519_verify.json 2> $O/ed25519_verify.err
timeout 300 python bench.py --curve WEI25519 --no-cpu-baseline --no-traffic --no-secondary --parity-items 4096 --steps 8 --warmup 3 > $O/bench_wei25519.json 2> $O/bench_wei25519.err
timeout 200 python tools/bench_msm.py > $O/eddsa_msm.json 2> $O/eddsa_msm.err || true
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_ed -- python $R/tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_ed.json 2> $O/prof_ed.err
db=$(ls -S $(find $O/prof_ed -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ed25519_verify.md
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_x -- python $R/tools/bench_protocols.py --workload x25519 --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_x.json 2> $O/prof_x.err
db=$(ls -S $(find $O/prof_x -name '*.db') | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_x25519.md
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 12 $O/pytest_subset.log
for f in x25519 ed25519_verify bench_wei25519; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
tail -n 3 $O/x25519.err $O/ed25519_verify.err; tail -n 5 $O/eddsa_msm.json
head -10 $O/kernels_ed25519_verify.md | cut -c1-150; head -8 $O/kernels_x25519.md | cut -c1-150
