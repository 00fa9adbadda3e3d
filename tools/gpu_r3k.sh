#!/bin/bash
# Round 3, GPU pass k: k_ecdsa_prep on the context's side stream, beside the table / affine kernels of the chunk (A/B through
# $ECAMD_NO_SIDE_STREAM on the same box), ECDSA tests.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
( time timeout 500 python -m pytest tests -m gpu -x -q -k "ecdsa or libecc_typed or two_streams or self_tests or crafted" --durations=5 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
for rep in 1 2; do
  for v in side noside; do
    if [ $v = side ]; then unset ECAMD_NO_SIDE_STREAM; else export ECAMD_NO_SIDE_STREAM=1; fi
    for c in SECP256R1 SECP384R1 SECP521R1; do
      timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 1024 --steps 8 --warmup 3 > $O/verify_${c}_${v}_$rep.json 2> $O/verify_${c}_${v}_$rep.err
    done
  done
done
unset ECAMD_NO_SIDE_STREAM
timeout 200 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20.txt 2>&1
tail -n 8 $O/pytest_subset.log
for f in $O/verify_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
head -3 $O/compat_bench_20.txt | cut -c1-160
