#!/bin/bash
# Round 3, GPU pass p: the tail of the EdDSA verifications on the radix-2^29 / 2^28 units (k_ed_fin_g, ecamd_rcbg.h).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
( time timeout 150 python -m pytest tests -m gpu -x -q -k "eddsa or ed25519 or edge_fixtures or rfc8032 or zero_challenge or msm_verdict" --deselect tests/test_gpu_fullsize.py::test_ed25519_full_size_vs_reference_binary ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
timeout 100 python tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline --steps 6 --warmup 2 > $O/ed25519_verify.json 2> $O/ed25519_verify.err
timeout 100 python tools/bench_protocols.py --workload ed448_verify --no-cpu-baseline --steps 5 --warmup 2 > $O/ed448_verify.json 2> $O/ed448_verify.err
tail -n 6 $O/pytest_subset.log
for f in ed25519_verify ed448_verify; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), j["config"].get("parity_gate") if isinstance(j.get("config"), dict) else "")
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
done
tail -n 2 $O/ed25519_verify.err $O/ed448_verify.err
