#!/bin/bash
# round 2, GPU pass x: Ed25519 verification with R's Weierstrass map moved to the shared inversion of k_ed_hA_fin
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2x
mkdir -p $O
cd $R
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_msm.py -x -q -m gpu -k "eddsa25519 or eddsa_verify or device_pointer_entry or edge_fixtures or zero_challenge or host_pipeline or fallback or chunked or verdict or large_batch" 2>&1 | tail -n 6 > $O/pytest.log
tail -n 3 $O/pytest.log
timeout 60 python tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline > $O/ed.json 2> $O/ed.err
python -c "import json;j=json.loads(open('$O/ed.json').read().strip().splitlines()[-1]);print('ed25519_verify late_map', round(j['value']/1e6,2), j.get('ms_per_step'))"
ECAMD_NO_ED_LATE_MAP=1 timeout 60 python tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline > $O/ed_old.json 2> $O/ed_old.err
python -c "import json;j=json.loads(open('$O/ed_old.json').read().strip().splitlines()[-1]);print('ed25519_verify old', round(j['value']/1e6,2), j.get('ms_per_step'))"
