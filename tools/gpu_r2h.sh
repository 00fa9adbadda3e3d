#!/bin/bash
# round 2, GPU pass h: long / masked scalars on the generic radix-2^29 window kernel; MSM kernel breakdown
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "secret or blind" 2>&1 | tail -30 > $O/pytest_secret.log
tail -5 $O/pytest_secret.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scalar or long or builtin or ecdsa or cdh" 2>&1 | tail -30 > $O/pytest_parity.log
tail -5 $O/pytest_parity.log
timeout 300 python tools/bench_secret_mode.py > $O/secret_mode.json 2> $O/secret_mode.err
cat $O/secret_mode.json
for c in WEI25519 SECP256K1 SECP384R1 SECP521R1 BRAINPOOLP256R1; do
  timeout 300 python bench.py --no-cpu-baseline --parity-items 1024 --curve $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "import json;j=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]);print('$c', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
done
cd /tmp; export TMPDIR=/tmp
MSM_LOG2=20 MSM_K=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_msm -- python $R/tools/bench_msm.py > $O/prof_msm.json 2> $O/prof_msm.err
db=$(ls -S $(find $O/prof_msm -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_msm.md
cat $O/kernels_msm.md | cut -c1-200
find $O -name '*.db' -delete; find $O -size +1M -delete
