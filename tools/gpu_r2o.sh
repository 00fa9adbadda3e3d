#!/bin/bash
# round 2, GPU pass o: 2-bit-window field inversion in the generic units -- parity and rates
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2o
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scalar or builtin or reference_binary or user_curve or cofactor or point_add or fp_ops" 2>&1 | tail -n 30 > $O/pytest.log
tail -n 4 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --parity-items 4096 --steps 5 --warmup 2"
for c in SECP384R1 SECP521R1 BRAINPOOLP256R1 WEI448 SECP224R1 SECP192R1 BRAINPOOLP320R1 BRAINPOOLP512R1; do
  timeout 300 $B --curve $c > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "import json;j=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]);r=j['roofline'];print('$c', round(j['value']/1e6,2), round(r['frac'],3), round(r['pipeline_frac'],3), r['pipeline_ms'])"
done
