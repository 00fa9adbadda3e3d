#!/bin/bash
# round 2, GPU pass i: affine-table (mixed addition) pipeline on the generic radix-2^29 units: parity, A/B, rates
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_parity.log
tail -5 $O/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_formats.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_multi.log
tail -5 $O/pytest_multi.log
B="python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 5 --warmup 2"
for c in SECP384R1 SECP521R1 BRAINPOOLP256R1 WEI448 SECP224R1 BRAINPOOLP512R1 WEI25519 SECP256K1; do
  timeout 300 $B --curve $c > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "import json;j=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]);print('$c', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
done
for v in jactab384 pre384; do
  ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so timeout 300 $B --curve SECP384R1 > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json;j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print('$v', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
done
for v in jactab521 pre521; do
  ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so timeout 300 $B --curve SECP521R1 > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json;j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print('$v', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
done
timeout 300 python tools/bench_secret_mode.py > $O/secret_mode.json 2> $O/secret_mode.err
cat $O/secret_mode.json
