#!/bin/bash
# round 2, GPU pass s: waves per SIMD of the big-field loop kernels (A/B)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2s
mkdir -p $O
cd $R
B="python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 3 --warmup 1"
run() { # name lib curve
  ECAMD_LIB_PATH=$2 timeout 300 $B --curve $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python -c "import json;j=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]);print('$1', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
}
run base512 $R/libecc_amd/lib/libecc_amd.so BRAINPOOLP512R1
run w2_512 $R/libecc_amd/lib/variants/libecc_amd_w2_512.so BRAINPOOLP512R1
run base384 $R/libecc_amd/lib/libecc_amd.so SECP384R1
run w3_384 $R/libecc_amd/lib/variants/libecc_amd_w3_384.so SECP384R1
run base521 $R/libecc_amd/lib/libecc_amd.so SECP521R1
run w3_521 $R/libecc_amd/lib/variants/libecc_amd_w3_521.so SECP521R1
