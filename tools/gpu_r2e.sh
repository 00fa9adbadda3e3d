#!/bin/bash
# profiles of pipeline v2 (kernel trace, PMC), A/B of batching factors and waves per SIMD, the other fields with their kernel traces
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --parity-items 1024"
# 1. kernel trace of the headline bench
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_p256 -- $B --steps 10 --warmup 3 > $O/prof_p256_bench.json 2> $O/prof_p256.err
db=$(find $O/prof_p256 -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_p256.md
# 2. PMC passes (separate runs, counters only)
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c -- $B --steps 2 --warmup 1 > /dev/null 2> $O/pmc_$c.err
  db=$(find $O/pmc_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py pmc $db > $O/pmc_$c.md
done
# 3. A/B variants of the secp256r1 pipeline
for v in aff4 aff16 w4; do
  ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so timeout 300 $B --steps 10 --warmup 3 > $O/bench_$v.json 2> $O/bench_$v.err
done
timeout 300 $B --steps 10 --warmup 3 > $O/bench_base.json 2> $O/bench_base.err
# 4. the other fields: default build with kernel trace, then waves-per-SIMD variants
for c in WEI25519 SECP256K1 SECP384R1 SECP521R1 WEI448 BRAINPOOLP256R1; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -- $B --curve $c --steps 5 --warmup 2 > $O/bench_$c.json 2> $O/prof_$c.err
  db=$(find $O/prof_$c -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_$c.md
done
for w in 2 3 4; do
  ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_c25519w$w.so timeout 300 $B --curve WEI25519 --steps 5 --warmup 2 > $O/bench_WEI25519_w$w.json 2> $O/bench_WEI25519_w$w.err
done
for w in 2 3; do
  ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_k256w$w.so timeout 300 $B --curve SECP256K1 --steps 5 --warmup 2 > $O/bench_SECP256K1_w$w.json 2> $O/bench_SECP256K1_w$w.err
done
find $O -name "*.db" -delete; find $O -size +1M -delete
cd $R
for f in $O/bench_*.json $O/prof_p256_bench.json; do echo "$f $(python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(j['value']/1e6,2), j['roofline']['pipeline_ms'])" 2>/dev/null)"; done
