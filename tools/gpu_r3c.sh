#!/bin/bash
# Round 3, GPU pass c: the comb addition of the fused verification as a kernel of its own (k_comb_add_g), BIP0340 on the GPU behind
# ec_verify_batch, the adaptive chunking of the typed boundary.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r3c.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q -k "fused or ecdsa or libecc_typed or self_tests or crafted or rfc" --durations=8 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
( time timeout 600 libecc_amd/lib/compat_check 256 ) > $O/compat_check.txt 2>&1; echo "rc=$?" >> $O/compat_check.txt
timeout 200 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20.txt 2>&1
timeout 200 libecc_amd/lib/compat_check bench 18 > $O/compat_bench_18.txt 2>&1
timeout 200 libecc_amd/lib/compat_check bench 16 > $O/compat_bench_16.txt 2>&1
for c in SECP384R1 SECP521R1 BRAINPOOLP256R1 SECP224R1 BRAINPOOLP512R1; do
  timeout 300 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 1024 --steps 5 --warmup 2 > $O/ecdsa_verify_$c.json 2> $O/ecdsa_verify_$c.err
done
cd /tmp; export TMPDIR=/tmp
for c in SECP384R1 SECP521R1; do
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$c -- python $R/tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_$c.json 2> $O/prof_$c.err
db=$(ls -S $(find $O/prof_$c -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ecdsa_verify_$c.md
done
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 5 $O/pytest_subset.log; tail -n 4 $O/compat_check.txt; cat $O/compat_bench_20.txt; for f in $O/ecdsa_verify_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'])"; done; head -8 $O/kernels_ecdsa_verify_SECP521R1.md | cut -c1-160
