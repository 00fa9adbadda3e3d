#!/bin/bash
# Round 3, GPU pass b: the whole GPU suite on the new engine paths (fused double-scalar verification of the generic units,
# device-resident key import, RFC vectors, 2^16-item reference samples), the typed boundary again, its end-to-end rates over
# chunk sizes with the buffered get_random, the default bench.py line (live PMC traffic + secondary records), and ECDSA
# verification on secp384r1 / secp521r1 with and without the fused loop.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r3b.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 600 libecc_amd/lib/compat_check 256 ) > $O/compat_check.txt 2>&1; echo "rc=$?" >> $O/compat_check.txt
for c in 32768 65536 131072 262144; do
  ECAMD_COMPAT_CHUNK=$c timeout 200 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20_chunk$c.txt 2>&1
done
ECAMD_COMPAT_PUBLIC_SCALARS=1 timeout 200 libecc_amd/lib/compat_check bench 20 > $O/compat_bench_20_public_scalars.txt 2>&1
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
for c in SECP384R1 SECP521R1 BRAINPOOLP256R1; do
  timeout 300 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --steps 5 --warmup 2 > $O/ecdsa_verify_$c.json 2> $O/ecdsa_verify_$c.err
  ECAMD_NO_FUSED_VERIFY=1 timeout 300 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/ecdsa_verify_${c}_two_smul.json 2> $O/ecdsa_verify_${c}_two_smul.err
done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_v384 -- python $R/tools/bench_protocols.py --workload ecdsa_verify --curve SECP384R1 --no-cpu-baseline --ref-items 0 --steps 5 --warmup 2 > $O/prof_v384.json 2> $O/prof_v384.err
db=$(ls -S $(find $O/prof_v384 -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ecdsa_verify_SECP384R1.md
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 6 $O/pytest.log; tail -n 3 $O/compat_check.txt; head -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err; for f in $O/compat_bench_20_chunk*.txt; do head -3 $f; done; for f in $O/ecdsa_verify_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f', j['value'], j['ms_per_step'])"; done
