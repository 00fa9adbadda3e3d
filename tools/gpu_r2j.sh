#!/bin/bash
# round 2, GPU pass j: 25519 flavour on the affine-table pipeline, on-device ECDSA redo, PMC traffic of the secp256r1 loop, kernel traces
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_parity.log
tail -5 $O/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_multi.log
tail -5 $O/pytest_multi.log
B="python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 5 --warmup 2"
for c in WEI25519 SECP384R1 SECP521R1 BRAINPOOLP256R1; do
  timeout 300 $B --curve $c > $O/bench_$c.json 2> $O/bench_$c.err
  python -c "import json;j=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]);print('$c', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
  timeout 300 $B --curve $c --batch-log2 16 > $O/bench16_$c.json 2> $O/bench16_$c.err
  python -c "import json;j=json.loads(open('$O/bench16_$c.json').read().strip().splitlines()[-1]);print('$c 2^16', round(j['value']/1e6,2), j['roofline']['pipeline_ms'])"
done
timeout 300 python tools/bench_secret_mode.py > $O/secret_mode.json 2> $O/secret_mode.err
cat $O/secret_mode.json
for w in ecdsa_verify ed25519_verify x25519 ed448_verify ecdsa_sign ecccdh; do
  timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline > $O/proto_$w.json 2> $O/proto_$w.err || tail -3 $O/proto_$w.err
  python -c "import json;j=json.loads(open('$O/proto_$w.json').read().strip().splitlines()[-1]);print('$w', round(j['value']/1e6,2), j.get('ms_per_step'))"
done
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c -- python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 2 --warmup 1 > $O/pmc_$c.json 2> $O/pmc_$c.err
  db=$(ls -S $(find $O/pmc_$c -name '*.db') | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py pmc $db > $O/pmc_$c.md
  grep p256 $O/pmc_$c.md | cut -c1-200
done
for c in SECP384R1 SECP521R1; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -- $B --curve $c > $O/prof_$c.json 2> $O/prof_$c.err
  db=$(ls -S $(find $O/prof_$c -name '*.db') | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_$c.md
  head -8 $O/kernels_$c.md | cut -c1-200
done
find $O -name '*.db' -delete; find $O -size +1M -delete
