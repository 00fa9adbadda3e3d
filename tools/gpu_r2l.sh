#!/bin/bash
# round 2, GPU pass l: split Ed25519 kernels adopted -- tests, protocol rates, MSM timing
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -30 > $O/pytest_msm.log
tail -4 $O/pytest_msm.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ed or xdh or x25519 or 25519 or compat or self" 2>&1 | tail -30 > $O/pytest_ed.log
tail -4 $O/pytest_ed.log
for w in ed25519_verify ecdsa_verify; do
  timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline > $O/proto_$w.json 2> $O/proto_$w.err || tail -3 $O/proto_$w.err
  python -c "import json;j=json.loads(open('$O/proto_$w.json').read().strip().splitlines()[-1]);print('$w', round(j['value']/1e6,2), j.get('ms_per_step'))"
done
MSM_LOG2=16,17,18,19,20 MSM_K=0,8 timeout 300 python tools/bench_msm.py > $O/bench_msm.json 2> $O/bench_msm.err || tail -3 $O/bench_msm.err
python -c "import json;j=json.load(open('$O/bench_msm.json'));print({k:{a:round(b,2) for a,b in v.items() if a.endswith('_ms') or a.endswith('speedup')} for k,v in j.items()})"
timeout 300 python bench.py --no-cpu-baseline --parity-items 4096 --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python -c "import json;j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('headline', round(j['value']/1e6,2), j['roofline']['frac'], j['roofline']['traffic'], j['roofline']['pipeline_ms'])"
