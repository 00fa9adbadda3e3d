#!/bin/bash
# round 2, GPU pass k: Ed25519 [h]A kernel split / preload, MSM loop pipelining (A/B variants)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2k
mkdir -p $O
cd $R
for v in base edsplit edsplitpre edpre; do
  L=$R/libecc_amd/lib/variants/libecc_amd_$v.so; [ $v = base ] && L=$R/libecc_amd/lib/libecc_amd.so
  ECAMD_LIB_PATH=$L timeout 300 python tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline > $O/ed_$v.json 2> $O/ed_$v.err || tail -3 $O/ed_$v.err
  python -c "import json;j=json.loads(open('$O/ed_$v.json').read().strip().splitlines()[-1]);print('$v', round(j['value']/1e6,2), j.get('ms_per_step'))"
done
for v in base edmpipe; do
  L=$R/libecc_amd/lib/variants/libecc_amd_$v.so; [ $v = base ] && L=$R/libecc_amd/lib/libecc_amd.so
  ECAMD_LIB_PATH=$L MSM_LOG2=18,20 MSM_K=0,8 timeout 300 python tools/bench_msm.py > $O/msm_$v.json 2> $O/msm_$v.err || tail -3 $O/msm_$v.err
  python -c "import json;j=json.load(open('$O/msm_$v.json'));print('$v', {k:{a:round(b,2) for a,b in v.items() if a.endswith('_ms')} for k,v in j.items()})"
done
