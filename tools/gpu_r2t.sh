#!/bin/bash
# round 2, GPU pass t: items per inversion of the ECDSA preparation kernel (A/B)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2t
mkdir -p $O
cd $R
for v in base prep12 prep16; do
  L=$R/libecc_amd/lib/variants/libecc_amd_$v.so; [ $v = base ] && L=$R/libecc_amd/lib/libecc_amd.so
  for w in ecdsa_verify ecdsa_sign; do
    ECAMD_LIB_PATH=$L timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline > $O/${w}_$v.json 2> $O/${w}_$v.err || tail -n 3 $O/${w}_$v.err
    python -c "import json;j=json.loads(open('$O/${w}_$v.json').read().strip().splitlines()[-1]);print('$v $w', round(j['value']/1e6,2), j.get('ms_per_step'))"
  done
done
