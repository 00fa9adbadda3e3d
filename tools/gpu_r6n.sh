#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6n
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fullsize.py -m gpu -x -q -k "msm or ed25519_whole or combination or device_pointer" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
run() {
  env $1 timeout 400 python tools/bench_protocols.py --workload ed25519_msm --ref-items 0 --no-cpu-baseline --steps 10 --warmup 2 $2 $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$*: %.2f M/s %.3f ms kernel %s %.3f'%(j['value']/1e6, j['ms_per_step'], r.get('kernel'), r.get('kernel_ms')))"
}
run ECAMD_ED_MSM_ALGO=bucket
run ECAMD_ED_MSM_ALGO=straus
run ECAMD_ED_MSM_ALGO=bucket --batch-log2 18
run ECAMD_ED_MSM_ALGO=straus --batch-log2 18
run ECAMD_ED_MSM_ALGO=bucket --batch-log2 17
run ECAMD_ED_MSM_ALGO=straus --batch-log2 17
