#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6g
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_schnorr_msm.py -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
for algo in straus bucket; do
  ( time ECAMD_SCHNORR_MSM_ALGO=$algo timeout 400 python tools/bench_protocols.py --workload bip0340_msm --ref-items 4096 --steps 10 --warmup 2 ) > $O/bip_$algo.json 2> $O/bip_$algo.err
  python - $O/bip_$algo.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "%.2f M/s"%(j["value"]/1e6), "ms", round(j["ms_per_step"],3), r.get("kernel"), r.get("kernel_ms"), j["config"]["parity_gate"][:100])
except Exception as e:
    print("unreadable", e)
PY
  tail -n 3 $O/bip_$algo.err
done
cd /tmp
ECAMD_SCHNORR_MSM_ALGO=bucket timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bkt -o bkt -- python $R/tools/bench_protocols.py --workload bip0340_msm --ref-items 0 --no-cpu-baseline --steps 5 --warmup 1 > $O/prof.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_bkt -name "*.db" | head -1) > $O/bkt_kernels.md 2>&1
head -30 $O/bkt_kernels.md
