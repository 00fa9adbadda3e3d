#!/bin/bash
# round 6: is k_bkt_accum_g bound by its gathers?  A variant whose gathers all land in 4096 records (wrong sums; the kernel's time is what is read)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zf
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
for v in base local; do
  if [ $v = base ]; then unset ECAMD_LIB_PATH; else export ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so; fi
  rm -rf /tmp/prof_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o t -- python $R/tools/bench_protocols.py --workload bip0340_msm --no-cpu-baseline --steps 4 --warmup 1 --ref-items 0 > $O/$v.log 2>&1
  DB=$(find /tmp/prof_$v -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py kernels $DB 2>/dev/null | grep "k_bkt_accum_g\|k_bkt_file\|k_bkt_reduce" | cut -c1-150 | sed "s/^/$v /"
  tail -2 $O/$v.log | cut -c1-200
done
