#!/bin/bash
# round 6, first GPU pass: the changed paths under their tests, then where a typed-boundary verification call spends its time
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
export TMPDIR=/tmp
( lscpu | head -25; nproc; cat /sys/fs/cgroup/cpu.max; free -g | head -2 ) > $O/host.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q -k "schnorr_msm or secret_half or typed_boundary or test_gpu_multi or self_tests" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
ECAMD_COMPAT_TIMING=1 timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_bench.txt 2>&1
grep -v "^libecc_amd compat timing" $O/typed_bench.txt | tail -n 12
ECAMD_COMPAT_TIMING=1 timeout 300 libecc_amd/lib/compat_check bench_schnorr 20 > $O/typed_schnorr.txt 2>&1
grep -v "^libecc_amd compat timing" $O/typed_schnorr.txt | tail -n 6
cd /tmp
rm -rf /tmp/prof_typed
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_typed -o typed -- $R/libecc_amd/lib/compat_check benchv 20 > $O/prof_typed.log 2>&1
DB=$(find /tmp/prof_typed -name "*.db" | head -1)
python $R/tools/timeline.py $DB 30 > $O/timeline_ecdsa.md 2>&1
tail -n 45 $O/timeline_ecdsa.md
