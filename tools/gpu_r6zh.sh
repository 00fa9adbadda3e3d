#!/bin/bash
# round 6: point records padded to one 128-byte line each -- tests of the whole-batch forms, the three workloads, kernel_bound
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6zh
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "test_gpu_schnorr_msm or test_gpu_msm or test_gpu_ed448_msm" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for i in 1 2; do
for w in bip0340_msm ed25519_msm ed448_msm; do
  timeout 300 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps 8 --warmup 2 --ref-items 0 2> /dev/null | tail -1 | python -c "
import json, sys
j = json.loads(sys.stdin.read())
r = j.get('roofline') or {}
print('$w: %.3f ms, %.1f M/s, %s %.3f ms' % (j.get('ms_per_step', 0), j.get('value', 0) / 1e6, r.get('kernel'), r.get('kernel_ms') or 0))"
done
done
for w in bip0340_msm ed25519_msm; do
  timeout 900 python tools/kernel_bound.py --workload $w > $O/kernel_bound_$w.md 2> $O/kernel_bound_$w.err
  grep "accum\|points\|file" $O/kernel_bound_$w.md | cut -c1-200
done
