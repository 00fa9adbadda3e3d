#!/bin/bash
# round 2, GPU pass m: Ed448 decoding on the Goldilocks radix-2^29 unit
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2m
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats.py tests/test_wycheproof.py -x -q -m gpu -k "448 or ed or edge or wycheproof" 2>&1 | tail -30 > $O/pytest_448.log
tail -n 4 $O/pytest_448.log
for v in new old; do
  [ $v = old ] && export ECAMD_NO_G448_DECODE=1
  timeout 300 python tools/bench_protocols.py --workload ed448_verify --no-cpu-baseline > $O/ed448_$v.json 2> $O/ed448_$v.err || tail -n 3 $O/ed448_$v.err
  python -c "import json;j=json.loads(open('$O/ed448_$v.json').read().strip().splitlines()[-1]);print('ed448_verify $v', round(j['value']/1e6,2), j.get('ms_per_step'))"
done
unset ECAMD_NO_G448_DECODE
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ed448 -- python $R/tools/bench_protocols.py --workload ed448_verify --no-cpu-baseline > $O/prof_ed448.json 2> $O/prof_ed448.err
db=$(ls -S $(find $O/prof_ed448 -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/kernels_ed448.md
head -n 12 $O/kernels_ed448.md | cut -c1-170
find $O -name '*.db' -delete; find $O -size +1M -delete
