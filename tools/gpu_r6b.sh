#!/bin/bash
# round 6, second GPU pass: the f4 records and tests, the typed boundary as JSON, and the same typed calls with several contexts per device
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -k "whole_batch_vs_the_reference or cofactor or abscissa_beyond" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for w in bip0340_msm ed25519_msm; do
  ( time timeout 400 python tools/bench_protocols.py --workload $w --ref-items 16384 --traffic --steps 10 --warmup 2 ) > $O/$w.json 2> $O/$w.err
  tail -c 1500 $O/$w.json; tail -n 4 $O/$w.err
done
( time timeout 400 libecc_amd/lib/compat_check benchj 20 ) > $O/benchj.json 2> $O/benchj.err
cat $O/benchj.json | cut -c1-400; tail -n 3 $O/benchj.err
for d in 0,0 0,0,0; do
  ECAMD_DEVICES=$d timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_bench_$d.txt 2>&1
  grep "^bench" $O/typed_bench_$d.txt
done
