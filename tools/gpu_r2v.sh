#!/bin/bash
# final validation of round 2: the whole GPU suite at its default sizes, smoke, the bench line, and the end-to-end ec_verify_batch rate
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 100 libecc_amd/lib/compat_check bench 16 > $O/compat_bench_16.txt 2>&1
timeout 100 libecc_amd/lib/compat_check bench 18 > $O/compat_bench_18.txt 2>&1
tail -n 6 $O/pytest.log; tail -n 2 $O/smoke.log; head -c 300 $O/bench.json; cat $O/compat_bench_16.txt $O/compat_bench_18.txt
