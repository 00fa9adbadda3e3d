#!/bin/bash
# the whole GPU suite at its default (full) sizes, as the driver runs it, with timings; smoke; bench (with the CPU baseline leg)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q
mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --ubench-json $O/ubench.json > $O/bench.json 2> $O/bench.err
tail -n 8 $O/pytest.log; tail -n 2 $O/smoke.log; head -c 600 $O/bench.json
