#!/bin/bash
# the whole GPU suite at its default (full) sizes, as the driver runs it, with timings; smoke; bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=20 ) > gpurun_out/r2f/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2f/smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
tail -6 gpurun_out/r2f/pytest.log; cat gpurun_out/r2f/smoke.log | tail -2; head -c 400 gpurun_out/r2f/bench.json
