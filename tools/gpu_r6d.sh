#!/bin/bash
# round 6, fourth GPU pass: the typed layer after the lock removal -- the check program (all six configurations run by the test), then benchj
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "typed_boundary or self_tests or cfg1" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
( time timeout 400 libecc_amd/lib/compat_check benchj 20 ) > $O/benchj.json 2> $O/benchj.err
cat $O/benchj.json | cut -c1-330; tail -n 3 $O/benchj.err
( time timeout 400 libecc_amd/lib/compat_check benchj 19 ) > $O/benchj19.json 2> $O/benchj19.err
tail -n 3 $O/benchj19.json | cut -c1-330
