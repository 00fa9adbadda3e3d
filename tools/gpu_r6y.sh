#!/bin/bash
# round 6: the Ed25519 combination filed chunk by chunk (A/B by $ECAMD_NO_ED_STREAM): tests, then the typed boundary
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6y
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -x -q -k "test_gpu_msm or whole_batch or typed_boundary or fullsize" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for ab in on off on off; do
  if [ $ab = off ]; then export ECAMD_NO_ED_STREAM=1; else unset ECAMD_NO_ED_STREAM; fi
  timeout 600 libecc_amd/lib/compat_check benchv 20 ed25519 2> /dev/null | grep -o '"call": "ec_verify_batch EDDSA25519", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*' | sed "s/^/$ab /"
done
unset ECAMD_NO_ED_STREAM
cd /tmp
rm -rf /tmp/prof_ed
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_ed -o typed -- $R/libecc_amd/lib/compat_check benchv 20 ed25519 > $O/prof_ed.log 2>&1
DB=$(find /tmp/prof_ed -name "*.db" | head -1)
python $R/tools/timeline.py $DB 18 > $O/timeline_ed25519.md 2>&1
tail -n 80 $O/timeline_ed25519.md
