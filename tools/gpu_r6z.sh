#!/bin/bash
# round 6: the Schnorr-type combination filed chunk by chunk (A/B by $ECAMD_NO_SCHNORR_STREAM): tests, then the typed boundary's BIP0340 call
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6z
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "test_gpu_schnorr_msm or typed_boundary" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for ab in on off on off; do
  if [ $ab = off ]; then export ECAMD_NO_SCHNORR_STREAM=1; else unset ECAMD_NO_SCHNORR_STREAM; fi
  ECAMD_COMPAT_TIMING=1 timeout 600 libecc_amd/lib/compat_check benchv 20 bip0340 2> $O/benchv_$ab.err | grep -o '"call": "ec_verify_batch BIP0340[^,]*", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*' | sed "s/^/$ab /"
  grep "timing" $O/benchv_$ab.err | tail -4 | cut -c1-200
done
unset ECAMD_NO_SCHNORR_STREAM
timeout 600 libecc_amd/lib/compat_check benchv 20 ed25519 2> /dev/null | grep -o '"call": "ec_verify_batch EDDSA25519", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*'
cd /tmp
rm -rf /tmp/prof_bip
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_bip -o typed -- $R/libecc_amd/lib/compat_check benchv 20 bip0340 > $O/prof_bip.log 2>&1
DB=$(find /tmp/prof_bip -name "*.db" | head -1)
python $R/tools/timeline.py $DB 24 > $O/timeline_bip0340.md 2>&1
tail -n 100 $O/timeline_bip0340.md
