#!/bin/bash
# round 6: the typed boundary without its pass over the keys (A/B by $ECAMD_COMPAT_FULL_SCAN) and with the key prefetch (A/B by $ECAMD_COMPAT_NO_PREFETCH)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6za
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "typed_boundary" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for ab in default fullscan noprefetch default fullscan noprefetch; do
  unset ECAMD_COMPAT_FULL_SCAN ECAMD_COMPAT_NO_PREFETCH
  if [ $ab = fullscan ]; then export ECAMD_COMPAT_FULL_SCAN=1; fi
  if [ $ab = noprefetch ]; then export ECAMD_COMPAT_NO_PREFETCH=1; fi
  ECAMD_COMPAT_TIMING=1 timeout 600 libecc_amd/lib/compat_check benchv 20 bip0340 2> $O/benchv_bip_$ab.err | grep -o '"call": "ec_verify_batch BIP0340[^,]*", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*' | sed "s/^/$ab /"
  grep "timing" $O/benchv_bip_$ab.err | tail -4 | cut -c1-200
done
for ab in default fullscan default fullscan; do
  unset ECAMD_COMPAT_FULL_SCAN ECAMD_COMPAT_NO_PREFETCH
  if [ $ab = fullscan ]; then export ECAMD_COMPAT_FULL_SCAN=1; fi
  timeout 600 libecc_amd/lib/compat_check benchv 20 ed25519 2> /dev/null | grep -o '"call": "ec_verify_batch EDDSA25519", "n": [0-9]*, "ms": [0-9.]*, "rate": [0-9.]*, "accepted": [a-z]*' | sed "s/^/$ab /"
  timeout 600 libecc_amd/lib/compat_check benchv 20 2> /dev/null | grep "bench ec_verify_batch" | cut -c1-120 | sed "s/^/$ab /"
done
