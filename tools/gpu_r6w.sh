#!/bin/bash
# round 6: the EdDSA encoding step on the 2^255 - 19 unit (k_ed_enc_c25519; A/B by $ECAMD_NO_ED_ENC_G) -- tests, then the typed boundary's benchj
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6w
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -x -q -k "eddsa or ed25519 or encode or sign or typed_boundary or secret_half or gpu_hash or gpu_msm" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for ab in on off; do
  if [ $ab = off ]; then export ECAMD_NO_ED_ENC_G=1; else unset ECAMD_NO_ED_ENC_G; fi
  ( time timeout 600 libecc_amd/lib/compat_check benchj 20 ) > $O/benchj_$ab.json 2> $O/benchj_$ab.err
  python - $O/benchj_$ab.json <<'PY'
import json, sys
t = open(sys.argv[1]).read()
j = json.loads(t[t.index("{"):])
for r in j["records"]:
    print(sys.argv[1].split("/")[-1], r.get("call"), "ms", r.get("ms"), "rate %.3e" % r["rate"] if "rate" in r else r)
PY
done
