#!/bin/bash
# A/B of the Schnorr loop's one-ahead prefetch: tests first, then bench_schnorr with the library as built and with the -DMSM_NO_PREFETCH variant
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PASS:-r5l}
mkdir -p $O
cd $R
( timeout 400 python -m pytest tests/test_gpu_schnorr_msm.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "" ${VARIANTS:-msmnopf}; do
  if [ -n "$v" ]; then export ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so; fi
  timeout 300 python tools/bench_schnorr.py --curves SECP256K1 --log2 ${LOG2:-18,19,20} --reps 5 --k ${KS:-0,8} > $O/bench_${v:-default}.md 2>$O/bench_${v:-default}.err
  echo "== ${v:-default}"; tail -7 $O/bench_${v:-default}.md
done
