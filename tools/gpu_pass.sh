#!/bin/bash
# One parametrised GPU pass (replaces the per-pass scripts of rounds 2-3).  Run on the GPU box through gpurun:
#   gpurun --timeout 900 -- 'PASS=r4a PYTEST_K="eddsa or x25519" WORKLOADS="ed25519_verify x25519" bash tools/gpu_pass.sh'
# Environment:
#   PASS        name of the pass; outputs go to gpurun_out/$PASS/
#   PYTEST_K    -k expression for `pytest tests -m gpu` ("" = skip pytest, "ALL" = the whole GPU suite)
#   PYTEST_T    timeout of the pytest leg in seconds (default 300)
#   WORKLOADS   tools/bench_protocols.py workloads to time (each: --steps $STEPS --warmup 2, no CPU baseline)
#   AB          list of NAME=ENVVAR=VALUE triples: every workload is timed again with that variable set (A/B of a code path)
#   VARIANTS    names of libecc_amd/lib/variants/libecc_amd_<name>.so builds to time every workload with (tools/build_variant.py)
#   AB_WORKLOADS the workloads AB / VARIANTS apply to (default: all of WORKLOADS)
#   BENCH       "1": also run the default `python bench.py` line into bench_line.json
#   EXTRA       a command line run last, verbatim
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
PASS=${PASS:-pass}
O=$R/gpurun_out/$PASS
STEPS=${STEPS:-6}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -n "$PYTEST_K" ]; then
  if [ "$PYTEST_K" = "ALL" ]; then
    ( time timeout ${PYTEST_T:-900} python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  else
    ( time timeout ${PYTEST_T:-300} python -m pytest tests -m gpu -x -q -k "$PYTEST_K" ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  fi
  tail -n 8 $O/pytest.log
fi
show() {
python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "value", j.get("value"), "ms", j.get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"),
          "pipeline", r.get("pipeline_frac"), (j.get("config") or {}).get("parity_gate"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
PY
}
for w in $WORKLOADS; do
  timeout 150 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps $STEPS --warmup 2 > $O/$w.json 2> $O/$w.err
  show $O/$w.json; tail -n 2 $O/$w.err
  case " ${AB_WORKLOADS:-$WORKLOADS} " in *" $w "*) ;; *) continue ;; esac
  for ab in $AB; do
    name=${ab%%=*}; kv=${ab#*=}
    env "$kv" timeout 150 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps $STEPS --warmup 2 > $O/${w}_$name.json 2> $O/${w}_$name.err
    show $O/${w}_$name.json
  done
  for v in $VARIANTS; do
    ECAMD_LIB_PATH=$R/libecc_amd/lib/variants/libecc_amd_$v.so timeout 150 python tools/bench_protocols.py --workload $w --no-cpu-baseline --steps $STEPS --warmup 2 > $O/${w}_$v.json 2> $O/${w}_$v.err
    show $O/${w}_$v.json
  done
done
if [ "$BENCH" = "1" ]; then
  ( time timeout 600 python bench.py ) > $O/bench_line.json 2> $O/bench_line.err
  show $O/bench_line.json; tail -n 3 $O/bench_line.err
fi
if [ -n "$EXTRA" ]; then
  bash -c "$EXTRA" > $O/extra.log 2>&1; tail -n 30 $O/extra.log
fi
