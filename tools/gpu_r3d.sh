#!/bin/bash
# Round 3, GPU pass d: ECFSDSA behind ec_verify_batch, the bucket-MSM data-movement prototype, A/B of ECDSA_PREP_K and of the
# MAD-by-one folds of the plain-residue fields, the single-device leg of tools/scale_check.py.
#   /usr/local/graft/bin/gpurun --timeout 1300 -- 'bash tools/gpu_r3d.sh'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
V=$R/libecc_amd/lib/variants
mkdir -p $O
cd $R
( time timeout 500 libecc_amd/lib/compat_check 256 ) > $O/compat_check.txt 2>&1; echo "rc=$?" >> $O/compat_check.txt
# bucket MSM prototype (measurement only)
timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/msm_bucket_proto.hip -o $O/msm_bucket_proto > $O/msm_build.txt 2>&1
for lg in 20 18 16; do timeout 120 $O/msm_bucket_proto $lg 5 > $O/msm_bucket_$lg.json 2>&1; done
rm -f $O/msm_bucket_proto
# MAD-by-one folds: product (paired MADs) against the 64-bit additions (-DG29_FOLD_ADD variants), same box, alternating
for rep in 1 2; do
  for v in prod foldadd; do
    if [ $v = prod ]; then unset ECAMD_LIB_PATH; else export ECAMD_LIB_PATH=$V/libecc_amd_foldadd_255c.so; fi
    timeout 200 python tools/bench_protocols.py --workload x25519 --no-cpu-baseline --ref-items 0 --steps 8 --warmup 3 > $O/fold_x25519_${v}_$rep.json 2> $O/fold_x25519_${v}_$rep.err
    timeout 200 python tools/bench_protocols.py --workload ed25519_verify --no-cpu-baseline --ref-items 0 --steps 8 --warmup 3 > $O/fold_ed25519_${v}_$rep.json 2> $O/fold_ed25519_${v}_$rep.err
    if [ $v = foldadd ]; then export ECAMD_LIB_PATH=$V/libecc_amd_foldadd_256k.so; fi
    timeout 200 python bench.py --curve SECP256K1 --no-cpu-baseline --no-traffic --no-secondary --parity-items 1024 --steps 8 --warmup 3 > $O/fold_k256_${v}_$rep.json 2> $O/fold_k256_${v}_$rep.err
    if [ $v = foldadd ]; then export ECAMD_LIB_PATH=$V/libecc_amd_foldadd_448g.so; fi
    timeout 200 python bench.py --curve WEI448 --no-cpu-baseline --no-traffic --no-secondary --parity-items 1024 --steps 8 --warmup 3 > $O/fold_w448_${v}_$rep.json 2> $O/fold_w448_${v}_$rep.err
  done
done
unset ECAMD_LIB_PATH
# k_ecdsa_prep: items per inversion 8 (product) / 16 / 4
for c in SECP256R1 SECP384R1 SECP521R1; do
  for v in prod prepk16 prepk4; do
    if [ $v = prod ]; then unset ECAMD_LIB_PATH; else export ECAMD_LIB_PATH=$V/libecc_amd_$v.so; fi
    timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 0 --steps 6 --warmup 2 > $O/prepk_${c}_$v.json 2> $O/prepk_${c}_$v.err
  done
done
unset ECAMD_LIB_PATH
( time timeout 300 python tools/scale_check.py --gpus 1 --log2 18 ) > $O/scale_check_1.txt 2>&1; echo "rc=$?" >> $O/scale_check_1.txt
( time timeout 600 python -m pytest tests -m gpu -x -q -k "x25519 or xdh or ed25519 or eddsa or k256 or secp256k1 or 448 or rfc or libecc_typed" --durations=5 ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
find $O -size +1M -delete
tail -n 6 $O/compat_check.txt; grep -h "ECFSDSA\|ECKCDSA" $O/compat_check.txt; cat $O/msm_bucket_20.json
for f in $O/fold_*.json $O/prepk_*.json; do python - "$f" <<'EOF'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j.get("value"), j.get("ms_per_step"), (j.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1].split("/")[-1], "unreadable", e)
EOF
done
tail -n 5 $O/scale_check_1.txt; tail -n 4 $O/pytest_subset.log
