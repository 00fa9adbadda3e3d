# A/B of the prefetching affine / finalisation kernels against the variants built without it
cd $GRAFT_REPO_ROOT
O=gpurun_out/${PASS:-r4r}
mkdir -p $O
V=libecc_amd/lib/variants
one() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "value %.3f M/s" % (j["value"] / 1e6), "ms %.3f" % j["ms_per_step"], "kernel_ms", r.get("kernel_ms"), "frac", r.get("frac"), "pipeline", r.get("pipeline_frac"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
B="python bench.py --no-secondary --no-traffic --no-cpu-baseline --steps 10 --warmup 3 --parity-items 256"
one p256_pf $B
one p256_nopf env ECAMD_LIB_PATH=$PWD/$V/libecc_amd_nopf256.so $B
one p256_pf2 $B
one p256_nopf2 env ECAMD_LIB_PATH=$PWD/$V/libecc_amd_nopf256.so $B
one p384_pf $B --curve SECP384R1
one p384_nopf env ECAMD_LIB_PATH=$PWD/$V/libecc_amd_nopf384.so $B --curve SECP384R1
one p521_pf $B --curve SECP521R1
one p521_nopf env ECAMD_LIB_PATH=$PWD/$V/libecc_amd_nopf521.so $B --curve SECP521R1
P="python tools/bench_protocols.py --no-cpu-baseline --steps 8 --warmup 2"
one ecdsa256_pf $P --workload ecdsa_verify
one ecdsa256_nopf env ECAMD_LIB_PATH=$PWD/$V/libecc_amd_nopf256.so $P --workload ecdsa_verify
one ecdsa384_pf $P --workload ecdsa_verify --curve SECP384R1
one ecdsa384_nopf env ECAMD_LIB_PATH=$PWD/$V/libecc_amd_nopf384.so $P --workload ecdsa_verify --curve SECP384R1
