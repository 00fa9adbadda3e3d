cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${PASS:-r4p}
mkdir -p $O
B=$GRAFT_REPO_ROOT/libecc_amd/lib/compat_check
export ECAMD_COMPAT_TIMING=1
run() { name=$1; shift; ( env "$@" timeout 200 $B benchv 20 $CURVE ) > $O/benchv_${CURVE}_$name.txt 2>&1; echo "== $CURVE $name"; grep -E "M verif" $O/benchv_${CURVE}_$name.txt | cut -c40-130; grep -E "timing" $O/benchv_${CURVE}_$name.txt | grep -v "1 chunks" | tail -2 | cut -c1-200; }
for CURVE in 256 384; do
run ramp_h19 A=1
run ramp_h18 ECAMD_HOST_CHUNK=262144
run noramp_h19 ECAMD_NO_HOST_RAMP=1
run noramp_h18 ECAMD_NO_HOST_RAMP=1 ECAMD_HOST_CHUNK=262144
run ramp32k_h19 ECAMD_HOST_RAMP_MIN=32768
run nostream ECAMD_COMPAT_NO_STREAM=1
done
unset ECAMD_COMPAT_TIMING
( timeout 300 $B 256 ) > $O/compat_check_256.txt 2>&1; tail -2 $O/compat_check_256.txt
( timeout 300 $B bench 20 ) > $O/compat_bench_20.txt 2>&1; grep -E "bench " $O/compat_bench_20.txt | cut -c1-150
