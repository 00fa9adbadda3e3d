cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${PASS:-r4t}
mkdir -p $O
B=$GRAFT_REPO_ROOT/libecc_amd/lib/compat_check
( timeout 300 $B 256 ) > $O/compat_check_256.txt 2>&1; grep -iE "FAIL|all ok" $O/compat_check_256.txt | tail -12
( timeout 300 $B bench 20 ) > $O/compat_bench_20.txt 2>&1; grep -E "bench " $O/compat_bench_20.txt | cut -c1-150
( ECAMD_COMPAT_ED_TWO_PASS=1 timeout 300 $B bench 20 ) > $O/compat_bench_20_twopass.txt 2>&1; grep -E "bench .*EDDSA" $O/compat_bench_20_twopass.txt | cut -c1-150
( ECAMD_COMPAT_TIMING=1 timeout 300 $B bench 20 ) > $O/compat_bench_20_timing.txt 2>&1; grep -B12 "bench ec_verify_batch EDDSA" $O/compat_bench_20_timing.txt | grep -E "timing" | grep -v "1 chunks" | cut -c1-200 | tail -4
