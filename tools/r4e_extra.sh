cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${PASS:-r4v}
mkdir -p $O
B=$GRAFT_REPO_ROOT/libecc_amd/lib/compat_check
( timeout 300 $B 256 ) > $O/compat_check_256.txt 2>&1; grep -iE "FAIL|all ok" $O/compat_check_256.txt | tail -4
( timeout 300 $B bench 20 ) > $O/compat_bench_20.txt 2>&1; grep -E "bench " $O/compat_bench_20.txt | cut -c1-150
( ECAMD_NO_SECRET_COMB=1 timeout 300 $B bench 20 ) > $O/compat_bench_20_nosecretcomb.txt 2>&1; grep -E "bench .*(sign|key_pair)" $O/compat_bench_20_nosecretcomb.txt | cut -c1-150
( ECAMD_COMPAT_TIMING=1 timeout 300 $B bench 20 ) > $O/compat_bench_20_timing.txt 2>&1; grep -E "timing" $O/compat_bench_20_timing.txt | grep -v "1 chunks" | cut -c1-200 | tail -9
