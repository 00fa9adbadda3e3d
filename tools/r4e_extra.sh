# scratch helper of the typed-boundary passes: compat_check benchv under the chunking variants, then a timeline of the default
cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${PASS:-r4l}
mkdir -p $O
B=$GRAFT_REPO_ROOT/libecc_amd/lib/compat_check
( timeout 200 $B benchv 20 ) > $O/benchv_default.txt 2>&1
( ECAMD_COMPAT_CHUNK=262144 ECAMD_HOST_CHUNK=262144 timeout 200 $B benchv 20 ) > $O/benchv_quarter.txt 2>&1
( ECAMD_COMPAT_CHUNK=524288 ECAMD_HOST_CHUNK=524288 timeout 200 $B benchv 20 ) > $O/benchv_half.txt 2>&1
( ECAMD_COMPAT_CHUNK=524288 ECAMD_HOST_CHUNK=262144 timeout 200 $B benchv 20 ) > $O/benchv_half_q.txt 2>&1
( ECAMD_NO_PRJ_IMPORT_G29=1 timeout 200 $B benchv 20 ) > $O/benchv_satimport.txt 2>&1
( timeout 200 $B benchv 20 384 ) > $O/benchv_p384.txt 2>&1
( ECAMD_NO_PRJ_IMPORT_G29=1 timeout 200 $B benchv 20 384 ) > $O/benchv_p384_satimport.txt 2>&1
for f in p384 p384_satimport default quarter half half_q satimport; do echo "== $f"; grep -E "M/s|ms" $O/benchv_$f.txt | cut -c1-160 | tail -8; done
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof -- $B benchv 20 > $O/benchv_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
DB=$(ls -S $O/prof/*/*.db | head -1)
python tools/timeline.py $DB 45 > $O/timeline.md
rm -rf $O/prof
cat $O/timeline.md | cut -c1-150
