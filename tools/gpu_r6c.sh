#!/bin/bash
# round 6, third GPU pass: the doubling chunk schedule of the host-pointer pipeline against the old one, and the EdDSA typed call's timeline
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_new.txt 2>&1
grep "^bench" $O/typed_new.txt
ECAMD_HOST_SCHEDULE=65536,524288 timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_old.txt 2>&1
grep "^bench" $O/typed_old.txt
ECAMD_HOST_SCHEDULE=65536,131072,262144,294912 timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_s3.txt 2>&1
grep "^bench" $O/typed_s3.txt
ECAMD_HOST_SCHEDULE=32768,65536,131072,262144 timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_s4.txt 2>&1
grep "^bench" $O/typed_s4.txt
ECAMD_HOST_SCHEDULE=65536,196608,393216 timeout 300 libecc_amd/lib/compat_check bench 20 > $O/typed_s5.txt 2>&1
grep "^bench" $O/typed_s5.txt
cd /tmp
rm -rf /tmp/prof_typed
cat > /tmp/ed.sh <<'EOS'
#!/bin/bash
exec $1/libecc_amd/lib/compat_check bench 20
EOS
chmod +x /tmp/ed.sh
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_typed -o typed -- /tmp/ed.sh $R > $O/prof_typed.log 2>&1
DB=$(find /tmp/prof_typed -name "*.db" | head -1)
python - $DB > $O/timeline_all.md 2>&1 <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
ev = []
for name, st, en in con.execute("select name, start, end from kernels"):
    ev.append((st, en, "K " + name.split("(")[0][-44:]))
cols = [r[1] for r in con.execute("pragma table_info(memory_copies)")]
q = "select start, end, size" + (", name" if "name" in cols else "") + " from memory_copies"
for row in con.execute(q):
    ev.append((row[0], row[1], "C %s %.1f MB" % (row[3] if len(row) > 3 else "copy", row[2] / 1e6)))
ev.sort()
t0 = ev[0][0]
busy = t0
print("| start ms | dur ms | gap ms | what |\n|---|---|---|---|")
for st, en, what in ev:
    gap = max(0.0, (st - busy) / 1e6)
    if (en - st) / 1e6 >= 0.04 or gap >= 0.2:
        print(f"| {(st - t0) / 1e6:.3f} | {(en - st) / 1e6:.3f} | {gap:.3f} | {what} |")
    busy = max(busy, en)
PY
wc -l $O/timeline_all.md
