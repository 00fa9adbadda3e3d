#!/bin/bash
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6f
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/prof_typed
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_typed -o typed -- $R/libecc_amd/lib/compat_check benchj 20 > $O/prof_typed.log 2>&1
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
grep -n "k_msm_loop_g" $O/timeline_all.md | tail -3
