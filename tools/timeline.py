#!/usr/bin/env python3
"""Timeline of the LAST `window_ms` of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd sqlite): kernels and copies in start
order with the idle gaps between them -- to see whether copies overlap kernels and where a host-pointer entry point waits.
usage: tools/timeline.py <results.db> [window_ms]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    ev = []
    for name, st, en in con.execute("select name, start, end from kernels"):
        ev.append((st, en, "K " + name.split("(")[0][-40:]))
    cols = [r[1] for r in con.execute("pragma table_info(memory_copies)")]
    if cols:
        q = "select start, end, size" + (", name" if "name" in cols else "") + " from memory_copies"
        for row in con.execute(q):
            ev.append((row[0], row[1], "C %s %.1f MB" % (row[3] if len(row) > 3 else "copy", row[2] / 1e6)))
    ev.sort()
    t_end = max(e[1] for e in ev)
    ev = [e for e in ev if e[0] >= t_end - win * 1e6]
    t0 = ev[0][0]
    busy_until = t0
    kern = cop = 0.0
    print("| start ms | dur ms | gap before (nothing running) ms | what |\n|---|---|---|---|")
    for st, en, what in ev:
        gap = max(0.0, (st - busy_until) / 1e6)
        if (en - st) / 1e6 >= 0.05 or gap >= 0.05:
            print(f"| {(st - t0) / 1e6:.3f} | {(en - st) / 1e6:.3f} | {gap:.3f} | {what} |")
        busy_until = max(busy_until, en)
        if what.startswith("K"):
            kern += (en - st) / 1e6
        else:
            cop += (en - st) / 1e6
    print(f"\nwindow {(t_end - t0) / 1e6:.2f} ms: kernels {kern:.2f} ms, copies {cop:.2f} ms")


if __name__ == "__main__":
    main()
