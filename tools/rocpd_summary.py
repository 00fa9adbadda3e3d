#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the small text files committed under profiles/.

usage: tools/rocpd_summary.py kernels <results.db>      per-kernel count / total / average duration
       tools/rocpd_summary.py pmc <results.db>          per-kernel, per-counter sums per dispatch
"""
import sqlite3
import sys


def kernels(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % | vgpr | sgpr | scratch | lds | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0][:80]} | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e6:.4f} | {r[4]/1e6:.4f} | {r[5]/1e6:.4f} | {100*r[2]/tot:.1f} | "
              f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


def pmc(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    rows = con.execute("select * from counters_collection").fetchall()
    ni, ci, vi, di = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    agg = {}
    for r in rows:
        agg.setdefault((r[ni], r[ci]), {}).setdefault(r[di], 0.0)
        agg[(r[ni], r[ci])][r[di]] += r[vi]
    print("| kernel | counter | dispatches | mean per dispatch | max per dispatch |")
    print("|---|---|---|---|---|")
    for (k, c), d in sorted(agg.items()):
        v = list(d.values())
        print(f"| {k[:70]} | {c} | {len(v)} | {sum(v)/len(v):.6g} | {max(v):.6g} |")


if __name__ == "__main__":
    {"kernels": kernels, "pmc": pmc}[sys.argv[1]](sys.argv[2])
