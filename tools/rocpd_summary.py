#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the small text files committed under profiles/.

usage: tools/rocpd_summary.py kernels <results.db>      per-kernel count / total / average duration
       tools/rocpd_summary.py pmc <results.db>          per-kernel, per-counter sums per dispatch
       tools/rocpd_summary.py dispatches <results.db> <name part>   every dispatch of the matching kernels, in launch order
       tools/rocpd_summary.py clock <results.db>        effective shader clock per kernel: GRBM_GUI_ACTIVE / dispatch duration
                                                        (a --pmc GRBM_GUI_ACTIVE --kernel-trace run; MI355X_MICROARCH.md, "DVFS give-back")
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


def dispatches(db, part):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end - start, grid_x from kernels where name like ? order by start", (f"%{part}%",)).fetchall()
    print("| # | kernel | grid | ms |")
    print("|---|---|---|---|")
    for i, r in enumerate(rows):
        print(f"| {i} | {r[0][:60]} | {r[3]} | {r[2]/1e6:.4f} |")
    full = [r[2] / 1e6 for r in rows if r[3] == max(x[3] for x in rows)]
    if full:
        print(f"\nfull-size dispatches: {len(full)}, mean {sum(full)/len(full):.4f} ms, last ten mean {sum(full[-10:])/len(full[-10:]):.4f} ms")


def clock(db):
    con = sqlite3.connect(db)
    ccols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kcols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    rows = con.execute("select * from counters_collection").fetchall()
    ni, ci, vi, di = ccols.index("kernel_name"), ccols.index("counter_name"), ccols.index("value"), ccols.index("dispatch_id")
    per = {}   # dispatch_id -> [kernel, sum of values, number of rows (counter instances: XCDs / SEs)]
    for r in rows:
        if r[ci] != "GRBM_GUI_ACTIVE":
            continue
        e = per.setdefault(r[di], [r[ni], 0.0, 0])
        e[1] += r[vi]
        e[2] += 1
    dur = {}
    if "dispatch_id" in kcols:
        for d, st, en in con.execute("select dispatch_id, start, end from kernels"):
            dur[d] = en - st
    else:   # no id column in this view: the n-th counter record belongs to the n-th kernel in launch order
        ks = con.execute("select start, end from kernels order by start").fetchall()
        for d, (st, en) in zip(sorted(per), ks):
            dur[d] = en - st
    agg = {}
    for d, (k, v, n) in per.items():
        if d in dur and dur[d] > 0:
            agg.setdefault(k, []).append((v, n, dur[d]))
    print(f"(kernels view columns: {kcols}; counter rows per dispatch = instances summed)")
    print("| kernel | dispatches | mean ms | GRBM_GUI_ACTIVE (sum over instances) | instances | cycles per instance | effective clock MHz |")
    print("|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(x[2] for x in kv[1])):
        cyc = sum(x[0] for x in v) / len(v)
        inst = sum(x[1] for x in v) / len(v)
        ns = sum(x[2] for x in v) / len(v)
        print(f"| {k[:70]} | {len(v)} | {ns/1e6:.4f} | {cyc:.6g} | {inst:.1f} | {cyc/inst:.6g} | {1e3*cyc/inst/ns:.1f} |")


if __name__ == "__main__":
    if sys.argv[1] == "dispatches":
        dispatches(sys.argv[2], sys.argv[3])
    else:
        {"kernels": kernels, "pmc": pmc, "clock": clock}[sys.argv[1]](sys.argv[2])
