#!/usr/bin/env python3
"""HBM bytes per kernel launch from PMC counters, shared by bench.py and tools/bench_protocols.py.

Two child runs of a workload under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, nothing else traced, as
MI355X_MICROARCH.md's HBM section prescribes -- and the guide's gfx950 correction: bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB.
Per kernel the LARGEST dispatch is reported (the full-size launches of the timed batch; smaller ones are redo lanes and set-up).
Never an estimate: (None, note) when rocprofv3 is missing or a pass fails.

Also `valu_counters()`: SQ_INSTS_VALU / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel (one pass per counter), the
cross-check of the analytic work models W_impl that SURVEY.md section 8d asks for."""
import glob
import os
import shutil
import sqlite3
import subprocess
import tempfile


def _rocprof():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    return exe if os.path.exists(exe) else None


def _pass(exe, counter, child_cmd, tmp, timeout):
    """one `rocprofv3 --pmc <counters>` run of child_cmd (counter: one name, or several separated by blanks -- they must fit one
    pass); returns {kernel: {dispatch_id: value}} for one counter, {kernel: {counter: {dispatch_id: value}}} for several, or an
    error string"""
    names = counter.split()
    d = os.path.join(tmp, "_".join(names))
    cmd = [exe, "--pmc"] + names + ["-d", d, "--"] + list(child_cmd)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getsize)
    if r.returncode != 0 or not dbs:
        return f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"
    con = sqlite3.connect(dbs[-1])
    cols = [x[1] for x in con.execute("pragma table_info(counters_collection)")]
    ni, ci, vi, di = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    agg = {}
    for row in con.execute("select * from counters_collection"):
        if row[ci] in names:
            a = agg.setdefault(row[ni], {}) if len(names) == 1 else agg.setdefault(row[ni], {}).setdefault(row[ci], {})
            a.setdefault(row[di], 0.0)
            a[row[di]] += row[vi]
    con.close()
    return agg


def hbm_bytes_per_launch(child_cmd, timeout=300):
    """({kernel name: bytes of its largest launch}, note) for the workload `child_cmd` runs"""
    exe = _rocprof()
    if not exe:
        return None, "rocprofv3 not found"
    per = {}
    tmp = tempfile.mkdtemp(prefix="ecamd_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            agg = _pass(exe, counter, child_cmd, tmp, timeout)
            if isinstance(agg, str):
                return None, agg
            for k, dd in agg.items():
                per.setdefault(k, {})[counter] = max(dd.values())
    except Exception as e:
        return None, f"PMC pass failed: {type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    by_kernel = {k: (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for k, v in per.items()
                 if k.startswith(("k_", "void k_"))}
    return by_kernel, ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this run's batch, separate passes; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB "
                       "per launch (largest dispatch of each kernel)")


def valu_counters(child_cmd, counters=("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"), timeout=300):
    """({kernel: {counter: value of the largest dispatch}}, note); one profiled child run per counter"""
    exe = _rocprof()
    if not exe:
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="ecamd_pmc_", dir="/tmp")
    try:
        for counter in counters:
            agg = _pass(exe, counter, child_cmd, tmp, timeout)
            if isinstance(agg, str):
                return (out or None), agg
            for k, dd in agg.items():
                if not k.startswith(("k_", "void k_")):
                    continue
                if len(counter.split()) == 1:
                    out.setdefault(k, {})[counter] = max(dd.values())
                else:
                    # several counters in one pass: the dispatch with the most VALU work (or the largest first counter) speaks for all
                    first = dd.get(counter.split()[0]) or next(iter(dd.values()))
                    best = max(first, key=first.get)
                    for c, per in dd.items():
                        out.setdefault(k, {})[c] = per.get(best, max(per.values()))
    except Exception as e:
        return (out or None), f"PMC pass failed: {type(e).__name__}: {e}"[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, "rocprofv3 --pmc <counter>, one pass per counter, largest dispatch of each kernel"
