#!/usr/bin/env python3
"""What bounds every kernel of the headline step ("A/B a fused table+affine kernel, or a committed profile showing the
loss"): per kernel of one secp256r1 scalar-multiplication batch its duration (rocprofv3 --kernel-trace), its HBM bytes (FETCH_SIZE /
WRITE_SIZE passes, tools/pmc.py), the bandwidth that makes, and how busy the VALUs were (SQ_ACTIVE_INST_VALU against GRBM_GUI_ACTIVE, as
tools/valu_counters.py).  A kernel near 1.0 VALU busy gains nothing from a fusion that only removes traffic; one far below it and far
below 8 TB/s is latency-bound.

    python tools/kernel_bound.py [--curve SECP256R1] > profiles/r4_kernel_bound.md          (on the GPU box; ~2 minutes)"""
import argparse
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc  # noqa: E402


def kernel_ms(child_cmd, timeout=300):
    """{kernel: (calls, ms of its longest dispatch)} from one --kernel-trace run"""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="ecamd_kt_", dir="/tmp")
    try:
        r = subprocess.run([exe, "--kernel-trace", "-d", tmp, "--"] + list(child_cmd), capture_output=True, text=True, timeout=timeout,
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        dbs = sorted(glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True), key=os.path.getsize)
        if r.returncode != 0 or not dbs:
            return None
        con = sqlite3.connect(dbs[-1])
        out = {}
        for name, st, en in con.execute("select name, start, end from kernels"):
            c, m = out.get(name, (0, 0.0))
            out[name] = (c + 1, max(m, (en - st) / 1e6))
        con.close()
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="SECP256R1")
    ap.add_argument("--batch-log2", default="20")
    ap.add_argument("--workload", default=None, help="a tools/bench_protocols.py workload instead of the scalar multiplication (ed25519_verify, x25519, ecdsa_verify ...)")
    a = ap.parse_args()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--traffic-child", "--curve", a.curve, "--batch-log2", a.batch_log2, "--steps", "2", "--warmup", "1"]
    if a.workload:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_protocols.py"), "--workload", a.workload, "--no-cpu-baseline", "--steps", "2", "--warmup", "1"]
        a.curve = a.workload
    ms = kernel_ms(cmd) or {}
    hbm, note = pmc.hbm_bytes_per_launch(cmd)
    sq, _ = pmc.valu_counters(cmd, counters=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES",))
    gr, _ = pmc.valu_counters(cmd, counters=("GRBM_GUI_ACTIVE",))
    print(f"# What bounds each kernel of one {a.curve} batch of 2^{a.batch_log2} items\n")
    print("`tools/kernel_bound.py` on one MI355X: durations from `rocprofv3 --kernel-trace` (longest dispatch = the full-size launch), HBM bytes = "
          "(2 x FETCH_SIZE + WRITE_SIZE) KiB of the largest dispatch, VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); "
          "each counter in its own pass.\n")
    print("| kernel | ms | HBM MB | GB/s | of 8 TB/s | VALU instr. / wave | VALU busy |")
    print("|---|---|---|---|---|---|---|")
    rows = []
    for k, (calls, m) in ms.items():
        if not k.startswith(("k_", "void k_")):
            continue
        b = (hbm or {}).get(k)
        v = (sq or {}).get(k, {})
        g = (gr or {}).get(k, {}).get("GRBM_GUI_ACTIVE")
        waves = v.get("SQ_WAVES") or 0
        busy = (4.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * g / 8.0)) if g else None
        rows.append((m, k, b, v.get("SQ_INSTS_VALU", 0) / waves if waves else 0, busy))
    for m, k, b, ipw, busy in sorted(rows, reverse=True):
        gbs = (b / 1e9) / (m / 1e3) if b and m else None
        print(f"| {k.split('(')[0]} | {m:.3f} | {b / 1e6 if b else 0:.1f} | {gbs or 0:.0f} | {(gbs or 0) / 8000:.3f} | {ipw:.0f} | "
              f"{'-' if busy is None else round(busy, 3)} |")
    print(f"\n({note})")


if __name__ == "__main__":
    main()
