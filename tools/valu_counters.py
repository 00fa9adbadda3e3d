#!/usr/bin/env python3
"""SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAVES and GRBM_GUI_ACTIVE of the dominant kernels of the
five BASELINE workloads, as markdown ("W_impl derived analytically from the kernel's own
parameters and cross-checked with rocprof SQ_INSTS_VALU / VALU-busy").

Every lane runs one item, so one wave-level v_mad_u64_u32 is 64 lane-MADs of the work model: a kernel whose model says W MADs per
item must show at least W VALU instructions per wave, and W / (SQ_INSTS_VALU / SQ_WAVES) is the share of the MADs in its
instruction stream BY COUNT (the roofline fraction weighs the same stream by issue cycles: 5.0 for a MAD, 2.6 / 4.5 for the rest).
VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), the guide's effective-clock reading.

    python tools/valu_counters.py > profiles/r4_valu_counters.md       (on the GPU box; ~3 minutes)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pmc  # noqa: E402


def bench_model(curve):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_model", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from oracles import CURVES
    cp = CURVES[curve]
    slen = (cp["q"].bit_length() + 7) // 8
    nw = (cp["p"].bit_length() + 31) // 32
    mm, mads, kname, kmads = bench.work_model(cp, nw, slen, 1 << 20)
    return kname, kmads


def main():
    py = sys.executable
    bench, proto = os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "tools", "bench_protocols.py")
    jobs = []
    for curve in ("SECP256R1", "SECP384R1", "SECP521R1"):
        kname, kmads = bench_model(curve)
        jobs.append((f"{curve} scalar mult", kname, kmads,
                     [py, bench, "--traffic-child", "--curve", curve, "--batch-log2", "20", "--steps", "2", "--warmup", "1"]))
    # the protocol kernels' models: tools/bench_protocols.py (kept in step with it by hand; the numbers are printed there too)
    jobs.append(("ECDSA verify secp256r1", "k_p256_verify_loop", 64 * (24 * 117 + 19 * 81) + 18 * (8 * 117 + 3 * 81) + 81 + 6 * 117,
                 [py, proto, "--workload", "ecdsa_verify", "--traffic-child", "--steps", "2", "--warmup", "1"]))
    jobs.append(("Ed25519 verify", "k_ed_smul_c25519<1>", 64 * (20 * 97 + 16 * 61),
                 [py, proto, "--workload", "ed25519_verify", "--traffic-child", "--steps", "2", "--warmup", "1"]))
    jobs.append(("X25519", "k_x25519_ladder", 255 * (5 * 97 + 4 * 61 + 9) + 97,
                 [py, proto, "--workload", "x25519", "--traffic-child", "--steps", "2", "--warmup", "1"]))
    print("# Round 4: VALU counters of the dominant kernels against the work models\n")
    print("`tools/valu_counters.py` on one MI355X, batch 2^20 per workload; rocprofv3 --pmc, the SQ counters in one pass, GRBM_GUI_ACTIVE in "
          "another (tools/pmc.py); the largest dispatch of each kernel.  W = the model's v_mad_u64_u32 per item (= per lane).\n")
    print("| workload | kernel | W (model) | SQ_WAVES | SQ_INSTS_VALU / wave | W share of VALU instr. | SQ_ACTIVE_INST_VALU | SQ_BUSY_CYCLES | "
          "SQ_WAVE_CYCLES | GRBM_GUI_ACTIVE | VALU busy |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    raw = {}
    for name, kernel, w, cmd in jobs:
        sq, note1 = pmc.valu_counters(cmd, counters=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES",))
        gr, note2 = pmc.valu_counters(cmd, counters=("GRBM_GUI_ACTIVE",))
        key = kernel.split("<")[0]
        rows = [(k, v) for k, v in (sq or {}).items() if key in k]
        if not rows:
            print(f"| {name} | {kernel} | {w} | - | - | - | - | - | - | - | {note1} |")
            continue
        k, v = max(rows, key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0))
        g = (gr or {}).get(k, {}).get("GRBM_GUI_ACTIVE")
        waves = v.get("SQ_WAVES") or 0
        per_wave = v.get("SQ_INSTS_VALU", 0) / waves if waves else 0
        busy = (4.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * g / 8.0)) if g else None
        raw[name] = {"kernel": k, "model_mads_per_item": w, **v, "GRBM_GUI_ACTIVE": g}
        print(f"| {name} | {kernel} | {w} | {waves:.0f} | {per_wave:.0f} | {w / per_wave if per_wave else 0:.3f} | {v.get('SQ_ACTIVE_INST_VALU', 0):.4g} | "
              f"{v.get('SQ_BUSY_CYCLES', 0):.4g} | {v.get('SQ_WAVE_CYCLES', 0):.4g} | {g or 0:.4g} | {busy if busy is None else round(busy, 3)} |")
    print("\nraw: `" + json.dumps(raw) + "`")


if __name__ == "__main__":
    main()
