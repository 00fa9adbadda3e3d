#!/usr/bin/env python3
"""Opcode-class histogram of a window loop, weighted by how often each region runs per window, from the gfx950 code object inside an
object file under libecc_amd/lib/ (static analysis, no GPU): the dynamic VALU instruction count it predicts is reconciled with the
SQ_INSTS_VALU counter of profiles/r4_valu_counters.md, and the cycles with GRBM_GUI_ACTIVE of the same run.

The window loop of k_p256_loop<8, false> is laid out by hipcc as  [mixed addition] [doubling, a rolled loop of four] [digit + look-up]
with the back edges  doubling -> doubling  (trip count 4)  and  look-up -> mixed addition  (one per window).  Regions are found from
the back edges: the smallest backward loop holding >= 300 MADs is the doubling; everything else between the lowest back-edge target
and the highest back-edge source runs once per window.

Classes (by opcode and immediate):
  mad            v_mad_u64_u32
  column end     the digit mask `v_and_b32 0x1fffffff` and the 64-bit `>> 29` that follow every product column (17 per multiplication)
  carry / fold   the 32-bit `>> 29`, masks and adds of carry() / fold() between multiplications
  limb add/sub   v_add_u32 / v_sub_u32 / v_add3 / v_lshl_add_u32 of field additions, subtractions with their 2p / 4p biases
  small multiple v_lshlrev_b32 by 1..3: 2a for the squarings' cross terms, 2 / 4 / 8 times an element
  select / table v_cndmask, v_alignbit (other than by 29), v_perm, v_or, v_bitop3, v_lshl_or: digit select, negation select, the
                 saturated-word -> 29-bit-digit conversion of a table entry, the scalar's shift register
  move           v_mov_b32 / v_mov_b64 (register shuffles around the asm statements)
  other VALU     compares, address arithmetic

usage: python tools/loop_isa_mix.py [obj] [kernel substring] > profiles/r5_loop_isa_mix.md"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_mix as km  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# issue cycles per wave64 instruction per SIMD at the sustained clock (profiles/ubench_r2g.json, r3a_effective_clock.md)
CYC = {"mad": 5.0, "vop3": 4.4, "vop2": 2.6}
VOP2 = km.VOP2 | {"v_lshlrev_b32", "v_lshrrev_b32"}


def parse(body):
    ins = []
    for ln in body.split("\n")[1:]:
        m = re.match(r"\s+(\S+)\s+(.*?)//\s*([0-9A-F]+):\s*(.*)$", ln)
        if not m:
            continue
        t = re.search(r"\+0x([0-9a-f]+)>", ln)
        ins.append({"addr": int(m.group(3), 16), "op": m.group(1), "args": m.group(2).strip(), "words": len(m.group(4).split("<")[0].split()),
                    "target": int(t.group(1), 16) if t else None})
    base = ins[0]["addr"]
    for i in ins:
        i["off"] = i["addr"] - base
    return ins


def classify(i):
    op, a = i["op"], i["args"]
    b = op.replace("_e32", "").replace("_e64", "")
    if b in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return "mad"
    if b == "v_and_b32" and "0x1fffffff" in a:
        return "column end"          # (carry() masks are counted here too; split below by the neighbouring shift)
    if b == "v_lshrrev_b64" and re.search(r"\], 29,", a):
        return "column end"
    if b == "v_alignbit_b32" and a.rstrip().endswith(", 29"):
        return "column end"
    if b == "v_lshrrev_b32" and re.match(r"v\d+, 29,", a):
        return "carry / fold"
    if b in ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add3_u32", "v_lshl_add_u32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"):
        return "limb add/sub"
    if b == "v_lshlrev_b32" and re.match(r"v\d+, [1-3],", a):
        return "small multiple"
    if b in ("v_cndmask_b32", "v_alignbit_b32", "v_perm_b32", "v_or_b32", "v_or3_b32", "v_bitop3_b32", "v_lshl_or_b32", "v_and_b32", "v_lshrrev_b32",
             "v_lshlrev_b32", "v_and_or_b32", "v_xor_b32", "v_bfe_u32", "v_lshlrev_b32_sdwa"):
        return "select / table"
    if b in ("v_mov_b32", "v_mov_b64", "v_accvgpr_write_b32", "v_accvgpr_read_b32"):
        return "move"
    if op.startswith("v_"):
        return "other VALU"
    return None


def cycles(i, cls):
    if cls == "mad":
        return CYC["mad"]
    b = i["op"].replace("_e32", "")
    is_vop3 = i["op"].endswith("_e64") or b not in VOP2
    return CYC["vop3"] if is_vop3 else CYC["vop2"]


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "libecc_amd", "lib", "ecamd_p256_kernel.o")
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_p256_loop<8, false>"
    text = km.disasm(obj)
    body = next(f for f in re.split(r"\n(?=[0-9a-f]{16} <)", text) if pat in f.split("\n", 1)[0])
    ins = parse(body)
    nm = lambda lo, hi: sum(1 for i in ins if lo <= i["off"] <= hi and i["op"].startswith("v_mad_u64"))
    back = [(i["target"], i["off"]) for i in ins if i["target"] is not None and i["target"] < i["off"] and (i["op"].startswith("s_cbranch") or i["op"] == "s_branch")]
    big = [(lo, hi) for lo, hi in back if nm(lo, hi) >= 300]
    inner = min(big, key=lambda b: b[1] - b[0])
    outer = (min(lo for lo, hi in big), max(hi for lo, hi in big))
    regions = {"doubling (x4 per window)": (inner, 4), "mixed addition + digit + look-up (x1)": (outer, 1)}
    hist = {}
    for name, ((lo, hi), trip) in regions.items():
        c, cy = collections.Counter(), collections.Counter()
        for i in ins:
            if not (lo <= i["off"] <= hi):
                continue
            if name.startswith("mixed") and inner[0] <= i["off"] <= inner[1]:
                continue
            cls = classify(i)
            if cls is None:
                c["(non-VALU: s_*, waitcnt, loads)"] += 1
                continue
            c[cls] += 1
            cy[cls] += cycles(i, cls)
        hist[name] = (c, cy, trip)
    prologue = sum(1 for i in ins if i["op"].startswith("v_") and not (outer[0] <= i["off"] <= outer[1]))
    order = ["mad", "column end", "carry / fold", "limb add/sub", "small multiple", "select / table", "move", "other VALU"]
    tot_c, tot_cy = collections.Counter(), collections.Counter()
    for c, cy, trip in hist.values():
        for k in c:
            tot_c[k] += trip * c[k]
            tot_cy[k] += trip * cy[k]
    valu = sum(tot_c[k] for k in order)
    cyc = sum(tot_cy[k] for k in order)
    NWIN, MEAS_VALU, GUI, WAVES_PER_SIMD = 64, 470074, 2.406e8 / 8, 16  # profiles/r4_valu_counters.md (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    meas_cyc_window = GUI / (WAVES_PER_SIMD * NWIN)
    print(f"# Round 5: opcode-class mix of `{pat}`'s window body (tools/loop_isa_mix.py; static, from {os.path.relpath(obj, ROOT)})\n")
    print(f"Regions from the back edges: doubling loop 0x{inner[0]:x}..0x{inner[1]:x} (trip count 4), window body 0x{outer[0]:x}..0x{outer[1]:x} (64 windows for a 32-byte scalar); "
          f"{prologue} VALU instructions outside the window loop (scalar recoding, first look-up, result store).\n")
    print("## Per region (static counts)\n\n| region | " + " | ".join(order) + " | VALU | non-VALU |\n|---|" + "---|" * (len(order) + 2))
    for name, (c, cy, trip) in hist.items():
        print(f"| {name} | " + " | ".join(str(c[k]) for k in order) + f" | {sum(c[k] for k in order)} | {c['(non-VALU: s_*, waitcnt, loads)']} |")
    print("\n## Per window (dynamic: 4 x doubling + 1 x the rest), cycles at the issue rates `ubench` measured "
          f"(v_mad_u64_u32 {CYC['mad']}, other VOP3 {CYC['vop3']}, VOP2/VOP1 {CYC['vop2']} per wave64 instruction per SIMD)\n")
    print("| class | instructions / window | share of VALU instr. | cycles / window | share of cycles |\n|---|---|---|---|---|")
    for k in order:
        print(f"| {k} | {tot_c[k]} | {tot_c[k] / valu:.3f} | {tot_cy[k]:.0f} | {tot_cy[k] / cyc:.3f} |")
    print(f"| **total** | **{valu}** | 1 | **{cyc:.0f}** | 1 |")
    pred = NWIN * valu + prologue
    print(f"\n## Reconciliation with the counters (`profiles/r4_valu_counters.md`, batch 2^20)\n")
    print(f"* predicted VALU instructions per wave: {NWIN} x {valu} + {prologue} = **{pred}**; measured `SQ_INSTS_VALU` / wave = **{MEAS_VALU}** "
          f"({(pred - MEAS_VALU) / MEAS_VALU * 100:+.2f} %: windows with digit 0 and the first window take shorter paths).")
    print(f"* MADs per window {tot_c['mad']} = 4 x (4M + 4S) + (8M + 3S) + bookkeeping with M = 117, S = 81 (model: 4347); MAD share of the VALU instructions "
          f"{tot_c['mad'] / valu:.3f} (counter: 0.592).")
    print(f"* measured cycles per wave-window: GRBM_GUI_ACTIVE / 8 XCDs / ({WAVES_PER_SIMD} waves per SIMD x {NWIN} windows) = **{meas_cyc_window:.0f}**; "
          f"the table's cycle model gives {cyc:.0f} ({cyc / meas_cyc_window:.2f} of measured -- VALU busy is 1.0, so the issue-rate model accounts for the time to within "
          f"{abs(1 - cyc / meas_cyc_window) * 100:.0f} %).  MAD cycles {tot_cy['mad']:.0f} = **{tot_cy['mad'] / meas_cyc_window:.3f}** of the measured window: that is `roofline.frac`.")
    nonmad = valu - tot_c["mad"]
    print(f"* the {nonmad} non-MAD instructions take the remaining {meas_cyc_window - tot_cy['mad']:.0f} cycles: {(meas_cyc_window - tot_cy['mad']) / nonmad:.2f} cycles each.")


if __name__ == "__main__":
    main()
