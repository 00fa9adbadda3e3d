#!/usr/bin/env python3
"""Static instruction mix of one kernel's hottest loop, from the gfx950 code object inside an object file under
libecc_amd/lib/ (no GPU needed): objcopy the .hip_fatbin section, unbundle, llvm-objdump, find the kernel by a
substring of its demangled name, take the largest backward-branch loop (or the whole body with --whole) and weigh every
instruction class with the issue cycles ubench measured (profiles/r3a_effective_clock.md): v_mad_u64_u32 / v_mad_i64_i32 5.0,
other VOP3 4.5, VOP2/VOP1 e32 2.6, everything else (SALU, s_waitcnt, memory) 1 (they issue beside the VALU).

usage: python tools/kernel_mix.py libecc_amd/lib/ecamd_g29_255c.o 'k_ed_smul_c25519<1>' [--whole] [--top 12]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
VOP2 = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_mov_b32", "v_cndmask_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32",
        "v_min_u32", "v_max_u32", "v_not_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_subbrev_co_u32", "v_subrev_co_u32"}


def disasm(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "a.fatbin"), os.path.join(tmp, "a.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={fat}", f"--output={co}", "--unbundle"])
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co], capture_output=True, text=True, check=True).stdout


def cost(op, enc_e64):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return 5.0
    if op.startswith("v_"):
        base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
        if base in VOP2 and not enc_e64:
            return 2.6
        return 4.5
    return 0.0


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    whole = "--whole" in sys.argv
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 14
    report(obj, pat, whole, top, verbose=True)


def report(obj, pat, whole=False, top=14, verbose=False):
    """{'instructions', 'valu', 'mads', 'weighted_cycles', 'mad_cycle_share', 'mad_count_share_of_valu'} of the kernel's hottest
    loop (or whole body); prints the table when verbose"""
    text = disasm(obj)
    # split into functions
    funcs = re.split(r"\n(?=[0-9a-f]{16} <)", text)
    body = None
    for f in funcs:
        head = f.split("\n", 1)[0]
        if pat in head:
            body = f
            break
    if body is None:
        raise SystemExit(f"no kernel matching {pat!r}")
    lines = []
    for ln in body.split("\n")[1:]:
        m = re.match(r"\s+(\S+)\s+(.*?)//\s*([0-9A-F]+):\s*(.*)$", ln)
        if not m:
            continue
        op, args, addr, enc = m.group(1), m.group(2), int(m.group(3), 16), m.group(4).split()
        lines.append((addr, op, args.strip(), len(enc)))
    lo, hi = lines[0][0], lines[-1][0]
    if not whole:
        # largest backward branch
        best = None
        for addr, op, args, nw in lines:
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>|<([^>+]*)>", args)
                off = None
                mm = re.search(r"\+0x([0-9a-f]+)>", args)
                if mm:
                    off = lines[0][0] + int(mm.group(1), 16)
                if off is not None and off < addr and (best is None or addr - off > best[1] - best[0]):
                    best = (off, addr)
        if best:
            lo, hi = best
    sel = [l for l in lines if lo <= l[0] <= hi]
    cnt, cyc = collections.Counter(), collections.Counter()
    for addr, op, args, nw in sel:
        e64 = nw >= 2 and not op.endswith("_e32") and op.startswith("v_") and ("_e64" in op or nw == 2)
        # objdump prints e32 encodings of VOP2 as one dword (or two with a literal); VOP3 always two
        is_vop3 = op.endswith("_e64") or (op.startswith("v_") and op.replace("_e32", "") not in VOP2 and not op.endswith("_e32"))
        c = cost(op, is_vop3)
        cnt[op] += 1
        cyc[op] += c
    total = sum(cyc.values())
    n = sum(cnt.values())
    mad = sum(v for k, v in cyc.items() if k.startswith("v_mad_u64_u32") or k.startswith("v_mad_i64_i32"))
    nmad = sum(v for k, v in cnt.items() if k.startswith("v_mad_u64_u32") or k.startswith("v_mad_i64_i32"))
    nvalu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    if verbose:
        print(f"kernel {pat}: {'whole body' if whole else f'loop 0x{lo:x}..0x{hi:x}'}: {n} instructions ({nvalu} VALU), {nmad} MADs, "
              f"{total:.0f} weighted VALU cycles, MAD share {mad / total:.3f} of the cycles, {nmad / max(1, nvalu):.3f} of the VALU instructions")
        for op, c in cyc.most_common(top):
            print(f"  {op:28s} {cnt[op]:6d}  {c:8.0f}  {c / total:6.3f}")
        rest = [(op, cnt[op]) for op in cnt if cyc[op] == 0]
        print("  non-VALU:", ", ".join(f"{op} {k}" for op, k in sorted(rest, key=lambda x: -x[1])[:10]))
    return {"instructions": n, "valu": nvalu, "mads": nmad, "weighted_cycles": total, "mad_cycle_share": mad / total,
            "mad_count_share_of_valu": nmad / max(1, nvalu)}


if __name__ == "__main__":
    main()
