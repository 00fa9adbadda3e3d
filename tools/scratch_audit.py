#!/usr/bin/env python3
"""Per-kernel private-segment (scratch) size, VGPR count and SPILLED-VGPR count of every gfx950 code object of libecc_amd.so,
read from the objects under libecc_amd/lib/ (no GPU needed): objcopy the .hip_fatbin section, unbundle the gfx950 code
object, parse the AMDGPU metadata note.  rocprofv3's `scratch` column is .private_segment_fixed_size: it is non-zero for
stack arrays that are indexed dynamically (exponent strings, word-major tables of the saturated kernels) as well as for
register spills; `.vgpr_spill_count` tells the two apart.

usage: python tools/scratch_audit.py > profiles/<name>.md"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(n):
    try:
        return subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(glob.glob(os.path.join(ROOT, "libecc_amd", "lib", "*.o"))):
            base = os.path.basename(o)[:-2]
            fat, co = os.path.join(tmp, base + ".fatbin"), os.path.join(tmp, base + ".co")
            if subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat], capture_output=True).returncode or not os.path.getsize(fat):
                continue
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}", "--unbundle"], capture_output=True)
            if not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for e in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", e) or [None, "?"])[1]
                rows.append((base, demangle(g("name")), int(g("private_segment_fixed_size")), int(g("vgpr_count")), int(g("vgpr_spill_count")),
                             int(g("sgpr_spill_count")), int(g("group_segment_fixed_size"))))
    spills = [r for r in rows if r[4] or r[5]]
    stack = [r for r in rows if r[2] and not (r[4] or r[5])]
    print("# Scratch audit of the gfx950 code objects (tools/scratch_audit.py; static metadata, no GPU run)\n")
    print(f"{len(rows)} kernels; {len(spills)} spill registers; {len(stack)} more have a private segment that is NOT a spill (stack arrays indexed at run "
          "time: exponent / scalar byte strings, the word-major window tables and multi-precision temporaries of the saturated-word kernels).\n")
    print("## Kernels with register spills\n\n| unit | kernel | scratch B | VGPRs | spilled VGPRs | spilled SGPRs |\n|---|---|---|---|---|---|")
    for r in sorted(spills, key=lambda r: -r[4]):
        print(f"| {r[0]} | `{r[1][:90]}` | {r[2]} | {r[3]} | {r[4]} | {r[5]} |")
    print("\n## Kernels with a private segment and no spill\n\n| unit | kernel | scratch B | VGPRs | LDS B |\n|---|---|---|---|---|")
    for r in sorted(stack, key=lambda r: (r[0], r[1])):
        print(f"| {r[0]} | `{r[1][:90]}` | {r[2]} | {r[3]} | {r[6]} |")


if __name__ == "__main__":
    main()
