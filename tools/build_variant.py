#!/usr/bin/env python3
"""Build an A/B variant of the product library: recompile ONE translation unit of libecc_amd/csrc with extra flags and link
it with the other (already built) objects into libecc_amd/lib/variants/libecc_amd_<name>.so.  Run the variant with
ECAMD_LIB_PATH=<that file> (libecc_amd/api.py: developer override).  usage: build_variant.py NAME UNIT.hip [flags...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libecc_amd import build as b  # noqa: E402


def main():
    name, unit, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    # UNIT is a source file, or OBJECT=SOURCE for one of the per-size objects of ecamd_g29_kernel.hip (e.g. ecamd_g29_255c.o=ecamd_g29_kernel.hip:
    # the unit's own flags from libecc_amd/build.py are kept and the extra ones added)
    b.build()
    vdir = os.path.join(b.LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    target_obj = None
    if "=" in unit:
        target_obj, unit = unit.split("=")
    base_flags = []
    for src, o, fl in b._jobs():
        if (target_obj and o == target_obj) or (not target_obj and src == unit and o == os.path.splitext(unit)[0] + ".o"):
            base_flags, target_obj = fl, o
    obj = os.path.join(vdir, f"{os.path.splitext(target_obj)[0]}_{name}.o")
    subprocess.check_call([b.HIPCC] + b.FLAGS + base_flags + extra + ["-x", "hip", "-c", os.path.join(b.CSRC, unit), "-o", obj])
    objs = []
    for src, o, _ in b._jobs():
        objs.append(obj if o == target_obj else os.path.join(b.LIBDIR, o))
    lib = os.path.join(vdir, f"libecc_amd_{name}.so")
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-lpthread"])
    print(lib)


if __name__ == "__main__":
    main()
