"""Build the HIP shared library in-tree: libecc_amd/lib/libecc_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libecc_amd.so")
SOURCES = ["ecamd_kernels.hip", "ecamd_p256_kernel.hip", "ecamd_hash.hip", "ecamd_host.cpp", "ecamd_multi.cpp"]
DEPS = ["ecamd_madchain.h", "ecamd_field.h", "ecamd_point.h", "ecamd_u29.h", "ecamd_p256.h", "ecamd_u29g.h", "ecamd_jacg.h",
        "ecamd_internal.h", "ecamd_lattice.h", "ecamd_randmod.h",
        "ecamd_curve_table.inc"]
# the public header only matters to the host-side translation units (the kernels see ecamd_internal.h)
HOST_DEPS = [os.path.join("..", "..", "include", "libecc_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-bitwise-instead-of-logical", "-DU29_ASM_MAD"]


def _stale(target, inputs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(i) > t for i in inputs)


G29_SIZES = [192, 224, 255, 256, 320, 384, 448, 511, 512, 521]


def _jobs():
    """(source, object, extra flags): ecamd_g29_kernel.hip is compiled once per field size and once
    more as the dispatcher, so that the big template instantiations build in parallel."""
    jobs = [(src, os.path.splitext(src)[0] + ".o", []) for src in SOURCES]
    # the 19-limb dense units need two registers too many for two waves per SIMD; pinned there the window loop fits without a
    # spill: brainpoolP512r1 6.0 -> 7.1 M/s (profiles/r2s_waves_per_simd.md; three waves on the 384 / 521-bit units: no gain / worse)
    # (round 5: the 521-bit units too -- left alone the window loop takes 258 / 269 registers, ONE wave per SIMD; pinned at two it compiles to
    # 212 / 224 without a spill)
    jobs += [("ecamd_g29_kernel.hip", f"ecamd_g29_{pb}.o", [f"-DG29_PB={pb}"] + (["-DG29_WAVES=2"] if pb in (511, 512, 521) else []))
             for pb in G29_SIZES]
    # secp521r1: plain residues on 18 limbs, 2^522 = 2 folded inside the product columns (ecamd_u29g.h:mul_m521p)
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_521m.o", ["-DG29_PB=521", "-DG29_M521P", "-DG29_WAVES=2"]))
    # the two nine-limb plain-residue units run best at three waves per SIMD (profiles/r2e_variants.md: 62.7 -> 63.5 and 68.0 -> 69.6 M/s)
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_255c.o", ["-DG29_PB=255", "-DG29_P25519", "-DG29_WAVES=3"]))
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_256k.o", ["-DG29_PB=256", "-DG29_K256", "-DG29_WAVES=3"]))
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_448g.o", ["-DG29_PB=448", "-DG29_P448"]))
    # secp384r1: Montgomery reduction on the four signed digits of p + 1 (ecamd_u29g.h:p384s_reduction)
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_384n.o", ["-DG29_PB=384", "-DG29_P384S"]))
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_224s.o", ["-DG29_PB=224", "-DG29_P224S"]))
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_192s.o", ["-DG29_PB=192", "-DG29_P192S"]))
    jobs.append(("ecamd_g29_kernel.hip", "ecamd_g29_dispatch.o", ["-DG29_DISPATCH"]))
    return jobs


def build(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in DEPS]
    hdeps = [os.path.join(CSRC, d) for d in HOST_DEPS]
    todo, objs = [], []
    for src, obj, extra in _jobs():
        spath, opath = os.path.join(CSRC, src), os.path.join(LIBDIR, obj)
        objs.append(opath)
        cmd = [HIPCC] + FLAGS + extra + ["-x", "hip", "-c", spath, "-o", opath]
        # an object is also stale when it was built with other flags (its command line is kept beside it)
        try:
            same_cmd = open(opath + ".cmd").read() == " ".join(cmd)
        except OSError:
            same_cmd = False
        if force or not same_cmd or _stale(opath, [spath] + deps + (hdeps if src.endswith(".cpp") else [])):
            todo.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        if "-c" in cmd:
            with open(cmd[-1] + ".cmd", "w") as f:
                f.write(" ".join(cmd))

    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(run, todo))
    if force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
