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
SOURCES = ["ecamd_kernels.hip", "ecamd_p256_kernel.hip", "ecamd_host.cpp"]
DEPS = ["ecamd_field.cuh", "ecamd_point.cuh", "ecamd_u29.cuh", "ecamd_p256.cuh", "ecamd_internal.h",
        "ecamd_curve_table.inc",
        os.path.join("..", "..", "include", "libecc_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-bitwise-instead-of-logical", "-DU29_ASM_MAD"]


def _stale(target, inputs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(i) > t for i in inputs)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in DEPS]
    objs = []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        if force or _stale(obj, [spath] + deps):
            cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", spath, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
