"""libecc_amd -- MI355X (gfx950) batched short-Weierstrass scalar multiplication behind
libecc's API surface.

The product is the C-ABI shared library `libecc_amd/lib/libecc_amd.so` (header:
`include/libecc_amd.h`, sources: `libecc_amd/csrc/`).  This Python package is a thin ctypes
binding used by the tests and the benchmark; it contains no arithmetic and NO CPU fallback:
if the HIP library is missing or there is no GPU, calls raise.
"""
from .api import (Context, Curve, Multi, MultiCurve, EcamdError, lib_path, load_library, ECAMD_OK, ECAMD_ERR, ECAMD_INF,
                  FP_MUL_MONTY, FP_ADD, FP_SUB, FP_MUL, FP_INV, EXPORTED_SYMBOLS)

__all__ = ["Context", "Curve", "Multi", "MultiCurve", "EcamdError", "lib_path", "load_library", "ECAMD_OK", "ECAMD_ERR",
           "ECAMD_INF", "FP_MUL_MONTY", "FP_ADD", "FP_SUB", "FP_MUL", "FP_INV", "EXPORTED_SYMBOLS"]
