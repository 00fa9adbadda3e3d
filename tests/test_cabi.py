"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import pytest

import libecc_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "libecc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ec(?:amd)?_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(libecc_amd.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = libecc_amd.load_library()
    for name in header_functions():
        assert hasattr(L, name), name


def test_no_torch_types_in_signatures():
    src = open(os.path.join(ROOT, "include", "libecc_amd.h")).read()
    assert "torch" not in src and "at::" not in src and "#include <hip" not in src


def test_fails_loudly_without_gpu():
    L = libecc_amd.load_library()
    if L.ecamd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(libecc_amd.EcamdError, match="no HIP device"):
        libecc_amd.Context(0)


def test_product_does_not_touch_the_oracle():
    """the product tree must not reference oracle/ (parity claims are void otherwise)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "libecc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".cuh", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "ecc_oracle" not in txt and "libecc_ref" not in txt, f
