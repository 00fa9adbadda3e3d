"""CPU test of libecc_amd/csrc/ecamd_randmod.h (nn_get_random_mod given its random bytes: the little-endian integer of 2 * qlen bytes
modulo q - 1, plus one): host build of the product header against Python integers, on the group orders of five curves, random and edge
byte strings.  That the formula IS the reference's is pinned in tests/test_oracle.py::test_random_mod_vs_reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracles import CURVES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "randmod_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "randmod_host_shim.cpp")])
    return C.CDLL(so)


def randmod_cases(q, rng, nrand=200):
    """byte strings of 2 * qlen bytes: edge values around multiples of q - 1, all-zero, all-ones, sparse, random"""
    ql = (q.bit_length() + 7) // 8
    top = 1 << (16 * ql)
    vals = [0, 1, 2, q - 2, q - 1, q, q + 1, 2 * (q - 1) - 1, 2 * (q - 1), 2 * (q - 1) + 1, top - 1, top - 2, top // 2, top // 2 - 1,
            (q - 1) * (q - 1), (q - 1) * (q - 1) - 1, (q - 1) * (q - 1) + 1, (q - 1) << (8 * ql), ((q - 1) << (8 * ql)) - 1, 1 << (8 * ql), (1 << (8 * ql)) - 1]
    vals += [((top // (q - 1)) * (q - 1) + d) % top for d in (-1, 0, 1)]
    vals = [v % top for v in vals]
    vals += [int.from_bytes(rng.bytes(2 * ql), "little") for _ in range(nrand)]
    vals += [int.from_bytes(rng.bytes(k), "little") << (8 * s) for k in (1, 5, ql) for s in (0, ql, 2 * ql - k)]
    return ql, [v % top for v in vals]


@pytest.mark.parametrize("curve", ["SECP192R1", "SECP224R1", "SECP256R1", "SECP384R1", "SECP521R1", "WEI25519"])
def test_randmod_matches_python(lib, curve):
    q = CURVES[curve]["q"]
    nw = (q.bit_length() + 31) // 32
    rng = np.random.default_rng(q % 1000)
    ql, vals = randmod_cases(q, rng)
    qw = (C.c_uint32 * nw)(*[(q >> (32 * i)) & 0xffffffff for i in range(nw)])
    out = (C.c_uint32 * nw)()
    for v in vals:
        raw = v.to_bytes(2 * ql, "little")
        assert lib.randmod_host(nw, out, raw, 2 * ql, qw) == 0
        got = sum(out[i] << (32 * i) for i in range(nw))
        assert got == v % (q - 1) + 1, (curve, hex(v))
        assert 1 <= got <= q - 1
