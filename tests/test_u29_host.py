"""CPU tests of the radix-2^29 lazy-reduction field / Jacobian code (libecc_amd/csrc/ecamd_u29.h,
ecamd_p256.h): the product headers are compiled for the host (tests/u29_host_shim.cpp, g++) and
driven against Python integers, including operands sitting at the extreme of their declared bound
classes (the compile-time bound tracking must make those overflow-free)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import u29_consts as K  # noqa: E402

W, MASK, p, R = K.W, K.MASK, K.p, K.R
Rinv = pow(R, p - 2, p)
b = K.b
BUILD = os.path.join(ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "u29_host_shim.cpp")])
    return C.CDLL(so)


def limbs(x, n=9):
    d = [(x >> (W * i)) & MASK for i in range(n - 1)]
    d.append(x >> (W * (n - 1)))
    return d


def val(l):
    return sum(int(v) << (W * i) for i, v in enumerate(l))


def arr(l):
    assert all(0 <= v < 2**32 for v in l)
    return (C.c_uint32 * len(l))(*l)


def call(fn, *ins, n_out=9):
    out = (C.c_uint32 * n_out)()
    r = fn(*[arr(x) for x in ins], out)
    return list(out), r


def loose(rng, v, lb, tb):
    """a representation of integer v with random 'looseness': limbs up to lb (top up to tb)"""
    l = limbs(v)
    for i in range(8):
        # move k units of 2^29 from limb i+1 down into limb i
        room = (lb - l[i]) >> W
        k = min(room, l[i + 1], int(rng.integers(0, 8)))
        l[i] += k << W
        l[i + 1] -= k
    assert val(l) == v and max(l[:8]) <= lb and l[8] <= tb
    return l


def test_constants(lib):
    out, _ = call(lib.t_consts, n_out=49)
    assert out[0:9] == limbs(p)
    assert out[9:18] == limbs((1 << 256) % p)
    assert out[18:27] == limbs(R * R % p)
    assert out[27:36] == limbs(R % p)
    assert out[36:45] == limbs(b * R % p)
    q = limbs(p + 1)
    assert q[:3] == [0, 0, 0] and q[4] == q[5] == 0 and out[45:49] == [q[3], q[6], q[7], q[8]]
    K.inv_chain()


def test_mul_sqr_random_and_extreme(lib):
    rng = np.random.default_rng(21)
    cases = []
    for _ in range(300):
        cases.append((int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**62)) ** 4 % (2 * p),
                      int.from_bytes(rng.bytes(40), "big") % (2 * p)))
    cases += [(0, 0), (2 * p - 1, 2 * p - 1), (p, p), (p - 1, 1), (2**256 - 1, 2**256 - 1)]
    for x, y in cases:
        out, _ = call(lib.t_mul, limbs(x), limbs(y))
        assert val(out) % p == x * y * Rinv % p
        assert val(out) < 2 * p and max(out[:8]) <= MASK
        out, _ = call(lib.t_sqr, limbs(x))
        assert val(out) % p == x * x * Rinv % p and val(out) < 2 * p
    # loose class: every limb at 2^30, top limb at its bound -- worst case for the column accumulator
    worst = [1 << 30] * 8 + [6 << 24]
    out, _ = call(lib.t_mul_loose, worst, worst)
    assert val(out) % p == val(worst) ** 2 * Rinv % p
    for _ in range(200):
        v1 = int.from_bytes(rng.bytes(40), "big") % (5 * p)
        v2 = int.from_bytes(rng.bytes(40), "big") % (5 * p)
        l1, l2 = loose(rng, v1, 1 << 30, 6 << 24), loose(rng, v2, 1 << 30, 6 << 24)
        out, _ = call(lib.t_mul_loose, l1, l2)
        assert val(out) % p == v1 * v2 * Rinv % p and max(out[:8]) <= MASK


def test_fold_and_canonical(lib):
    rng = np.random.default_rng(22)
    for _ in range(300):
        v = int.from_bytes(rng.bytes(40), "big") % (6 * p - (1 << 240))
        l = loose(rng, v, 0xfffffffe, 6 << 24)
        out, _ = call(lib.t_fold, l)
        assert val(out) % p == v % p
        assert val(out) * 16 < 17 * p and max(out[:8]) <= MASK + 16
    for v in (0, 1, p - 1, p, p + 1, 2 * p - 1, int.from_bytes(rng.bytes(32), "big") % (2 * p)):
        words = (C.c_uint32 * 8)()
        lib.t_canonical_words(arr(limbs(v)), words)
        assert sum(int(w) << (32 * i) for i, w in enumerate(words)) == v % p
        back, _ = call(lib.t_from_words, [((v % p) >> (32 * i)) & 0xffffffff for i in range(8)])
        assert back == limbs(v % p)
    for k in range(4):
        assert lib.t_is_zero(arr(limbs(k * p))) == 1
        assert lib.t_is_zero(arr(limbs(k * p + 1))) == 0
        if k:
            assert lib.t_is_zero(arr(limbs(k * p - 1))) == 0


def test_inversion_chain(lib):
    rng = np.random.default_rng(23)
    for _ in range(5):
        x = int.from_bytes(rng.bytes(32), "big") % p or 1
        out, _ = call(lib.t_inv, limbs(x * R % p))
        assert val(out) % p == pow(x, p - 2, p) * R % p


# ---- Jacobian formulas vs independent affine arithmetic ----
def aff_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = (3 * P[0] * P[0] - 3) * pow(2 * P[1], p - 2, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], p - 2, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return (x, (lam * (P[0] - x) - P[1]) % p)


def aff_mul(k, P):
    Rr = None
    while k:
        if k & 1:
            Rr = aff_add(Rr, P)
        P = aff_add(P, P)
        k >>= 1
    return Rr


G = (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
     0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)


def jac_of(rng, P, xmul, ymul, zmul, classes):
    """Montgomery-domain Jacobian representation of affine P with random Z and non-canonical values
    (xmul etc. multiples of p added) and loose limbs within the (lb, tb) class bounds"""
    z = int.from_bytes(rng.bytes(32), "big") % p or 1
    X = (P[0] * z * z % p) * R % p + xmul * p
    Y = (P[1] * z * z * z % p) * R % p + ymul * p
    Z = z * R % p + zmul * p
    return [loose(rng, v, lb, tb) for v, (lb, tb) in zip((X, Y, Z), classes)]


def jac_to_aff(l27):
    X, Y, Z = (val(l27[0:9]) * Rinv % p, val(l27[9:18]) * Rinv % p, val(l27[18:27]) * Rinv % p)
    if Z == 0:
        return None
    zi = pow(Z, p - 2, p)
    return (X * zi * zi % p, Y * zi * zi * zi % p)


ACC = [(MASK + 16, (1 << 24) + 16), (MASK + 8, (6 << 24) + 8), (2 * MASK, 4 << 24)]
TBL = [(MASK + 16, (1 << 24) + 16), (MASK + 16, 4 << 24), (2 * MASK, 4 << 24)]


def test_jacobian_dbl_add(lib):
    rng = np.random.default_rng(24)
    for it in range(40):
        k1, k2 = int(rng.integers(1, 2**40)), int(rng.integers(1, 2**40))
        P, Q = aff_mul(k1, G), aff_mul(k2, G)
        # accumulator class: X < 17/16 p, Y < 6p, Z < 3.5p
        Pj = jac_of(rng, P, 0, int(rng.integers(0, 5)), int(rng.integers(0, 3)), ACC)
        flat = Pj[0] + Pj[1] + Pj[2]
        out, _ = call(lib.t_dbl, flat, n_out=27)
        assert jac_to_aff(out) == aff_add(P, P)
        assert max(out[0:8]) <= MASK + 16 and max(out[9:17]) <= MASK + 8 and max(out[18:26]) <= 2 * MASK
        assert val(out[0:9]) * 16 < 17 * p and val(out[9:18]) < 6 * p and val(out[18:27]) * 2 < 7 * p
        Qj = jac_of(rng, Q, 0, int(rng.integers(0, 3)), int(rng.integers(0, 3)), TBL)
        out, hz = call(lib.t_add, flat, Qj[0] + Qj[1] + Qj[2], n_out=27)
        assert hz == 0 and jac_to_aff(out) == aff_add(P, Q)
        assert val(out[0:9]) * 16 < 17 * p and val(out[9:18]) < 6 * p and val(out[18:27]) * 2 < 7 * p
    # exceptional pairs are detected: P + P and P + (-P) give h_is_zero
    P = aff_mul(12345, G)
    Pj = jac_of(rng, P, 0, 2, 1, ACC)
    Qj = jac_of(rng, P, 0, 2, 2, TBL)
    _, hz = call(lib.t_add, Pj[0] + Pj[1] + Pj[2], Qj[0] + Qj[1] + Qj[2], n_out=27)
    assert hz == 1
    Qn = jac_of(rng, (P[0], p - P[1]), 0, 2, 1, TBL)
    out, hz = call(lib.t_add, Pj[0] + Pj[1] + Pj[2], Qn[0] + Qn[1] + Qn[2], n_out=27)
    assert hz == 1 and jac_to_aff(out) is None


def test_mixed_addition(lib):
    rng = np.random.default_rng(26)
    lib.t_madd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for it in range(40):
        P, Q = aff_mul(int(rng.integers(1, 2**40)), G), aff_mul(int(rng.integers(1, 2**40)), G)
        Pj = jac_of(rng, P, 0, int(rng.integers(0, 5)), int(rng.integers(0, 3)), ACC)
        flat = Pj[0] + Pj[1] + Pj[2]
        qa = limbs(Q[0] * R % p) + limbs(Q[1] * R % p)
        for negate in (0, 1):
            out = (C.c_uint32 * 27)()
            hz = lib.t_madd(arr(flat), arr(qa), out, negate)
            Qs = (Q[0], (p - Q[1]) % p) if negate else Q
            assert hz == 0 and jac_to_aff(list(out)) == aff_add(P, Qs)
            o = list(out)
            assert val(o[0:9]) * 16 < 17 * p and val(o[9:18]) < 6 * p and val(o[18:27]) * 2 < 7 * p
    P = aff_mul(777, G)
    Pj = jac_of(rng, P, 0, 3, 2, ACC)
    qa = limbs(P[0] * R % p) + limbs(P[1] * R % p)
    out = (C.c_uint32 * 27)()
    assert lib.t_madd(arr(Pj[0] + Pj[1] + Pj[2]), arr(qa), out, 0) == 1
    assert lib.t_madd(arr(Pj[0] + Pj[1] + Pj[2]), arr(qa), out, 1) == 1 and jac_to_aff(list(out)) is None


def test_jacobian_chain_stays_in_bounds(lib):
    """200 consecutive doublings/additions through the loop-carried classes"""
    rng = np.random.default_rng(25)
    P = G
    cur = jac_of(rng, P, 0, 5, 2, ACC)
    flat = cur[0] + cur[1] + cur[2]
    aff = P
    for it in range(200):
        if it % 5 == 4:
            Q = aff_mul(int(rng.integers(1, 2**30)), G)
            Qj = jac_of(rng, Q, 0, 2, 2, TBL)
            flat, hz = call(lib.t_add, flat, Qj[0] + Qj[1] + Qj[2], n_out=27)
            assert hz == 0
            aff = aff_add(aff, Q)
        else:
            flat, _ = call(lib.t_dbl, flat, n_out=27)
            aff = aff_add(aff, aff)
        assert jac_to_aff(flat) == aff
        assert val(flat[0:9]) * 16 < 17 * p and val(flat[9:18]) < 6 * p and val(flat[18:27]) * 2 < 7 * p
