"""CPU tests of the generic radix-2^29 field / Jacobian code (ecamd_u29g.h, ecamd_jacg.h) for
every field size libecc's curves use: host build of the product headers (tests/u29g_host_shim.cpp)
against Python integers, with operands spread over their whole bound class."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import g29_consts as G  # noqa: E402
from oracles import CURVES  # noqa: E402

W, MASK = G.W, G.MASK
BUILD = os.path.join(ROOT, "tests", "_build")
CASES = ["SECP192R1", "SECP224R1", "WEI25519", "BRAINPOOLP256R1", "SECP256K1", "SECP256R1", "BRAINPOOLP320R1",
         "SECP384R1", "WEI448", "GOST512", "BRAINPOOLP512R1", "SECP521R1"]


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29g_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


@pytest.fixture(scope="module")
def lib_m521():
    """the secp521r1 flavour (plain residues on 18 limbs, 2^522 = 2 folded inside the product columns) of the same headers"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29g_host_m521p.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DG29_M521P", "-DSHIM_ONLY_521", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


@pytest.fixture(scope="module")
def lib_p25519():
    """the 2^255 - 19 flavour (nine limbs, plain residues, pseudo-Mersenne folds) of the same headers"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29g_host_p25519.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DG29_P25519", "-DSHIM_ONLY_255", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


def val(l, w=W):
    return sum(int(v) << (w * i) for i, v in enumerate(l))


def arr(l):
    assert all(0 <= v < 2**32 for v in l), l
    return (C.c_uint32 * len(l))(*l)


class Field:
    def __init__(self, lib, curve, flavour=0, iso_u=None):
        c = CURVES[curve]
        self.p, self.a, self.b = c["p"], c["a"], c["b"]
        self.pb = self.p.bit_length()
        img, self.nl = G.image(self.p, self.a, self.b, flavour, iso_u)
        self.W = G.width(flavour)          # 28 on the Goldilocks unit, 29 everywhere else
        self.MASK = (1 << self.W) - 1
        self.k = arr(img)
        self.lib = lib
        info = (C.c_uint32 * 8)()
        getattr(lib, f"g_info_{self.pb}")(info)
        assert info[0] == self.nl and info[5] == 4 * len(img), (list(info), len(img))
        self.head, self.va, self.fa_lb, self.fa_tb = info[1], info[4], info[6], info[7]
        self.R = 1 if flavour in (1, 2, 4, 5) else 1 << (self.W * self.nl)
        self.Rinv = pow(self.R, self.p - 2, self.p)

    def fn(self, name):
        return getattr(self.lib, f"g_{name}_{self.pb}")

    def val(self, l):
        return val(l, self.W)

    def digits(self, v):
        return G.digits(v, self.nl, self.W)

    def vmax(self, vb, tb):
        """largest multiple of p a class (value bound vb, top limb bound tb) can hold"""
        return min(vb, ((tb - 1) << (self.W * (self.nl - 1))) // self.p)

    def loose(self, rng, v, lb, tb):
        l = self.digits(v)
        for i in range(self.nl - 1):
            room = (lb - l[i]) >> self.W
            k = min(room, l[i + 1], int(rng.integers(0, 8)))
            l[i] += k << self.W
            l[i + 1] -= k
        assert self.val(l) == v and max(l[:-1]) <= lb and l[-1] <= tb, (l, lb, tb)
        return l

    def fa(self, rng, residue, mult=None):
        """an FA-class representation of `residue` mod p: value residue + mult*p, loose limbs"""
        vmax = self.vmax(self.va, self.fa_tb)
        mult = int(rng.integers(0, max(1, vmax - 1))) if mult is None else mult
        return self.loose(rng, residue % self.p + mult * self.p, self.fa_lb, self.fa_tb)

    def call(self, name, *ins, n_out=None):
        out = (C.c_uint32 * (n_out or self.nl))()
        r = self.fn(name)(self.k, *[arr(x) for x in ins], out)
        return list(out), r


@pytest.mark.parametrize("curve", CASES)
def test_field_ops(lib, curve, flavour=0):
    rng = np.random.default_rng(41)
    f = Field(lib, curve, flavour)
    p = f.p
    for it in range(60):
        x = int.from_bytes(rng.bytes(80), "big") % p
        y = int.from_bytes(rng.bytes(80), "big") % p
        lx, ly = f.fa(rng, x), f.fa(rng, y)
        if it == 0:  # extreme: largest value the class allows, every low limb at its bound
            vmax = max(0, f.vmax(f.va, f.fa_tb) - 1)
            lx = f.fa(rng, x, vmax)
            ly = f.fa(rng, y, vmax)
        if it == 1 and flavour in (1, 2, 4, 5):   # every limb at the class bound (the value is then what it is: only the residue matters)
            lx = [f.fa_lb] * (f.nl - 1) + [f.fa_tb]
            ly = [f.fa_lb] * (f.nl - 1) + [f.fa_tb]
            x, y = f.val(lx) % p, f.val(ly) % p
        out = (C.c_uint32 * f.nl)()
        f.fn("mul")(f.k, arr(lx), arr(ly), out, 0)
        assert f.val(out) % p == f.val(lx) * f.val(ly) * f.Rinv % p
        # exact low digits -- on the Goldilocks unit limbs 1 and 9 keep the high parts of the two wrap-around carries (< 2^10)
        slack = [(1 << 10) if ((flavour == 5 and i in (1, 9)) or (flavour == 1 and i == 1)) else
                 ((1 << 17) if (flavour == 2 and i == 1) else ((1 << 15) if (flavour == 4 and i == 2) else 0)) for i in range(f.nl)]
        assert f.val(out) < 2 * p + (f.val(lx) * f.val(ly) >> (f.W * f.nl) if flavour not in (1, 5) else 0)
        assert all(v <= f.MASK + sl for v, sl in zip(list(out)[:-1], slack))
        f.fn("mul")(f.k, arr(lx), arr(lx), out, 1)
        assert f.val(out) % p == f.val(lx) ** 2 * f.Rinv % p
        assert all(v <= f.MASK + sl for v, sl in zip(list(out)[:-1], slack)) and (flavour not in (1, 5) or f.val(out) < 2 * p)
    # canonical digits, negation, inversion
    for v in (0, 1, p - 1, p, p + 1, 2 * p - 1, int.from_bytes(rng.bytes(80), "big") % (2 * p)):
        d, _ = f.call("canon", f.digits(v))
        assert f.val(d) == v % p and max(d[:-1]) <= f.MASK
        n, _ = f.call("neg", f.digits(v))
        assert (f.val(n) + v) % p == 0 and max(n[:-1]) <= f.fa_lb and n[-1] <= f.fa_tb
    if flavour in (1, 2, 4, 5):
        # lazy representatives of a multiplication result: limbs 1 (and 9; secp256k1: 2) over their width, values around 2^|p|
        top = 1 << f.pb
        lazy = {2: 1 << 17, 4: 1 << 15}.get(flavour, 1 << 10)     # P25519_MULX / K256_MULX / P448_MULX
        topbits = f.pb - f.W * (f.nl - 1)                     # 23 for 2^255 - 19, 24 for secp256k1, 28 for the no-headroom flavours
        lazy_limbs = {5: (1, 9), 4: (2,)}.get(flavour, (1,))
        for v in (top - 1, top, top + 2**224, 2 * p - 1, p + 2**224 + 1, top + 5, top + 18, top + 19, top + 20, top + 2**32 + 976,
                  top + 2**32 + 977, top + 2**32 + 978):
            v = min(v, 2 * p - 1)
            l = f.digits(v)
            for i in lazy_limbs:
                k = min(l[i + 1], 3)
                l[i] += k << f.W
                l[i + 1] -= k
            if max(l[:-1]) <= f.MASK + lazy and l[-1] < (1 << topbits):
                d, _ = f.call("canon", l)
                assert f.val(d) == v % p and max(d) <= f.MASK, hex(v)
        # the largest member of the class: every limb at its mask, the lazy ones at mask + the slack (value above 2^|p|)
        for extra in (lazy, 1, 0):
            l = [f.MASK] * (f.nl - 1) + [(1 << topbits) - 1]
            for i in lazy_limbs:
                l[i] += extra
            d, _ = f.call("canon", l)
            assert f.val(d) == f.val(l) % p and max(d) <= f.MASK
    x = int.from_bytes(rng.bytes(80), "big") % p or 1
    iv, _ = f.call("inv", f.digits(x * f.R % p))
    assert f.val(iv) % p == pow(x, p - 2, p) * f.R % p


def aff_add(P, Q, a, p):
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = (3 * P[0] * P[0] + a) * pow(2 * P[1], p - 2, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], p - 2, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return (x, (lam * (P[0] - x) - P[1]) % p)


@pytest.mark.parametrize("curve", CASES)
def test_jacobian(lib, curve, flavour=0):
    rng = np.random.default_rng(42)
    f = Field(lib, curve, flavour)
    c = CURVES[curve]
    p, a = f.p, f.a
    G0 = (c["gx"], c["gy"])

    def jac(P):
        z = int.from_bytes(rng.bytes(80), "big") % p or 1
        X, Y, Z = P[0] * z * z % p * f.R % p, P[1] * z * z * z % p * f.R % p, z * f.R % p
        return f.fa(rng, X) + f.fa(rng, Y) + f.fa(rng, Z)

    def aff(l):
        nl = f.nl
        X, Y, Z = (f.val(l[0:nl]) * f.Rinv % p, f.val(l[nl:2 * nl]) * f.Rinv % p, f.val(l[2 * nl:3 * nl]) * f.Rinv % p)
        if Z == 0:
            return None
        zi = pow(Z, p - 2, p)
        return (X * zi * zi % p, Y * zi * zi * zi % p)

    def in_class(l):
        nl = f.nl
        for k in range(3):
            part = l[k * nl:(k + 1) * nl]
            assert max(part[:-1]) <= f.fa_lb and part[-1] <= f.fa_tb and f.val(part) < f.va * p

    P = G0
    Q = aff_add(G0, G0, a, p)
    cur = jac(P)
    acc = P
    for it in range(24):
        if it % 4 == 3:
            Q = aff_add(Q, G0, a, p)
            cur, hz = f.call("add", cur, jac(Q), n_out=3 * f.nl)
            assert hz == 0
            acc = aff_add(acc, Q, a, p)
        else:
            cur, _ = f.call("dbl", cur, n_out=3 * f.nl)
            acc = aff_add(acc, acc, a, p)
        assert aff(cur) == acc
        in_class(cur)
    # mixed addition and doubling on the tight accumulator class of the affine-table kernels (not for secp256k1's flavour,
    # which keeps the Jacobian table)
    it3 = (C.c_uint32 * 3)()
    getattr(lib, f"g_infot_{f.pb}")(it3)
    if it3[0]:
        vt, ft_lb, ft_tb = it3[0], it3[1], it3[2]

        def ft(residue):
            vmax = f.vmax(vt, ft_tb)
            mult = int(rng.integers(0, max(1, vmax - 1)))
            return f.loose(rng, residue % p + mult * p, ft_lb, ft_tb)

        def jac_t(P):
            z = int.from_bytes(rng.bytes(80), "big") % p or 1
            return ft(P[0] * z * z % p * f.R % p) + ft(P[1] * z * z * z % p * f.R % p) + ft(z * f.R % p)

        def in_class_t(l):
            for k in range(3):
                part = l[k * f.nl:(k + 1) * f.nl]
                assert max(part[:-1]) <= ft_lb and part[-1] <= ft_tb and f.val(part) < vt * p

        def affine_fa(P):
            return f.fa(rng, P[0] * f.R % p) + f.fa(rng, P[1] * f.R % p)
        cur, acc = jac_t(acc), acc
        for it in range(16):
            if it % 3 == 2:
                cur, _ = f.call("dblt", cur, n_out=3 * f.nl)
                acc = aff_add(acc, acc, a, p)
            else:
                Q = aff_add(Q, G0, a, p)
                T = Q if it % 2 else (Q[0], p - Q[1])
                cur, _ = f.call("madd", cur, affine_fa(T), n_out=3 * f.nl)
                acc = aff_add(acc, T, a, p)
            assert aff(cur) == acc
            in_class_t(cur)
            if it % 5 == 4:
                cur = jac_t(acc)      # a fresh loose representative of the class
        # the same x: Z3 = 0, and it stays 0 through later doublings and additions (the kernels test the final Z once)
        for T in (P, (P[0], p - P[1])):
            out, _ = f.call("madd", jac_t(P), affine_fa(T), n_out=3 * f.nl)
            assert aff(out) is None
            in_class_t(out)
            out, _ = f.call("dblt", out, n_out=3 * f.nl)
            assert aff(out) is None
            in_class_t(out)
            out, _ = f.call("madd", out, affine_fa(Q), n_out=3 * f.nl)
            assert aff(out) is None
            in_class_t(out)
    # exceptional pairs are flagged
    _, hz = f.call("add", jac(P), jac(P), n_out=3 * f.nl)
    assert hz == 1
    out, hz = f.call("add", jac(P), jac((P[0], p - P[1])), n_out=3 * f.nl)
    assert hz == 1 and aff(out) is None


def test_secp521r1_mersenne_flavour(lib_m521):
    test_field_ops(lib_m521, "SECP521R1", 1)
    test_jacobian(lib_m521, "SECP521R1", 1)


def _check_mul_word(lib, curve, flavour, cst, lazy, lazy_limbs):
    """mul_word (round 4: a24 e of the x-only ladders as NL MADs): residue, limb class of a multiplication result, value < 2p,
    on random operands and on operands with every limb at the top of its range"""
    rng = np.random.default_rng(44)
    f = Field(lib, curve, flavour)
    fn = getattr(lib, "g_mulword_%d" % f.pb)
    topmax = (1 << 29) - 1 if flavour == 5 else (1 << 32) - 1
    for it in range(40):
        if it == 0:
            l = [(1 << 32) - 1] * (f.nl - 1) + [topmax]
        elif it == 1:
            l = [0] * f.nl
        else:
            l = [int(rng.integers(0, 1 << 32)) for _ in range(f.nl - 1)] + [int(rng.integers(0, topmax + 1))]
            if it % 3 == 0:
                l = f.digits(int.from_bytes(rng.bytes(80), "big") % f.p)
        out = (C.c_uint32 * f.nl)()
        fn(arr(l), out)
        assert f.val(out) % f.p == f.val(l) * cst % f.p
        assert f.val(out) < 2 * f.p
        assert all(v <= f.MASK + (lazy if i in lazy_limbs else 0) for i, v in enumerate(list(out)[:-1]))
        assert out[f.nl - 1] < (1 << (f.pb - f.W * (f.nl - 1)))


def test_p25519_flavour(lib_p25519):
    test_field_ops(lib_p25519, "WEI25519", 2)
    test_jacobian(lib_p25519, "WEI25519", 2)
    _check_mul_word(lib_p25519, "WEI25519", 2, 121665, 1 << 17, (1,))


@pytest.fixture(scope="module")
def lib_n384():
    """secp384r1's flavour of the 384-bit unit: Montgomery reduction on the four signed digits of p + 1, signed column sums"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29g_host_n384.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DG29_P384S", "-DSHIM_ONLY_384", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


def _sparse_flavour_checks(lib, curve):
    test_field_ops(lib, curve)
    test_jacobian(lib, curve)
    # small products: the negative reduction terms outweigh a column's products, so column sums go negative and the carries must
    # be arithmetic
    rng = np.random.default_rng(384)
    f = Field(lib, curve)
    p = f.p
    out = (C.c_uint32 * f.nl)()
    for it in range(200):
        x = int.from_bytes(rng.bytes(48), "big") % p
        y = (0, 1, 2, 1 << 29, (1 << 29) - 1, 1 << 58, p - 1, int(rng.integers(0, 1 << 20)))[it % 8]
        lx, ly = f.digits(x), f.digits(y)
        if it % 3 == 0:
            lx, ly = ly, lx
        f.fn("mul")(f.k, arr(lx), arr(ly), out, 0)
        assert f.val(out) % p == x * y * f.Rinv % p and f.val(out) < 2 * p and max(list(out)[:-1]) <= f.MASK
        f.fn("mul")(f.k, arr(f.digits(y)), arr(f.digits(y)), out, 1)
        assert f.val(out) % p == y * y * f.Rinv % p and f.val(out) < 2 * p and max(list(out)[:-1]) <= f.MASK


def test_secp384r1_mpinv1_flavour(lib_n384):
    assert CURVES["SECP384R1"]["p"] == 2**384 - 2**128 - 2**96 + 2**32 - 1
    _sparse_flavour_checks(lib_n384, "SECP384R1")


def _sparse_lib(pb, flag):
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, f"u29g_host_s{pb}.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", flag, f"-DSHIM_ONLY_{pb}", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


def test_secp224r1_sparse_flavour():
    """p = 2^224 - 2^96 + 1 = +1 mod 2^29: negated quotient digits, "+ m_k" clears the column, m_k (p - 1) as two signed MADs"""
    assert CURVES["SECP224R1"]["p"] == 2**224 - 2**96 + 1
    _sparse_flavour_checks(_sparse_lib(224, "-DG29_P224S"), "SECP224R1")


def test_secp192r1_sparse_flavour():
    """p = 2^192 - 2^64 - 1 = -1 mod 2^29: m_k (p + 1) as two signed MADs"""
    assert CURVES["SECP192R1"]["p"] == 2**192 - 2**64 - 1
    _sparse_flavour_checks(_sparse_lib(192, "-DG29_P192S"), "SECP192R1")


@pytest.fixture(scope="module")
def lib_k256():
    """the secp256k1 flavour (nine limbs, plain residues, folds with 2^256 = 2^32 + 977) of the 256-bit unit"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29g_host_k256.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DG29_K256", "-DSHIM_ONLY_256", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


def test_secp256k1_flavour(lib_k256):
    assert CURVES["SECP256K1"]["p"] == 2**256 - 2**32 - 977
    test_field_ops(lib_k256, "SECP256K1", 4)
    test_jacobian(lib_k256, "SECP256K1", 4)


@pytest.fixture(scope="module")
def lib_p448():
    """the Goldilocks flavour (16 limbs, plain residues, folds with 2^448 = 2^224 + 1) of the 448-bit unit"""
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "u29g_host_p448.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DG29_P448", "-DSHIM_ONLY_448", "-o", so,
                           os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
    return C.CDLL(so)


def test_p448_flavour(lib_p448):
    assert CURVES["WEI448"]["p"] == 2**448 - 2**224 - 1
    test_field_ops(lib_p448, "WEI448", 5)
    test_jacobian(lib_p448, "WEI448", 5)
    _check_mul_word(lib_p448, "WEI448", 5, 39081, 1 << 10, (1, 9))


def test_mad_counts_match_the_work_model():
    """bench.py prices a kernel by the MADs its field operations issue (`field_mads`: per multiplication, per squaring).  The host
    build of the same headers counts the MADs it really executes (-DECAMD_COUNT_MADS: every chain and every single MAD bumps a
    counter): the two must agree for every flavour, or the roofline fractions of the bench lines are wrong."""
    import importlib.util
    import types
    for name in ("torch", "libecc_amd"):          # bench.py imports them at module level; the model itself is pure Python
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    spec = importlib.util.spec_from_file_location("bench_for_model", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.makedirs(BUILD, exist_ok=True)
    rng = np.random.default_rng(7)
    cases = [("SECP192R1", 0, ["-DG29_P192S", "-DSHIM_ONLY_192"]), ("SECP224R1", 0, ["-DG29_P224S", "-DSHIM_ONLY_224"]), ("BRAINPOOLP192R1", 0, []),
             ("BRAINPOOLP224R1", 0, []), ("BRAINPOOLP256R1", 0, []), ("BRAINPOOLP320R1", 0, []), ("BRAINPOOLP384R1", 0, []),
             ("BRAINPOOLP512R1", 0, []), ("SECP384R1", 0, ["-DG29_P384S", "-DSHIM_ONLY_384"]), ("SECP521R1", 1, ["-DG29_M521P", "-DSHIM_ONLY_521"]),
             ("WEI25519", 2, ["-DG29_P25519", "-DSHIM_ONLY_255"]), ("SECP256K1", 4, ["-DG29_K256", "-DSHIM_ONLY_256"]),
             ("WEI448", 5, ["-DG29_P448", "-DSHIM_ONLY_448"])]
    libs = {}
    for curve, flavour, flags in cases:
        key = tuple(flags)
        if key not in libs:
            so = os.path.join(BUILD, "u29g_count_%s.so" % ("_".join(f[2:] for f in flags) or "dense"))
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DECAMD_COUNT_MADS"] + flags +
                                  ["-o", so, os.path.join(ROOT, "tests", "u29g_host_shim.cpp")])
            libs[key] = C.CDLL(so)
        lib = libs[key]
        lib.g_madcount_get.restype = C.c_uint64
        f = Field(lib, curve, flavour)
        nl, M, S, _ = bench.field_mads(f.p)
        assert nl == f.nl, (curve, nl, f.nl)
        x, y = (int.from_bytes(rng.bytes(80), "big") % f.p for _ in range(2))
        out = (C.c_uint32 * f.nl)()
        lib.g_madcount_reset()
        f.fn("mul")(f.k, arr(f.digits(x)), arr(f.digits(y)), out, 0)
        got_m = lib.g_madcount_get()
        lib.g_madcount_reset()
        f.fn("mul")(f.k, arr(f.digits(x)), arr(f.digits(x)), out, 1)
        got_s = lib.g_madcount_get()
        assert (got_m, got_s) == (M, S), (curve, "counted", got_m, got_s, "model", M, S)


def _rcb_add_py(P, Q, a, b, p):
    """RCB Algorithm 1 (generic a) over Python integers: the polynomials of ecamd_point.h:pt_add"""
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    b3 = 3 * b
    t0, t1, t2 = X1 * X2 % p, Y1 * Y2 % p, Z1 * Z2 % p
    t3 = ((X1 + Y1) * (X2 + Y2) - t0 - t1) % p
    t4 = ((X1 + Z1) * (X2 + Z2) - t0 - t2) % p
    t5 = ((Y1 + Z1) * (Y2 + Z2) - t1 - t2) % p
    z3 = (b3 * t2 + a * t4) % p
    x3, z3 = (t1 - z3) % p, (t1 + z3) % p
    y3 = x3 * z3 % p
    t1 = (3 * t0 + a * t2) % p
    t4 = (b3 * t4 + a * (t0 - a * t2)) % p
    return ((t3 * x3 - t5 * t4) % p, (y3 + t1 * t4) % p, (t5 * z3 + t3 * t1) % p)


def _rcb_dbl_py(P, a, b, p):
    """RCB Algorithm 3 (generic a): the polynomials of ecamd_point.h:pt_dbl"""
    X, Y, Z = P
    b3 = 3 * b
    t0, t1, t2 = X * X % p, Y * Y % p, Z * Z % p
    t3, z3 = 2 * X * Y % p, 2 * X * Z % p
    y3 = (a * z3 + b3 * t2) % p
    x3, y3 = (t1 - y3) % p, (t1 + y3) % p
    y3 = x3 * y3 % p
    x3 = t3 * x3 % p
    t3 = (a * (t0 - a * t2) + b3 * z3) % p
    t0 = (3 * t0 + a * t2) % p
    t2 = 2 * Y * Z % p
    return ((x3 - t2 * t3) % p, (y3 + t0 * t3) % p, 4 * t2 * t1 % p)


@pytest.mark.parametrize("curve,flavour,fixture", [("WEI25519", 2, "lib_p25519"), ("WEI448", 5, "lib_p448")])
def test_complete_formulas_on_the_radix29_types(curve, flavour, fixture, request):
    """ecamd_rcbg.h (the tail of the EdDSA verifications on the 2^255 - 19 and Goldilocks units): the same FIELD ELEMENTS as the
    Renes-Costello-Batina polynomials over Python integers -- not just the same projective point -- for random pairs, P + P,
    P + (-P), infinity on either side, and the exceptional pairs of these even-order curves (P and P + T with T of order two:
    the result is (0 : 0 : 0), which the callers reject as libecc's prj_pt_add does)."""
    lib = request.getfixturevalue(fixture)
    rng = np.random.default_rng(99)
    f = Field(lib, curve, flavour)
    c = CURVES[curve]
    p, a, b = f.p, f.a, f.b
    G0 = (c["gx"], c["gy"])

    def proj(P):
        if P is None:
            return (0, 1, 0)
        z = int.from_bytes(rng.bytes(80), "big") % p or 1
        return (P[0] * z % p, P[1] * z % p, z)

    def limbs(P):
        return f.fa(rng, P[0]) + f.fa(rng, P[1]) + f.fa(rng, P[2])

    def got(out):
        nl = f.nl
        return tuple(f.val(out[k * nl:(k + 1) * nl]) % p for k in range(3))

    def in_class(out):
        nl = f.nl
        for k in range(3):
            part = out[k * nl:(k + 1) * nl]
            assert max(part[:-1]) <= f.fa_lb and part[-1] <= f.fa_tb and f.val(part) < f.va * p

    # the point of order two: (x0, 0) with x0 a root of x^3 + a x + b (x0 = A / 3 on these two curves' Montgomery twins)
    A = {"WEI25519": 486662, "WEI448": 156326}[curve]
    x0 = next(x for x in (A * pow(3, p - 2, p) % p, (-A) * pow(3, p - 2, p) % p) if (x * x * x + a * x + b) % p == 0)
    T2 = (x0, 0)
    pts = [G0]
    for _ in range(6):
        pts.append(aff_add(pts[-1], G0, a, p))
    cases = []
    for i in range(1, 6):
        cases.append((pts[i], pts[i - 1]))                 # generic pairs
    cases += [(pts[2], pts[2]), (pts[3], (pts[3][0], p - pts[3][1])), (None, pts[1]), (pts[1], None), (None, None),
              (T2, T2), (pts[1], T2)]
    Q = aff_add(pts[4], T2, a, p)
    cases += [(pts[4], Q), (Q, pts[4])]                    # exceptional: the difference has order two
    fn = f.fn("rcb")
    for P, Q in cases:
        PP, QQ = proj(P), proj(Q)
        out = (C.c_uint32 * (3 * f.nl))()
        flags = fn(f.k, arr(limbs(PP)), arr(limbs(QQ)), out, 0)
        exp = _rcb_add_py(PP, QQ, a, b, p)
        assert got(list(out)) == exp, (curve, P, Q)
        assert flags == (1 if exp[1] == 0 else 0) | (2 if exp[2] == 0 else 0)
        in_class(list(out))
        if P is not None and Q is not None and P != Q and (P[0] != Q[0]) and aff_add(P, (Q[0], p - Q[1]), a, p) != T2:
            R = aff_add(P, Q, a, p)
            zi = pow(exp[2], p - 2, p)
            assert (exp[0] * zi % p, exp[1] * zi % p) == R
        out2 = (C.c_uint32 * (3 * f.nl))()
        fn(f.k, arr(limbs(PP)), arr(limbs(PP)), out2, 1)
        assert got(list(out2)) == _rcb_dbl_py(PP, a, b, p)
        in_class(list(out2))
    # the exceptional pairs do give (0 : 0 : 0)
    e = _rcb_add_py(proj(pts[4]), proj(aff_add(pts[4], T2, a, p)), a, b, p)
    assert e == (0, 0, 0)


def _iso_u(p, a):
    """u with a u^4 = -3 mod p (p = 3 mod 4), or None: the isomorphism ecamd_host.cpp:upload_g29 looks for (the brainpool r1 curves have one)"""
    t = (p - 3) * pow(a, -1, p) % p                  # u^4
    for s2 in (1, -1):
        r = pow(t, (p + 1) // 4, p)                   # a square root of t, if any
        if r * r % p != t:
            return None
        r = r * s2 % p
        u = pow(r, (p + 1) // 4, p)
        if u * u % p == r:
            return u
    return None


@pytest.mark.parametrize("curve,flavour,fixture,iso", [
    ("SECP192R1", 0, "lib", False), ("SECP256R1", 0, "lib", False), ("SECP256K1", 0, "lib", False), ("SECP256K1", 4, "lib_k256", False),
    ("BRAINPOOLP256R1", 0, "lib", False), ("BRAINPOOLP256R1", 0, "lib", True), ("BRAINPOOLP320R1", 0, "lib", True), ("SECP384R1", 0, "lib", False),
    ("SECP384R1", 3, "lib_n384", False), ("WEI448", 5, "lib_p448", False), ("BRAINPOOLP512R1", 0, "lib", True), ("SECP521R1", 0, "lib", False),
    ("SECP521R1", 1, "lib_m521", False)])
def test_lift_x_even(curve, flavour, fixture, iso, request):
    """jacg::lift_x_even -- BIP0340's lift_x as the Schnorr multi-scalar multiplication runs it on the device (k_msm_table_g, r_fmt 1): for
    p = 3 mod 4, x < p, y = (x^3 + a x + b)^((p + 1) / 4) with the square test, and the root that is EVEN ON THE ORIGINAL CURVE also when the
    unit computes on the isomorphic a = -3 image (the parity goes through the export factor ey); every flavour's own multiplication."""
    lib = request.getfixturevalue(fixture)
    c = CURVES[curve]
    p, a, b = c["p"], c["a"], c["b"]
    assert p % 4 == 3
    u = _iso_u(p, a) if iso else None
    assert (u is not None) == iso
    f = Field(lib, curve, flavour, u)
    rng = np.random.default_rng(31 + flavour)
    nw = (f.pb + 31) // 32
    fn = getattr(lib, f"g_liftx_{f.pb}")
    xs = [c["gx"], 0, 1, 2, p - 1, p - 2] + [int.from_bytes(rng.bytes(80), "big") % p for _ in range(60)]
    n_ok = 0
    for x in xs + [p, p + 1, (1 << (32 * nw)) - 1]:
        if x >= 1 << (32 * nw):
            continue
        xw = arr([(x >> (32 * k)) & 0xffffffff for k in range(nw)])
        out = (C.c_uint32 * (2 * f.nl))()
        ok = fn(f.k, xw, out)
        rhs = (x * x * x + a * x + b) % p
        y = pow(rhs, (p + 1) // 4, p)
        want = x < p and y * y % p == rhs
        assert bool(ok) == want, (curve, flavour, hex(x))
        if not want:
            continue
        n_ok += 1
        y = y if y % 2 == 0 else p - y
        xo, yo = f.val(out[:f.nl]), f.val(out[f.nl:])
        assert xo < 2 * p and yo < 2 * p and max(out[:f.nl - 1]) <= f.MASK + (1 << 18)     # a multiplication result of the unit
        uu = u if u is not None else 1
        assert xo * f.Rinv % p == x * uu * uu % p, (curve, flavour, hex(x))               # the unit's (Montgomery / plain) form of u^2 x
        assert yo * f.Rinv % p == y * uu**3 % p, (curve, flavour, hex(x))                 # ... of u^3 y_even
    assert 20 <= n_ok <= len(xs)
