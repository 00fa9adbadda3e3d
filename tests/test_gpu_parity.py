"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the committed
golden vectors and size-independent algebraic properties.  Bit-exact (integer/byte work)."""
import json
import os

import numpy as np
import pytest

from oracles import (CURVES, GOLDEN, Oracle, RefLib, clen, digest, have_ref, py_smul_bytes, qlen,
                     rfc6979_nonce)

pytestmark = pytest.mark.gpu

# random items of a full-size batch that are compared with the unmodified reference (>= 2^16)
FULL_PARITY_ITEMS = int(os.environ.get("ECAMD_TEST_PARITY_ITEMS", str(1 << 16)))

MAIN = ["SECP256R1", "SECP384R1", "SECP521R1", "WEI25519"]
WIDTHS = ["SECP192R1", "SECP224R1", "BRAINPOOLP320R1", "WEI448", "BRAINPOOLP512R1", "SECP256K1"]


def rand_bytes(rng, n):
    return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()


def edge_scalars(curve, slen):
    q, order = CURVES[curve]["q"], CURVES[curve]["order"]
    top = (1 << (8 * slen)) - 1
    vals = [0, 1, 2, 3, 15, 16, 17, q - 1, q, q + 1, order - 1, order, order + 1, 2 * q, top, top - 1,
            1 << (8 * slen - 1)]
    return b"".join((v & top).to_bytes(slen, "big") for v in vals)


@pytest.mark.parametrize("curve", MAIN + WIDTHS)
def test_fp_ops_vs_oracle(gpu_ctx, curve):
    """rows a6/a7/a13/a14: nn_mul_redc1 (limb-for-limb, reference radix), fp_add, fp_sub, fp_mul, fp_inv"""
    rng = np.random.default_rng(1)
    p = CURVES[curve]["p"]
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        n = 512
        a = [int.from_bytes(rand_bytes(rng, o.clen + 8), "big") % p for _ in range(n)]
        b = [int.from_bytes(rand_bytes(rng, o.clen + 8), "big") % p for _ in range(n)]
        edge = [0, 1, 2, p - 1, p - 2, (1 << (64 * o.nl)) % p, pow(2, 32 * cv.words, p)]
        a[:len(edge)] = edge
        b[:len(edge)] = list(reversed(edge))
        for op in range(5):
            if op == 4:
                aa = [x if x else 1 for x in a]
                assert cv.fp_op(op, aa, b) == o.fp_op(op, aa, b), f"op {op}"
            else:
                assert cv.fp_op(op, a, b) == o.fp_op(op, a, b), f"op {op}"
        # independent check against Python ints
        nl = o.nl
        rinv = pow(pow(2, 64 * nl, p), p - 2, p)
        assert cv.fp_op(0, a[:64], b[:64]) == [x * y * rinv % p for x, y in zip(a[:64], b[:64])]
    finally:
        cv.free()


@pytest.mark.parametrize("curve", MAIN + WIDTHS)
def test_scalar_mult_vs_oracle(gpu_ctx, curve):
    """row a19: fixed base and variable base, random + edge scalars, vs the restatement oracle"""
    rng = np.random.default_rng(2)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        slen = o.qlen
        sc = edge_scalars(curve, slen) + rand_bytes(rng, slen * 47)
        got = cv.scalar_mult(sc)
        exp = o.scalar_mult(sc)
        assert got[1] == exp[1]
        assert got[0] == exp[0]
        assert set(exp[1]) >= {0, 2}
        # variable base: use the valid outputs above as base points
        pts = b"".join(exp[0][i * 2 * o.clen:(i + 1) * 2 * o.clen] for i in range(len(exp[1])) if exp[1][i] == 0)
        npts = len(pts) // (2 * o.clen)
        sc2 = (edge_scalars(curve, slen) + rand_bytes(rng, slen * npts))[:slen * npts]
        got = cv.scalar_mult(sc2, pts)
        exp = o.scalar_mult(sc2, pts)
        assert got == exp
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP521R1", "WEI25519"])
def test_scalar_mult_rejections(gpu_ctx, curve):
    """off-curve points and coordinates >= p must be rejected exactly like prj_pt_import_from_aff_buf"""
    rng = np.random.default_rng(3)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        c = CURVES[curve]
        p, n = c["p"], o.clen
        g = c["gx"].to_bytes(n, "big") + c["gy"].to_bytes(n, "big")
        bad = [
            c["gx"].to_bytes(n, "big") + ((c["gy"] + 1) % p).to_bytes(n, "big"),  # off curve
            ((c["gx"] + 1) % p).to_bytes(n, "big") + c["gy"].to_bytes(n, "big"),
            bytes(n) + bytes(n),                                                  # (0,0)
            g,
        ]
        if (c["gx"] + p) < (1 << (8 * n)):
            bad.append((c["gx"] + p).to_bytes(n, "big") + c["gy"].to_bytes(n, "big"))  # x >= p, same residue
        if (c["gy"] + p) < (1 << (8 * n)):
            bad.append(c["gx"].to_bytes(n, "big") + (c["gy"] + p).to_bytes(n, "big"))
        bad.append(p.to_bytes(n, "big") + c["gy"].to_bytes(n, "big"))
        bad.append(b"\xff" * (2 * n))
        bad += [rand_bytes(rng, 2 * n) for _ in range(8)]
        pts = b"".join(bad)
        sc = rand_bytes(rng, o.qlen * len(bad))
        got = cv.scalar_mult(sc, pts)
        exp = o.scalar_mult(sc, pts)
        assert got == exp
        assert 1 in exp[1] and 0 in exp[1]
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "SECP521R1", "SECP192R1", "SECP224R1"])
def test_ecccdh_golden(gpu_ctx, curve):
    """the reference's own NIST ECC-CDH KATs: d*G == exp_our_pub_key, x(d*Q) == exp_shared_secret"""
    kats = [k for k in json.load(open(os.path.join(GOLDEN, "ecccdh_kats.json"))) if k["curve"] == curve]
    assert len(kats) == 25
    cv = gpu_ctx.curve(curve)
    try:
        n = cv.clen
        d = b"".join(bytes.fromhex(k["our_priv_key"]) for k in kats)
        slen = len(d) // len(kats)
        pub, st = cv.scalar_mult(d, None, slen)
        assert set(st) == {0}
        assert pub == b"".join(bytes.fromhex(k["exp_our_pub_key"]) for k in kats)
        peers = b"".join(bytes.fromhex(k["peer_pub_key"]) for k in kats)
        sh, st = cv.scalar_mult(d, peers, slen)
        assert set(st) == {0}
        xs = b"".join(sh[i * 2 * n:i * 2 * n + n] for i in range(len(kats)))
        assert xs == b"".join(bytes.fromhex(k["exp_shared_secret"]) for k in kats)
    finally:
        cv.free()


@pytest.mark.parametrize("curve", MAIN + ["BRAINPOOLP320R1"])
def test_point_add_dbl_vs_oracle(gpu_ctx, curve):
    """rows a17/a18: prj_pt_add / prj_pt_dbl incl. P+P, P+(-P)"""
    rng = np.random.default_rng(4)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        c = CURVES[curve]
        sc = rand_bytes(rng, o.qlen * 32)
        pts, st = o.scalar_mult(sc)
        n = o.clen
        P = [pts[i * 2 * n:(i + 1) * 2 * n] for i in range(32) if st[i] == 0]
        neg = [x[:n] + ((c["p"] - int.from_bytes(x[n:], "big")) % c["p"]).to_bytes(n, "big") for x in P]
        p1 = b"".join(P + P[:4] + P[:4])
        p2 = b"".join(P[1:] + P[:1] + P[:4] + neg[:4])
        assert cv.pt_add(p1, p2) == o.pt_add(p1, p2)
        assert cv.pt_add(p1) == o.pt_add(p1)
    finally:
        cv.free()


def test_fast_path_exceptional_pairs(gpu_ctx):
    """secp256r1 fast path (Jacobian, incomplete addition): scalars around the group order drive the
    accumulator onto +-(table entry) -- k = q - 2 hits the doubling case, k = q the inverse case --
    and must come back bit-exact through the complete-formula redo kernel."""
    curve = "SECP256R1"
    rng = np.random.default_rng(8)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        q = CURVES[curve]["q"]
        ks = [q + j for j in range(-40, 41)] + [(1 << 256) - 1 - j for j in range(8)] + list(range(0, 40))
        ks += [16 * j + d for j in (1, 2, 3) for d in (-8, -1, 1, 7, 8)] + [(q - 2) >> 4, ((q - 2) >> 4) + 1]
        sc = b"".join(k.to_bytes(32, "big") for k in ks)
        exp = o.scalar_mult(sc)
        assert cv.scalar_mult(sc) == exp
        assert 2 in exp[1]
        pts, st = o.scalar_mult(rand_bytes(rng, 32 * len(ks)))
        assert set(st) == {0}
        assert cv.scalar_mult(sc, pts) == o.scalar_mult(sc, pts)
        # short scalars through the fast path
        for slen in (1, 2, 5, 31):
            s2 = rand_bytes(rng, slen * 33) + b"\xff" * slen + b"\x00" * slen + b"\x88" * slen + b"\x77" * slen
            assert cv.scalar_mult(s2, None, slen) == o.scalar_mult(s2, None, slen), slen
    finally:
        cv.free()


def test_cofactor_curve_small_order_points(gpu_ctx):
    """WEI25519 (cofactor 8): inputs of even order drive the Jacobian fast path through doublings that
    reach infinity silently and additions with Z = 0; every such lane must be redone by the
    complete-formula kernel and agree with the oracle"""
    curve = "WEI25519"
    c = CURVES[curve]
    p, a, q, order = c["p"], c["a"], c["q"], c["order"]
    # the 2-torsion point of Curve25519 (u, v) = (0, 0) is (A/3, 0) on the Weierstrass model
    A = 486662
    x2 = A * pow(3, p - 2, p) % p
    assert (x2 ** 3 + a * x2 + c["b"]) % p == 0
    n = 32
    T2 = x2.to_bytes(n, "big") + bytes(n)
    from oracles import py_add
    GT = py_add((c["gx"], c["gy"]), (x2, 0), a, p)          # order 2q
    GT = GT[0].to_bytes(n, "big") + GT[1].to_bytes(n, "big")
    ks = [0, 1, 2, 3, 4, 8, q - 1, q, q + 1, 2 * q, 2 * q + 1, 4 * q, order - 1, order % (1 << 256), 16, 17, 255, 256]
    rng = np.random.default_rng(9)
    ks += [int.from_bytes(rand_bytes(rng, 32), "big") for _ in range(14)]
    sc = b"".join((k % (1 << 256)).to_bytes(32, "big") for k in ks)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        for P in (T2, GT):
            pts = P * len(ks)
            exp = o.scalar_mult(sc, pts, 32)
            assert cv.scalar_mult(sc, pts, 32) == exp
        # order 2: error for every scalar; order 2q: finite results and infinity at multiples of 2q
        assert set(o.scalar_mult(sc, T2 * len(ks), 32)[1]) == {1}
        st = o.scalar_mult(sc, GT * len(ks), 32)[1]
        assert 2 in st and 0 in st
        # prj_pt_add on an exceptional pair (difference of order exactly 2) is an error in the reference
        G1 = c["gx"].to_bytes(n, "big") + c["gy"].to_bytes(n, "big")
        assert cv.pt_add(G1 + T2 + GT, GT + T2 + T2) == o.pt_add(G1 + T2 + GT, GT + T2 + T2)
    finally:
        cv.free()


def test_linearity_large_batch(gpu_ctx):
    """size-independent property at a large batch: [a]P + [b]P == [a+b]P and [a]([b]G) == [ab mod q]G,
    plus a spot check of a random subset against the oracle and chunking across launches."""
    curve = "SECP256R1"
    rng = np.random.default_rng(5)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        q = CURVES[curve]["q"]
        n = 1 << 15
        gpu_ctx.set_max_chunk(5000)  # force several chunks with a ragged tail
        a = [int.from_bytes(rand_bytes(rng, 40), "big") % q for _ in range(n)]
        b = [int.from_bytes(rand_bytes(rng, 40), "big") % q for _ in range(n)]
        A = b"".join(x.to_bytes(32, "big") for x in a)
        B = b"".join(x.to_bytes(32, "big") for x in b)
        bG, st = cv.scalar_mult(B)
        assert set(st) == {0}
        abG, st = cv.scalar_mult(A, bG)
        assert set(st) == {0}
        AB = b"".join((x * y % q).to_bytes(32, "big") for x, y in zip(a, b))
        abG2, st = cv.scalar_mult(AB)
        assert abG == abG2
        aG, _ = cv.scalar_mult(A)
        S = b"".join(((x + y) % q).to_bytes(32, "big") for x, y in zip(a, b))
        sG, st2 = cv.scalar_mult(S)
        sumG, st3 = cv.pt_add(aG, bG)
        assert st2 == st3 and sG == sumG
        idx = rng.choice(n, size=64, replace=False)
        sub = b"".join(A[i * 32:(i + 1) * 32] for i in idx)
        subp = b"".join(bG[i * 64:(i + 1) * 64] for i in idx)
        exp, _ = o.scalar_mult(sub, subp)
        assert exp == b"".join(abG[i * 64:(i + 1) * 64] for i in idx)
    finally:
        gpu_ctx.set_max_chunk(1 << 20)
        cv.free()


@pytest.mark.parametrize("curve", ["BRAINPOOLP256R1", "WEI25519", "SECP256K1", "SECP224R1", "BRAINPOOLP320R1", "SECP384R1",
                                   "WEI448", "BRAINPOOLP512R1", "SECP521R1"])
def test_generic_fast_path_properties(gpu_ctx, curve):
    """generic radix-2^29 path: 2^13 random items per curve through [a]([b]G) == [a b mod q]G and
    [a]G + [b]G == [a + b]G, plus 96 random items against the oracle"""
    rng = np.random.default_rng(18)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        q = CURVES[curve]["q"]
        n, ql, pl = 1 << 13, o.qlen, 2 * o.clen
        a = [int.from_bytes(rand_bytes(rng, ql + 8), "big") % q for _ in range(n)]
        b = [int.from_bytes(rand_bytes(rng, ql + 8), "big") % q for _ in range(n)]
        A = b"".join(x.to_bytes(ql, "big") for x in a)
        B = b"".join(x.to_bytes(ql, "big") for x in b)
        bG, st = cv.scalar_mult(B)
        assert set(st) <= {0, 2}
        abG, st1 = cv.scalar_mult(A, bG)
        abG2, st2 = cv.scalar_mult(b"".join((x * y % q).to_bytes(ql, "big") for x, y in zip(a, b)))
        ok = [i for i in range(n) if st[i] == 0]
        assert all(st1[i] == st2[i] for i in ok)
        assert all(abG[i * pl:(i + 1) * pl] == abG2[i * pl:(i + 1) * pl] for i in ok)
        aG, _ = cv.scalar_mult(A)
        sG, st3 = cv.scalar_mult(b"".join(((x + y) % q).to_bytes(ql, "big") for x, y in zip(a, b)))
        sumG, st4 = cv.pt_add(aG, bG)
        assert st3 == st4 and sG == sumG
        idx = rng.choice(n, size=96, replace=False)
        exp, est = o.scalar_mult(b"".join(A[i * ql:(i + 1) * ql] for i in idx), b"".join(bG[i * pl:(i + 1) * pl] for i in idx))
        assert exp == b"".join(abG[i * pl:(i + 1) * pl] for i in idx)
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "SECP521R1"])
def test_full_batch_properties(gpu_ctx, curve):
    """BASELINE.json configs[1] and [2] at their full size (2^20 items in one call, 4 / 6 / 9 x 64-bit
    limb curves): every item is checked through [a]([b]G) == [a b mod q]G computed two ways on the GPU,
    and 2^16 random items + a 4096-item edge slice against the unmodified reference binary"""
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        q = CURVES[curve]["q"]
        n, ql, pl = 1 << 20, o.qlen, 2 * o.clen
        rng = np.random.default_rng(10)
        raw = rng.integers(0, 256, size=(2, n, ql + 8), dtype=np.uint8)
        a = [int.from_bytes(raw[0, i].tobytes(), "big") % q for i in range(n)]
        b = [int.from_bytes(raw[1, i].tobytes(), "big") % q for i in range(n)]
        A = b"".join(x.to_bytes(ql, "big") for x in a)
        B = b"".join(x.to_bytes(ql, "big") for x in b)
        AB = b"".join((x * y % q).to_bytes(ql, "big") for x, y in zip(a, b))
        bG, st = cv.scalar_mult(B)
        assert set(st) <= {0, 2}
        abG, st1 = cv.scalar_mult(A, bG)
        abG2, st2 = cv.scalar_mult(AB)
        if set(st) == {0}:
            assert st1 == st2 and abG == abG2
        else:
            ok = [i for i in range(n) if st[i] == 0]
            assert all(st1[i] == st2[i] and abG[i * pl:(i + 1) * pl] == abG2[i * pl:(i + 1) * pl] for i in ok)
        # >= 2^16 random items of the batch and the whole 4096-item edge slice, byte for byte
        # against the unmodified reference (prj_pt_mul + prj_pt_unique) on all host threads
        from bench import edge_slice, host_cores
        ns = FULL_PARITY_ITEMS if have_ref() else 2048
        idx = np.sort(rng.choice(n, size=ns, replace=False))
        sub = b"".join(A[i * ql:(i + 1) * ql] for i in idx)
        subp = b"".join(bG[i * pl:(i + 1) * pl] for i in idx)
        ref = RefLib(curve) if have_ref() else None
        smul = (lambda s_, p_: ref.scalar_mult(s_, p_, ql, nthreads=host_cores())) if ref else (lambda s_, p_: o.scalar_mult(s_, p_))
        exp, est = smul(sub, subp)
        assert exp == b"".join(abG[i * pl:(i + 1) * pl] for i in idx)
        assert est == bytes(st1[i] for i in idx)
        e_sc, e_pt = edge_slice(CURVES[curve], ql, o.clen, bG, 4096)
        got = cv.scalar_mult(e_sc, e_pt)
        exp = smul(e_sc, e_pt)
        assert got[1] == exp[1] and got[0] == exp[0]
        assert got[1].count(1) > 1000 and got[1].count(2) > 100 and got[1].count(0) > 1500
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "BRAINPOOLP256R1", "SECP384R1", "SECP521R1", "WEI25519"])
def test_long_and_short_scalars(gpu_ctx, curve):
    """scalar_len other than |q|: 1 byte ... the field size (fast paths, left-aligned recoding) and
    beyond it (blinded-size scalars, m >= q^2 branch of the reference: saturated complete-formula kernel)"""
    rng = np.random.default_rng(6)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        for slen in (1, 2, 3, 8, o.qlen - 1, o.qlen + 1, 4 * ((CURVES[curve]["p"].bit_length() + 31) // 32),
                     4 * ((CURVES[curve]["p"].bit_length() + 31) // 32) + 1, 2 * o.qlen + 8):
            sc = rand_bytes(rng, slen * 13) + b"\xff" * slen + b"\x00" * slen + b"\x88" * slen
            assert cv.scalar_mult(sc, None, slen) == o.scalar_mult(sc, None, slen), (curve, slen)
    finally:
        cv.free()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "SECP521R1", "WEI25519", "BRAINPOOLP256R1", "SECP256K1", "WEI448"])
def test_scalar_mult_vs_reference_binary(gpu_ctx, curve):
    """directly against the unmodified reference (prj_pt_mul + prj_pt_unique), no restatement in between: fixed base and
    variable base, edge and random scalars, 2048 items each way"""
    from bench import host_cores
    rng = np.random.default_rng(7)
    cv = gpu_ctx.curve(curve)
    r = RefLib(curve)
    nt = host_cores()
    try:
        sc = edge_scalars(curve, r.qlen) + rand_bytes(rng, r.qlen * 2031)
        got = cv.scalar_mult(sc)
        assert got == r.scalar_mult(sc, None, r.qlen, nthreads=nt)
        pl = 2 * r.clen
        pts = b"".join(got[0][i * pl:(i + 1) * pl] for i in range(len(got[1])) if got[1][i] == 0)
        npts = len(pts) // pl
        sc2 = (edge_scalars(curve, r.qlen) * 3 + rand_bytes(rng, r.qlen * npts))[:r.qlen * npts]
        assert cv.scalar_mult(sc2, pts) == r.scalar_mult(sc2, pts, r.qlen, nthreads=nt)
    finally:
        cv.free()


def test_user_curve_from_params(gpu_ctx):
    """ecamd_curve_from_params == built-in curve"""
    import libecc_amd
    c = CURVES["BRAINPOOLP256R1"]
    cv = libecc_amd.Curve(gpu_ctx, params=c)
    cv2 = gpu_ctx.curve("BRAINPOOLP256R1")
    try:
        sc = bytes(range(32)) * 4
        assert cv.scalar_mult(sc) == cv2.scalar_mult(sc)
        assert cv.scalar_mult(sc)[0][:64] == py_smul_bytes("BRAINPOOLP256R1", int.from_bytes(sc[:32], "big"))
    finally:
        cv.free()
        cv2.free()


# ---------------------------------------------------------------------------------------------
# protocol callers: ECDSA verify, ECC-CDH
# ---------------------------------------------------------------------------------------------
def make_sigs(curve, h, n, rng):
    """n valid ECDSA signatures made by the CPU oracle (random key, nonce, 24-byte message)"""
    o = Oracle(curve)
    q = CURVES[curve]["q"]

    def rs():
        return ((int.from_bytes(rand_bytes(rng, o.qlen + 8), "big") % (q - 1)) + 1).to_bytes(o.qlen, "big")

    privs = b"".join(rs() for _ in range(n))
    ks = b"".join(rs() for _ in range(n))
    msgs = [rand_bytes(rng, 24) for _ in range(n)]
    dg = b"".join(digest(h, m) for m in msgs)
    hl = len(dg) // n
    sigs, st = o.ecdsa_sign(privs, ks, dg, hl)
    assert set(st) == {0}
    pubs, st = o.scalar_mult(privs)
    assert set(st) == {0}
    return o, pubs, sigs, dg, hl, b"".join(msgs)


@pytest.mark.parametrize("curve,h", [("SECP256R1", "SHA256"), ("SECP256R1", "SHA512"), ("SECP384R1", "SHA384"),
                                     ("SECP521R1", "SHA512"), ("SECP192R1", "SHA256"), ("BRAINPOOLP256R1", "SHA256"),
                                     ("WEI25519", "SHA512"), ("SECP256K1", "SHA256")])
def test_ecdsa_verify_vs_oracle(gpu_ctx, curve, h):
    rng = np.random.default_rng(31)
    n = 40
    o, pubs, sigs, dg, hl, msgs = make_sigs(curve, h, n, rng)
    c = CURVES[curve]
    q, p, cl, ql = c["q"], c["p"], o.clen, o.qlen
    pubs, sigs, dg = bytearray(pubs), bytearray(sigs), bytearray(dg)
    # corrupt items 10.. in every way the reference distinguishes
    sigs[10 * 2 * ql + 3] ^= 0x10                                  # r bit
    sigs[11 * 2 * ql + ql + 5] ^= 0x01                             # s bit
    dg[12 * hl] ^= 0x80                                            # digest bit
    pubs[13 * 2 * cl + cl - 1] ^= 1                                # public key off curve
    sigs[14 * 2 * ql:14 * 2 * ql + ql] = bytes(ql)                 # r = 0
    sigs[15 * 2 * ql + ql:16 * 2 * ql] = bytes(ql)                 # s = 0
    sigs[16 * 2 * ql:16 * 2 * ql + ql] = q.to_bytes(ql, "big")     # r = q
    sigs[17 * 2 * ql + ql:18 * 2 * ql] = q.to_bytes(ql, "big")     # s = q
    pubs[18 * 2 * cl:18 * 2 * cl + cl] = p.to_bytes(cl, "big")     # x = p
    sigs[19 * 2 * ql:20 * 2 * ql] = sigs[20 * 2 * ql:21 * 2 * ql]  # someone else's signature
    s21 = int.from_bytes(sigs[21 * 2 * ql + ql:22 * 2 * ql], "big")
    sigs[21 * 2 * ql + ql:22 * 2 * ql] = (q - s21).to_bytes(ql, "big")   # (r, -s) is also valid
    pubs[22 * 2 * cl:23 * 2 * cl] = bytes(2 * cl)                  # (0, 0)
    pubs, sigs, dg = bytes(pubs), bytes(sigs), bytes(dg)
    cv = gpu_ctx.curve(curve)
    try:
        got = cv.ecdsa_verify(pubs, sigs, dg, hl)
        exp = o.ecdsa_verify(pubs, sigs, dg, hl)
        assert got == exp
        assert exp[:10] == bytes(10) and exp[21] == 0 and set(exp[10:20]) == {1} and exp[22] == 1
        if have_ref() and h in ("SHA256", "SHA384", "SHA512"):
            # the unmodified reference hashes the message itself; items with a corrupted digest differ by design
            ref = RefLib(curve).ecdsa_verify(h, pubs, sigs, msgs, 24)
            assert bytes(x for i, x in enumerate(ref) if i != 12) == bytes(x for i, x in enumerate(got) if i != 12)
    finally:
        cv.free()


@pytest.mark.parametrize("curve,h", [("SECP256R1", "SHA256"), ("SECP256K1", "SHA256"), ("BRAINPOOLP256R1", "SHA256"),
                                     ("SECP384R1", "SHA384"), ("SECP256R1", "SHA512")])
def test_ecdsa_crafted_signatures(gpu_ctx, curve, h):
    """Wycheproof-style family built by key recovery (tests/test_oracle.py::ecdsa_crafted_cases, pinned there against the
    unmodified reference): valid signatures whose R has x >= q (only "x mod q == r" accepts them -- on secp256r1 that is the
    projective comparison of k_p256_verify_loop), tiny / huge r and s, digests 0, 1, q, q - 1, all-ones, and their invalid
    twins; both through host buffers and tiled to a batch that takes the comb + interleaved-loop path"""
    from test_oracle import ecdsa_crafted_cases
    rng = np.random.default_rng(82)
    cv = gpu_ctx.curve(curve)
    try:
        for hash_name in (None, h):
            case = ecdsa_crafted_cases(curve, rng, hash_name)
            pubs, sigs, dgs, hl, exp = case[:5]
            assert exp.count(0) >= 12 and exp.count(1) >= 20
            assert cv.ecdsa_verify(pubs, sigs, dgs, hl) == exp
            reps = 6000 // len(exp) + 1
            assert cv.ecdsa_verify(pubs * reps, sigs * reps, dgs * reps, hl) == exp * reps
    finally:
        cv.free()


def test_ecdsa_verify_golden(gpu_ctx):
    """every ECDSA / RFC 6979 signature vector of the reference must verify on the GPU (public key
    derived on the GPU from the vector's private key), and must fail with one bit flipped"""
    kats = json.load(open(os.path.join(GOLDEN, "ecdsa_kats.json")))
    for curve in sorted({k["curve"] for k in kats}):
        ks = [k for k in kats if k["curve"] == curve]
        cv = gpu_ctx.curve(curve)
        try:
            for hname in sorted({k["hash"] for k in ks}):
                sel = [k for k in ks if k["hash"] == hname]
                privs = b"".join(bytes.fromhex(k["priv_key"]).rjust(cv.qlen, b"\0")[-cv.qlen:] for k in sel)
                pubs, st = cv.scalar_mult(privs)
                assert set(st) == {0}
                dg = b"".join(digest(hname, bytes.fromhex(k["msg"])) for k in sel)
                hl = len(dg) // len(sel)
                sigs = b"".join(bytes.fromhex(k["exp_sig"]) for k in sel)
                assert cv.ecdsa_verify(pubs, sigs, dg, hl) == bytes(len(sel)), (curve, hname)
                bad = bytearray(sigs)
                for i in range(len(sel)):
                    bad[i * 2 * cv.qlen + 7] ^= 4
                assert cv.ecdsa_verify(pubs, bytes(bad), dg, hl) == b"\1" * len(sel)
        finally:
            cv.free()


def test_ecdsa_sign_golden(gpu_ctx):
    """the reference's ECDSA (fixed k) and RFC 6979 vectors: signature bytes must be identical"""
    kats = json.load(open(os.path.join(GOLDEN, "ecdsa_kats.json")))
    for curve in sorted({k["curve"] for k in kats}):
        ks = [k for k in kats if k["curve"] == curve]
        cv = gpu_ctx.curve(curve)
        o = Oracle(curve)
        try:
            privs, nonces, dgs, exp = b"", b"", {}, b""
            for hname in sorted({k["hash"] for k in ks}):
                sel = [k for k in ks if k["hash"] == hname]
                privs = b"".join(bytes.fromhex(k["priv_key"]).rjust(cv.qlen, b"\0")[-cv.qlen:] for k in sel)
                nonces = b"".join((int(k["k"], 16) if k["k"] else rfc6979_nonce(curve, hname, bytes.fromhex(k["priv_key"]),
                                                                                   bytes.fromhex(k["msg"]))).to_bytes(cv.qlen, "big") for k in sel)
                dg = b"".join(digest(hname, bytes.fromhex(k["msg"])) for k in sel)
                hl = len(dg) // len(sel)
                sigs, st = cv.ecdsa_sign(privs, nonces, dg, hl)
                assert set(st) == {0}
                assert sigs == b"".join(bytes.fromhex(k["exp_sig"]) for k in sel), (curve, hname)
                # edge nonces: 0, q, q-1, 1
                q = CURVES[curve]["q"]
                en = b"".join(v.to_bytes(cv.qlen, "big") for v in (0, q, q - 1, 1))
                ep, ed = privs[:cv.qlen] * 4, dg[:hl] * 4
                assert cv.ecdsa_sign(ep, en, ed, hl) == o.ecdsa_sign(ep, en, ed, hl)
        finally:
            cv.free()


@pytest.mark.parametrize("curve", ["SECP192R1", "SECP224R1", "SECP256R1", "SECP384R1", "SECP521R1"])
def test_ecccdh_derive_golden_and_oracle(gpu_ctx, curve):
    kats = [k for k in json.load(open(os.path.join(GOLDEN, "ecccdh_kats.json"))) if k["curve"] == curve]
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        # the vectors of secp521r1 carry their private keys on 68 bytes (two leading zero bytes): nn_init_from_buf takes any
        # length, the batch entry point takes BYTECEIL(|q|) = 66 bytes
        ds = [bytes.fromhex(k["our_priv_key"]) for k in kats]
        assert all(len(x) >= cv.qlen and not any(x[:len(x) - cv.qlen]) for x in ds)
        d = b"".join(x[len(x) - cv.qlen:] for x in ds)
        peers = bytearray(b"".join(bytes.fromhex(k["peer_pub_key"]) for k in kats))
        sec, st = cv.ecccdh(d, bytes(peers))
        assert set(st) == {0}
        assert sec == b"".join(bytes.fromhex(k["exp_shared_secret"]) for k in kats)
        peers[5] ^= 1                                      # invalid peer key
        d2 = bytes(cv.qlen) + d[cv.qlen:]                  # d = 0 -> infinity -> error
        assert cv.ecccdh(d2, bytes(peers)) == o.ecccdh(d2, bytes(peers))
    finally:
        cv.free()


@pytest.mark.parametrize("kind,curve,ln", [("X25519", "WEI25519", 32), ("X448", "WEI448", 56)])
def test_xdh_vs_oracle_and_golden(gpu_ctx, kind, curve, ln):
    """X25519 / X448 batch: RFC 7748 vectors, then edge + random inputs against the oracle, including
    every rejection the reference makes (u >= p, twist, small order, zero)"""
    from test_oracle import KAT_XDH, xdh_edge_inputs
    rng = np.random.default_rng(16)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        ks = [k for k in KAT_XDH if k["kind"] == kind]
        k = b"".join(bytes.fromhex(x["our_priv_key"]) for x in ks)
        u = b"".join(bytes.fromhex(x["peer_pub_key"]) for x in ks)
        assert cv.xdh(k, u) == (b"".join(bytes.fromhex(x["exp_shared_secret"]) for x in ks), bytes(len(ks)))
        base = (9 if ln == 32 else 5).to_bytes(ln, "little")
        assert cv.xdh(k, base * len(ks))[0] == b"".join(bytes.fromhex(x["exp_our_pub_key"]) for x in ks)
        ek, eu = xdh_edge_inputs(ln, rng)
        exp = o.xdh(ek, eu)
        assert cv.xdh(ek, eu) == exp
        assert 0 in exp[1] and 1 in exp[1]
        # Diffie-Hellman property on a larger random batch: X(a, X(b, base)) == X(b, X(a, base))
        n = 2048
        a, b = rand_bytes(rng, ln * n), rand_bytes(rng, ln * n)
        pa, sa = cv.xdh(a, base * n)
        pb, sb = cv.xdh(b, base * n)
        assert set(sa) == {0} and set(sb) == {0}
        assert cv.xdh(a, pb) == cv.xdh(b, pa)
    finally:
        cv.free()


def test_ecccdh_cofactor_curve(gpu_ctx):
    """ECC-CDH on WEI25519 (h = 8): peer keys inside the subgroup, outside it (G + 2-torsion) and of small order"""
    curve = "WEI25519"
    rng = np.random.default_rng(17)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        c = CURVES[curve]
        p, n = c["p"], 32
        from oracles import py_add
        x2 = 486662 * pow(3, p - 2, p) % p
        T2 = x2.to_bytes(n, "big") + bytes(n)
        GT = py_add((c["gx"], c["gy"]), (x2, 0), c["a"], p)
        GT = GT[0].to_bytes(n, "big") + GT[1].to_bytes(n, "big")
        good, st = o.scalar_mult(rand_bytes(rng, 32 * 10))
        assert set(st) == {0}
        peers = good + T2 + GT + bytes(64) + good[:64]
        privs = rand_bytes(rng, 32 * 13) + bytes(32)
        exp = o.ecccdh(privs, peers)
        assert cv.ecccdh(privs, peers) == exp
        assert exp[1][:10] == bytes(10) and set(exp[1][10:]) == {1}
    finally:
        cv.free()


def test_ecdsa_verify_exceptional_pairs(gpu_ctx):
    """secp256r1 interleaved verification loop: public key G (or -G) with u1 == u2 makes every window
    add the same point twice (doubling / inverse cases of the incomplete addition); such items must be
    re-verified through the two-scalar-mult path and agree with the oracle"""
    curve = "SECP256R1"
    c = CURVES[curve]
    q, p = c["q"], c["p"]
    o = Oracle(curve)
    cv = gpu_ctx.curve(curve)
    try:
        G = c["gx"].to_bytes(32, "big") + c["gy"].to_bytes(32, "big")
        negG = c["gx"].to_bytes(32, "big") + (p - c["gy"]).to_bytes(32, "big")
        pubs, sigs, dgs = b"", b"", b""
        for k in (5, 0x1234567, q - 3, (q + 1) // 2, 2**200 + 17):
            kG, st = o.scalar_mult(k.to_bytes(32, "big"))
            r = int.from_bytes(kG[:32], "big") % q
            kinv = pow(k, q - 2, q)
            # key d = 1 (Q = G), digest e = r: s = k^-1 (e + r d) = 2 r / k, u1 = u2 = k / 2
            s = kinv * (2 * r) % q
            pubs += G
            sigs += r.to_bytes(32, "big") + s.to_bytes(32, "big")
            dgs += r.to_bytes(32, "big")
            # key d = q - 1 (Q = -G), digest e = r: s = k^-1 (r - r) = 0 is invalid, so use e = 3 r:
            # s = k^-1 (3 r - r) = 2 r / k, u1 = 3k/2, u2 = k/2 -> W' = [3k/2]G - [k/2]G = [k]G
            pubs += negG
            sigs += r.to_bytes(32, "big") + s.to_bytes(32, "big")
            dgs += (3 * r % q).to_bytes(32, "big")
            # and a rejected one: Q = -G with e = r gives u1 == u2, W' = infinity
            pubs += negG
            sigs += r.to_bytes(32, "big") + s.to_bytes(32, "big")
            dgs += r.to_bytes(32, "big")
        exp = o.ecdsa_verify(pubs, sigs, dgs, 32)
        got = cv.ecdsa_verify(pubs, sigs, dgs, 32)
        assert got == exp
        assert exp == bytes([0, 0, 1] * 5)
    finally:
        cv.free()


def test_eddsa25519_verify_vs_oracle_and_golden(gpu_ctx):
    """Ed25519 batch verification: the reference's RFC 8032 Ed25519ctx / Ed25519ph vectors, then valid,
    torsion-shifted (accepted only by libecc's cofactored equation) and every class of rejected input
    against the oracle; finally a large tiled batch with corruptions at known positions"""
    from test_oracle import KAT_EDDSA, ed25519_cases, eddsa_kat_inputs
    rng = np.random.default_rng(34)
    cv = gpu_ctx.curve("WEI25519")
    o = Oracle("WEI25519")
    try:
        pubs, sigs, hram = eddsa_kat_inputs()
        assert cv.eddsa_verify(pubs, sigs, hram) == bytes(len(KAT_EDDSA))
        bad = bytearray(sigs)
        bad[64 + 3] ^= 1
        assert cv.eddsa_verify(pubs, bytes(bad), hram) == bytes([0, 1, 0, 0, 0])
        pubs, sigs, msgs, hram = ed25519_cases(rng, nvalid=40)
        exp = o.eddsa_verify(pubs, sigs, hram)
        assert 0 in exp and 1 in exp and exp[:52] == bytes(52)
        assert cv.eddsa_verify(pubs, sigs, hram) == exp
        assert cv.eddsa_verify(b"", b"", b"") == b""
        # 2^15 items: the cases tiled, with extra corruption of the hash at every 97th item
        n0 = len(exp)
        reps = (1 << 15) // n0 + 1
        P, S, H = pubs * reps, sigs * reps, bytearray(hram * reps)
        want = bytearray(exp * reps)
        for i in range(0, n0 * reps, 97):
            H[64 * i + 9] ^= 0x20
            want[i] = 1
        assert cv.eddsa_verify(P, S, bytes(H)) == bytes(want)
    finally:
        cv.free()


def test_eddsa_verify_rejects_other_curves(gpu_ctx):
    cv = gpu_ctx.curve("SECP256R1")
    try:
        with pytest.raises(Exception):
            cv.eddsa_verify(bytes(32), bytes(64), bytes(64))
    finally:
        cv.free()


def test_device_pointer_entry_points(gpu_ctx):
    """the *_dev forms (buffers already in HBM, caller's stream) give the same bytes as the host-pointer
    forms: scalar mult, ECDSA verify (interleaved secp256r1 loop incl. exceptional items, and the
    two-scalar-mult path of another curve), Ed25519 verify, X25519"""
    import torch
    from test_oracle import ed25519_cases
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(35)
    stream = torch.cuda.Stream(device=dev)

    def t(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    def empty(n):
        return torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)

    for curve, h in (("SECP256R1", "SHA256"), ("BRAINPOOLP256R1", "SHA256")):
        n = 96
        o, pubs, sigs, dg, hl, _ = make_sigs(curve, h, n, rng)
        sigs = bytearray(sigs)
        sigs[5 * 2 * o.qlen + 2] ^= 4
        sigs = bytes(sigs)
        if curve == "SECP256R1":
            # items that leave the interleaved loop through the exceptional-pair path (Q = G, u1 == u2)
            c = CURVES[curve]
            q = c["q"]
            G = c["gx"].to_bytes(32, "big") + c["gy"].to_bytes(32, "big")
            for k in (7, q - 5):
                kG, _ = o.scalar_mult(k.to_bytes(32, "big"))
                r = int.from_bytes(kG[:32], "big") % q
                s = pow(k, q - 2, q) * (2 * r) % q
                pubs += G
                sigs += r.to_bytes(32, "big") + s.to_bytes(32, "big")
                dg += r.to_bytes(32, "big")
            n += 2
        cv = gpu_ctx.curve(curve)
        try:
            exp = cv.ecdsa_verify(pubs, sigs, dg, hl)
            assert exp == o.ecdsa_verify(pubs, sigs, dg, hl) and exp[5] == 1 and exp[0] == 0
            dp, ds, dd, dr = t(pubs), t(sigs), t(dg), empty(n)
            torch.cuda.synchronize()
            cv.ecdsa_verify_dev(n, dp.data_ptr(), ds.data_ptr(), dd.data_ptr(), hl, dr.data_ptr(), stream.cuda_stream)
            stream.synchronize()   # every *_dev form only enqueues
            assert bytes(dr.cpu().numpy()) == exp
            # scalar multiplication on the caller's stream
            sc = rand_bytes(rng, n * o.qlen)
            e_out, e_st = cv.scalar_mult(sc, pubs[:n * 2 * o.clen])
            dsc, dout, dst = t(sc), empty(n * 2 * o.clen), empty(n)
            torch.cuda.synchronize()
            cv.scalar_mult_dev(n, dsc.data_ptr(), o.qlen, dp.data_ptr(), dout.data_ptr(), dst.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            assert (bytes(dout.cpu().numpy()), bytes(dst.cpu().numpy())) == (e_out, e_st)
        finally:
            cv.free()
    cv = gpu_ctx.curve("WEI25519")
    try:
        pubs, sigs, msgs, hram = ed25519_cases(rng, nvalid=20)
        n = len(pubs) // 32
        exp = cv.eddsa_verify(pubs, sigs, hram)
        assert 0 in exp and 1 in exp
        dp, ds, dh, dr = t(pubs), t(sigs), t(hram), empty(n)
        torch.cuda.synchronize()
        cv.eddsa_verify_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), dr.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        assert bytes(dr.cpu().numpy()) == exp
        n = 200
        k, base = rand_bytes(rng, 32 * n), (9).to_bytes(32, "little") * n
        pub, st = cv.xdh(k, base)
        k2 = rand_bytes(rng, 32 * n)
        exp = cv.xdh(k2, pub)
        dk, du, dout, dst = t(k2), t(pub), empty(32 * n), empty(n)
        torch.cuda.synchronize()
        cv.xdh_dev(n, dk.data_ptr(), du.data_ptr(), dout.data_ptr(), dst.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        assert (bytes(dout.cpu().numpy()), bytes(dst.cpu().numpy())) == exp
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP521R1", "WEI25519", "BRAINPOOLP256R1"])
def test_projective_wire_format(gpu_ctx, curve):
    """projective X || Y || Z in / out (prj_pt_import_from_buf, prj_pt_export_to_buf) against the oracle:
    scaled representatives, infinity, (0:0:0), off-curve and out-of-range triples; format mixes"""
    from test_oracle import prj_cases
    rng = np.random.default_rng(52)
    pts, scal, ql = prj_cases(curve, rng, nrand=80)
    o = Oracle(curve)
    cl = o.clen
    n = len(pts) // (3 * cl)
    cv = gpu_ctx.curve(curve)
    try:
        exp = o.prj(pts, scal, ql)
        got = cv.scalar_mult_fmt(scal, pts, 1, 1, ql)
        assert got == exp
        # projective in, affine out: the same points without the trailing 0..01
        ga, sa = cv.scalar_mult_fmt(scal, pts, 1, 0, ql)
        assert sa == exp[1]
        assert ga == b"".join(exp[0][3 * cl * i:3 * cl * i + 2 * cl] for i in range(n))
        # affine in, projective out == the classic entry point plus Z = 1
        aff_in = b"".join(ga[2 * cl * i:2 * cl * (i + 1)] for i in range(n))
        ca, cs = cv.scalar_mult(scal, aff_in, ql)
        pa, ps = cv.scalar_mult_fmt(scal, aff_in, 0, 1, ql)
        assert ps == cs
        one = (1).to_bytes(cl, "big")
        assert pa == b"".join((ca[2 * cl * i:2 * cl * (i + 1)] + one) if cs[i] == 0 else bytes(3 * cl) for i in range(n))
        # generator with projective output
        g3, gs = cv.scalar_mult_fmt(scal, None, 0, 1, ql)
        g2, gs2 = cv.scalar_mult(scal, None, ql)
        assert gs == gs2 and g3[:3 * cl] == g2[:2 * cl] + one
        # normalisation only
        assert cv.unique(pts, 1, 1) == o.prj(pts)
        ua, us = cv.unique(pts, 1, 0)
        assert us == o.prj(pts)[1]
        va, vs = cv.unique(aff_in, 0, 1)
        assert vs == bytes(1 if (sa[i] != 0) else 0 for i in range(n))   # zeros stand for failed items: (0, 0) is off the curve
    finally:
        cv.free()


def test_every_builtin_curve(gpu_ctx):
    """all 44 curves libecc ships (tests/golden/curves.json): fixed- and variable-base scalar mult with
    random and edge scalars, one ECDSA verification and one ECC-CDH each, against the oracle"""
    rng = np.random.default_rng(53)
    assert len(CURVES) == 44
    for curve in sorted(CURVES):
        o = Oracle(curve)
        cv = gpu_ctx.curve(curve)
        try:
            ql, cl = o.qlen, o.clen
            n = 12
            sc = rand_bytes(rng, ql * (n - 4)) + b"".join(v.to_bytes(ql, "big") for v in
                                                         (0, 1, CURVES[curve]["q"] % (1 << (8 * ql)), (1 << (8 * ql)) - 1))
            pub = o.scalar_mult(sc)
            assert cv.scalar_mult(sc) == pub, curve
            base = pub[0][:2 * cl] * n
            sc2 = rand_bytes(rng, ql * n)
            assert cv.scalar_mult(sc2, base) == o.scalar_mult(sc2, base), curve
            if CURVES[curve]["q"].bit_length() >= 160:
                _, pubs, sigs, dg, hl, _ = make_sigs(curve, "SHA256", 4, rng)
                bad = bytearray(sigs)
                bad[2 * ql + 1] ^= 2
                assert cv.ecdsa_verify(pubs, bytes(bad), dg, hl) == o.ecdsa_verify(pubs, bytes(bad), dg, hl) == bytes([0, 1, 0, 0]), curve
                d = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (CURVES[curve]["q"] - 1)) + 1).to_bytes(ql, "big")
                             for _ in range(4))
                assert cv.ecccdh(d, pubs) == o.ecccdh(d, pubs), curve
                ks = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (CURVES[curve]["q"] - 1)) + 1).to_bytes(ql, "big")
                              for _ in range(4))
                assert cv.ecdsa_sign(d, ks, dg, hl) == o.ecdsa_sign(d, ks, dg, hl), curve
        finally:
            cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "WEI25519", "SECP224R1", "BRAINPOOLP256R1", "SECP384R1", "SECP521R1", "WEI448"])
def test_fixed_base_comb_edges(gpu_ctx, curve):
    """fixed base runs on the 16-bit comb table of the generator once a batch is large enough: scalars
    that stress the signed recoding (digits 0x0000 / 0x7fff / 0x8000 / 0xffff in every window, powers of
    two around the window boundaries, q-1, q, q+1, all ones, zero) and every scalar length, against the
    oracle; the same through ECDSA signing and verification"""
    rng = np.random.default_rng(54)
    q = CURVES[curve]["q"]
    o = Oracle(curve)
    cv = gpu_ctx.curve(curve)
    try:
        ql = o.qlen
        top = 1 << (8 * ql)
        nwin = (8 * ql + 15) // 16
        vals = [0, 1, 2, q - 1, q, q + 1, top - 1, top >> 1, (top >> 1) - 1, top - (top >> 16), (top >> 16) - 1, 2 * q, 3 * q + 5]
        for pat in (0x0000, 0x0001, 0x7fff, 0x8000, 0x8001, 0xffff):
            vals.append(sum(pat << (16 * j) for j in range(nwin)))
            vals.append(sum((pat if j % 2 else 0x8000) << (16 * j) for j in range(nwin)))
        for j in range(nwin):
            vals += [1 << (16 * j), (1 << (16 * j)) - 1, 0x8000 << (16 * j), (0x8000 << (16 * j)) - 1, 0x7fff << (16 * j)]
        vals = [v % top for v in vals]
        sc = b"".join(v.to_bytes(ql, "big") for v in vals) + rand_bytes(rng, ql * 64)
        exp = o.scalar_mult(sc)
        assert cv.scalar_mult(sc) == exp
        assert 2 in exp[1] and 0 in exp[1]
        for slen in range(1, 4 * ((CURVES[curve]["p"].bit_length() + 31) // 32) + 1, 1 if ql <= 32 else 5):
            s2 = rand_bytes(rng, slen * 45) + b"\xff" * slen + b"\x00" * slen + b"\x80" * slen
            assert cv.scalar_mult(s2, None, slen) == o.scalar_mult(s2, None, slen), slen
        # sign with nonces of those shapes, verify what was signed
        ks = b"".join(v.to_bytes(ql, "big") for v in vals if 0 < v < q)
        n = len(ks) // ql
        d = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
        dg = rand_bytes(rng, 32 * n)
        sigs = cv.ecdsa_sign(d, ks, dg, 32)
        assert sigs == o.ecdsa_sign(d, ks, dg, 32)
        pubs, st = cv.scalar_mult(d)
        assert set(st) == {0}
        assert cv.ecdsa_verify(pubs, sigs[0], dg, 32) == o.ecdsa_verify(pubs, sigs[0], dg, 32)
    finally:
        cv.free()


def test_first_large_dev_call_builds_comb_safely():
    """a fresh context whose very first call is a large device-pointer Ed25519 verification on the caller's
    stream: the comb table of the generator is built in the middle of that call (after the [h]A
    multiplication was enqueued) and must not disturb it"""
    import torch
    import libecc_amd
    from test_oracle import ed25519_cases
    rng = np.random.default_rng(36)
    dev = torch.device("cuda:0")
    pubs, sigs, msgs, hram = ed25519_cases(rng, nvalid=30)
    exp = Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
    n0 = len(exp)
    reps = (1 << 16) // n0 + 1
    n = n0 * reps
    ctx = libecc_amd.Context(0)
    cv = ctx.curve("WEI25519")
    try:
        stream = torch.cuda.Stream(device=dev)
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        dp, ds, dh = t(pubs * reps), t(sigs * reps), t(hram * reps)
        dr = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        cv.eddsa_verify_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), dr.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        assert bytes(dr.cpu().numpy()) == exp * reps
    finally:
        cv.free()
        ctx.close()


@pytest.mark.parametrize("curve", ["WEI25519", "WEI448"])
def test_eddsa_verify_all_batch(gpu_ctx, curve):
    """ec_eddsa_verify_all_batch = the whole-batch predicate of the reference's ec_verify_batch (pinned in
    tests/test_oracle.py::test_eddsa_batch_predicate_vs_reference): accept iff the oracle accepts every item; the
    first rejected index is the oracle's; n = 0 is an error as in the reference"""
    import libecc_amd
    from test_oracle import ed25519_cases, ed448_cases, eddsa_subset, ED_MSG_LEN, ED448_MSG_LEN
    e448 = curve == "WEI448"
    rng = np.random.default_rng(72)
    kl, sl, ml, hl = (57, 114, ED448_MSG_LEN, 114) if e448 else (32, 64, ED_MSG_LEN, 64)
    pubs, sigs, msgs, hram = (ed448_cases if e448 else ed25519_cases)(rng, 10)
    n = len(pubs) // kl
    one = Oracle(curve).eddsa_verify(pubs, sigs, hram)
    good = [i for i in range(n) if one[i] == 0]
    bad = [i for i in range(n) if one[i]]
    cv = gpu_ctx.curve(curve)
    try:
        def run(idx):
            P, S, _, H = eddsa_subset(idx, pubs, sigs, msgs, hram, kl, sl, ml, hl)
            return cv.eddsa_verify_all(P, S, H)
        assert run(good) == (True, len(good))
        assert run(good[:1]) == (True, 1)
        big = good * 40                                   # several hundred items, all valid
        assert run(big) == (True, len(big))
        for b in bad:
            assert run([b]) == (False, 0)
            assert run(good[:5] + [b] + good[5:]) == (False, 5)
        assert run(big + [bad[0]] + good + [bad[1]]) == (False, len(big))
        assert run(list(range(n))) == (False, bad[0])
        with pytest.raises(libecc_amd.EcamdError):
            cv.eddsa_verify_all(b"", b"", b"")
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "BRAINPOOLP384R1", "WEI25519"])
def test_sign_and_ecccdh_device_pointer_forms(gpu_ctx, curve):
    """ec_ecdsa_sign_batch_dev / ec_ecccdh_derive_batch_dev: device pointers on a caller's stream, scratch bounded by a
    small max_chunk, against the oracle (edge nonces, bad and out-of-subgroup peer keys, d = 0 included)"""
    import torch
    import libecc_amd
    rng = np.random.default_rng(73)
    dev = torch.device("cuda:0")
    c = CURVES[curve]
    o = Oracle(curve)
    ctx2 = libecc_amd.Context(0)
    ctx2.set_max_chunk(192)
    cv = ctx2.curve(curve)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    try:
        n, ql, cl, q = 500, cv.qlen, cv.clen, c["q"]
        stream = torch.cuda.Stream(device=dev)
        # signing
        hl = 32
        privs = b"".join(int(rng.integers(1, 1 << 62)).to_bytes(ql, "big") for _ in range(n))
        raw = rand_bytes(rng, (ql + 8) * n)
        nonces = bytearray(b"".join((int.from_bytes(raw[(ql + 8) * i:(ql + 8) * (i + 1)], "big") % (q - 1) + 1).to_bytes(ql, "big")
                                    for i in range(n)))
        for i, v in enumerate((0, q, q - 1, 1, q + 1)):
            nonces[ql * i:ql * i + ql] = (v % (1 << (8 * ql))).to_bytes(ql, "big")
        nonces, dg = bytes(nonces), rand_bytes(rng, hl * n)
        exp = o.ecdsa_sign(privs, nonces, dg, hl)
        dp, dn, dd = t(privs), t(nonces), t(dg)
        dsig = torch.full((2 * ql * n,), 0xAA, dtype=torch.uint8, device=dev)
        dst = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        cv.ecdsa_sign_dev(n, dp.data_ptr(), dn.data_ptr(), dd.data_ptr(), hl, dsig.data_ptr(), dst.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        got = (bytes(dsig.cpu().numpy()), bytes(dst.cpu().numpy()))
        ok = [i for i in range(n) if exp[1][i] == 0]
        assert got[1] == exp[1] and 1 in exp[1] and len(ok) > n - 10
        assert all(got[0][2 * ql * i:2 * ql * (i + 1)] == exp[0][2 * ql * i:2 * ql * (i + 1)] for i in ok)
        assert cv.ecdsa_sign(privs, nonces, dg, hl)[1] == exp[1]
        # key agreement
        peers, st = o.scalar_mult(rand_bytes(rng, ql * n))
        peers = bytearray(peers)
        peers[5] ^= 1                                            # off the curve
        peers[2 * cl * 3:2 * cl * 4] = b"\xff" * (2 * cl)       # coordinates >= p
        if curve == "WEI25519":
            from oracles import py_add
            x2 = 486662 * pow(3, c["p"] - 2, c["p"]) % c["p"]
            GT = py_add((c["gx"], c["gy"]), (x2, 0), c["a"], c["p"])
            peers[2 * cl * 7:2 * cl * 8] = GT[0].to_bytes(cl, "big") + GT[1].to_bytes(cl, "big")   # outside the subgroup
            peers[2 * cl * 8:2 * cl * 9] = x2.to_bytes(cl, "big") + bytes(cl)                        # order 2
        peers = bytes(peers)
        d = bytearray(rand_bytes(rng, ql * n))
        d[ql * 9:ql * 10] = bytes(ql)                            # d = 0: infinity
        d[ql * 10:ql * 11] = q.to_bytes(ql, "big")
        d = bytes(d)
        exp = o.ecccdh(d, peers)
        dk, dq = t(d), t(peers)
        dsec = torch.full((cl * n,), 0xAA, dtype=torch.uint8, device=dev)
        dst = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        cv.ecccdh_dev(n, dk.data_ptr(), dq.data_ptr(), dsec.data_ptr(), dst.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        assert (bytes(dsec.cpu().numpy()), bytes(dst.cpu().numpy())) == exp
        assert cv.ecccdh(d, peers) == exp
        assert exp[1].count(1) >= (6 if curve == "WEI25519" else 4)
    finally:
        cv.free()
        ctx2.close()


def test_secp256k1_field_edges(gpu_ctx):
    """secp256k1 runs on the plain-residue field with folds by 2^256 = 2^32 + 977: coordinates whose limbs sit at the
    edges of that reduction (x or y close to 0, to p, to 2^32 + 977 multiples) and scalars at the edges of the group
    order, against the oracle; plus ECDSA verification (mod-q algebra next to the new mod-p field)"""
    curve = "SECP256K1"
    c = CURVES[curve]
    p, q = c["p"], c["q"]
    rng = np.random.default_rng(74)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        def lift(x):
            """smallest x' >= x with a point (x', y) on y^2 = x^3 + 7"""
            while True:
                t = (pow(x, 3, p) + 7) % p
                y = pow(t, (p + 1) // 4, p)
                if y * y % p == t:
                    return x, y
                x += 1
        pts = []
        for x0 in (1, 2**32 + 977, 2**29, 2**232, 2**255, p - 2**33, p - 1000, (p - 1) // 2, 2**256 - 2**33 - 5000, 3 * 2**224):
            x, y = lift(x0 % p)
            pts.append(x.to_bytes(32, "big") + y.to_bytes(32, "big"))
            pts.append(x.to_bytes(32, "big") + (p - y).to_bytes(32, "big"))
        pts.append((p).to_bytes(32, "big") + bytes(32))                 # x = p: rejected
        pts.append(bytes(31) + b"\x01" + bytes(32))                    # off the curve
        n = len(pts)
        scal = [1, 2, q - 1, q, q + 1, 2**255, 2**256 - 1, 0, (q - 1) // 2, 3]
        S = b"".join((scal[i % len(scal)]).to_bytes(32, "big") for i in range(n))
        P = b"".join(pts)
        assert cv.scalar_mult(S, P) == o.scalar_mult(S, P)
        R = rand_bytes(rng, 32 * n)
        assert cv.scalar_mult(R, P) == o.scalar_mult(R, P)
        G3 = cv.scalar_mult(S)
        assert G3 == o.scalar_mult(S)
        o2, pubs, sigs, dg, hl, _ = make_sigs(curve, "SHA256", 64, rng)
        bad = bytearray(sigs)
        bad[64 * 5 + 2] ^= 1
        assert cv.ecdsa_verify(pubs, bytes(bad), dg, hl) == o.ecdsa_verify(pubs, bytes(bad), dg, hl)
    finally:
        cv.free()


@pytest.mark.parametrize("curve,log2n", [("SECP256R1", 17), ("SECP256K1", 17), ("SECP384R1", 15), ("BRAINPOOLP256R1", 15)])
def test_protocol_round_trips_large(gpu_ctx, curve, log2n):
    """size-independent properties of the protocol entry points on batches far beyond what the oracle covers:
    sign -> verify accepts every signature and rejects exactly the corrupted ones; ECC-CDH is symmetric
    (x([a][b]G) both ways); 64 random items of each result are compared with the oracle"""
    rng = np.random.default_rng(75)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        n, ql, cl, q = 1 << log2n, cv.qlen, cv.clen, CURVES[curve]["q"]
        raw = rng.integers(0, 256, size=(3, n, ql + 8), dtype=np.uint8)
        scal = lambda r: b"".join(((int.from_bytes(r[i].tobytes(), "big") % (q - 1)) + 1).to_bytes(ql, "big") for i in range(n))
        da, db, ks = scal(raw[0]), scal(raw[1]), scal(raw[2])
        hl = 32
        dg = rand_bytes(rng, hl * n)
        pa, st = cv.scalar_mult(da)
        assert set(st) == {0}
        pb, st = cv.scalar_mult(db)
        assert set(st) == {0}
        sigs, st = cv.ecdsa_sign(da, ks, dg, hl)
        assert set(st) == {0}
        assert cv.ecdsa_verify(pa, sigs, dg, hl) == bytes(n)
        bad = np.frombuffer(sigs, dtype=np.uint8).copy().reshape(n, 2 * ql)
        which = np.arange(n) % 7 == 3
        bad[which, (np.arange(n) % (2 * ql))[which]] ^= 0x20
        res = cv.ecdsa_verify(pa, bad.tobytes(), dg, hl)
        assert res == bytes(which.astype(np.uint8))
        assert cv.ecdsa_verify(pb, sigs, dg, hl) == bytes([1]) * n        # someone else's key
        s1, st1 = cv.ecccdh(da, pb)
        s2, st2 = cv.ecccdh(db, pa)
        assert set(st1) == {0} and (s1, st1) == (s2, st2)
        idx = [int(i) for i in rng.choice(n, size=64, replace=False)]
        cut = lambda b, w: b"".join(b[w * i:w * i + w] for i in idx)
        assert o.ecdsa_sign(cut(da, ql), cut(ks, ql), cut(dg, hl), hl)[0] == cut(sigs, 2 * ql)
        assert o.ecdsa_verify(cut(pa, 2 * cl), cut(bad.tobytes(), 2 * ql), cut(dg, hl), hl) == bytes(res[i] for i in idx)
        assert o.ecccdh(cut(da, ql), cut(pb, 2 * cl))[0] == cut(s1, cl)
    finally:
        cv.free()


def test_eddsa25519_sign_steps(gpu_ctx):
    """ec_eddsa_sign_R_batch / ec_eddsa_sign_S_batch against the oracle (orc_eddsa25519_sign_R/S_batch, pinned against the
    reference's ec_sign in tests/test_oracle.py): with the two hashes done here, R || S must also be the signature bytes of
    the RFC 8032 signer of tests/oracles.py and, when oracle/_ref is there, of the unmodified reference; Ed25519ctx-style dom2 prefixes only change the hashes; edge values of
    the 64-byte hash (0, q, all ones) and of the secret scalar"""
    import hashlib
    import libecc_amd
    import oracles as O
    from test_oracle import ED_MSG_LEN
    rng = np.random.default_rng(76)
    q = O.ED_Q
    cv = gpu_ctx.curve("WEI25519")
    try:
        n = 300
        seeds, msgs = rand_bytes(rng, 32 * n), rand_bytes(rng, ED_MSG_LEN * n)
        doms = [b"" if i % 3 else b"SigEd25519 no Ed25519 collisions" + bytes([0, 3]) + b"ctx" for i in range(n)]
        A, sig, a_sc, r_hash = [], [], [], []
        for i in range(n):
            seed, m = seeds[32 * i:32 * i + 32], msgs[ED_MSG_LEN * i:ED_MSG_LEN * (i + 1)]
            pub, sg, _ = O.ed25519_sign(seed, m, dom=doms[i])
            hk = hashlib.sha512(seed).digest()
            a = (int.from_bytes(hk[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
            A.append(pub)
            sig.append(sg)
            a_sc.append(a.to_bytes(32, "little"))
            r_hash.append(hashlib.sha512(doms[i] + hk[32:] + m).digest())
        R, st = cv.eddsa_sign_R(b"".join(r_hash))
        assert st == bytes(n)
        assert R == b"".join(sg[:32] for sg in sig)
        o = Oracle("WEI25519")
        assert (R, st) == o.eddsa_sign_R(b"".join(r_hash))
        hram = [hashlib.sha512(doms[i] + R[32 * i:32 * i + 32] + A[i] + msgs[ED_MSG_LEN * i:ED_MSG_LEN * (i + 1)]).digest()
                for i in range(n)]
        S = cv.eddsa_sign_S(b"".join(r_hash), b"".join(hram), b"".join(a_sc))
        assert S == b"".join(sg[32:] for sg in sig)
        assert S == o.eddsa_sign_S(b"".join(r_hash), b"".join(hram), b"".join(a_sc))
        if have_ref():
            idx = [i for i in range(n) if doms[i] == b""][:24]
            rp, rs, rst = O.ref_ed25519_sign(b"".join(seeds[32 * i:32 * i + 32] for i in idx),
                                             b"".join(msgs[ED_MSG_LEN * i:ED_MSG_LEN * (i + 1)] for i in idx), ED_MSG_LEN)
            assert rst == bytes(len(idx))
            assert rs == b"".join(R[32 * i:32 * i + 32] + S[32 * i:32 * i + 32] for i in idx)
        # the signatures verify on the GPU as well
        plain = [i for i in range(n) if doms[i] == b""]
        cut = lambda b, w: b"".join(b[w * i:w * i + w] for i in plain)
        sigs = b"".join(R[32 * i:32 * i + 32] + S[32 * i:32 * i + 32] for i in plain)
        assert cv.eddsa_verify(b"".join(A[i] for i in plain), sigs, b"".join(hram[i] for i in plain)) == bytes(len(plain))
        # edge values
        eh = [bytes(64), q.to_bytes(64, "little"), (q - 1).to_bytes(64, "little"), (q + 1).to_bytes(64, "little"), b"\xff" * 64,
              (1).to_bytes(64, "little"), (1 << 511).to_bytes(64, "little")]
        ea = [bytes(32), b"\xff" * 32, (1 << 254).to_bytes(32, "little"), q.to_bytes(32, "little"), (q - 1).to_bytes(32, "little"),
              a_sc[0], a_sc[1]]
        R2, st2 = cv.eddsa_sign_R(b"".join(eh))
        assert st2 == bytes(len(eh)) and (R2, st2) == o.eddsa_sign_R(b"".join(eh))
        for i, h in enumerate(eh):
            r = int.from_bytes(h, "little") % q
            exp = (1).to_bytes(32, "little") if r == 0 else O.ed_encode(O.ed_mul(r, O.ED_B))
            assert R2[32 * i:32 * i + 32] == exp, i
        S2 = cv.eddsa_sign_S(b"".join(eh), b"".join(reversed(eh)), b"".join(ea))
        assert S2 == o.eddsa_sign_S(b"".join(eh), b"".join(reversed(eh)), b"".join(ea))
        for i in range(len(eh)):
            r, h, a = (int.from_bytes(x, "little") for x in (eh[i], eh[len(eh) - 1 - i], ea[i]))
            assert S2[32 * i:32 * i + 32] == ((r + h * a) % q).to_bytes(32, "little"), i
        other = gpu_ctx.curve("SECP256R1")
        try:
            with pytest.raises(libecc_amd.EcamdError):
                other.eddsa_sign_R(bytes(64))
        finally:
            other.free()
    finally:
        cv.free()


LIBDIR = os.path.join(os.path.dirname(GOLDEN), "..", "libecc_amd", "lib")


COMPAT_RUNS = {
    # every row, 640 items per case with the edge families
    "full_640": (["640"], {}),
    # the Ed25519 multi-scalar multiplication forced for every batch size: ec_verify_batch's EdDSA branch then decides valid batches by
    # the combination and falls back to the item-by-item pass for the others
    "msm_forced": (["200"], {"ECAMD_MSM_MIN": "1"}),
    # three ranks on device 0 with tiny chunks: the verification calls are streamed (the pool packs while the C ABI asks for each
    # range through the producer hook, which the multi-device layer offsets per shard), several chunks per shard, short first chunk
    "three_ranks_streamed": (["quick", "600"], {"ECAMD_DEVICES": "0,0,0", "ECAMD_COMPAT_READY_ITEMS": "512", "ECAMD_HOST_CHUNK": "96",
                                                "ECAMD_HOST_RAMP_MIN": "16"}),
    # eight ranks on device 0, uneven shards (the driver's 8-GPU shape without the hardware)
    "eight_ranks": (["quick", "203"], {"ECAMD_DEVICES": "0,0,0,0,0,0,0,0"}),
    # BIP0340 / ECFSDSA batches through the multi-scalar multiplication whatever their size (valid batches are decided by it, spoiled
    # ones fall through to the item-by-item pass); three ranks
    "schnorr_msm_forced": (["quick", "300"], {"ECAMD_COMPAT_SCHNORR_MSM_MIN": "1", "ECAMD_COMPAT_ED_MSM_MIN": "1", "ECAMD_DEVICES": "0,0,0"}),
    # the paths round 4 left as fall-backs: host hashing, chunked calls, two-pass EdDSA, nn_get_random_mod on the host, the scanned
    # window loop for secret fixed-base multiplications, the saturated-word projective import
    "fallback_paths": (["quick", "150"], {"ECAMD_COMPAT_HOST_HASH": "1", "ECAMD_COMPAT_NO_STREAM": "1", "ECAMD_COMPAT_ED_TWO_PASS": "1",
                                          "ECAMD_COMPAT_HOST_RANDMOD": "1", "ECAMD_NO_SECRET_COMB": "1", "ECAMD_NO_PRJ_IMPORT_G29": "1",
                                          "ECAMD_COMPAT_PRJ_KEYS": "1"}),
}


@pytest.mark.parametrize("run", sorted(COMPAT_RUNS))
def test_libecc_typed_boundary_vs_scalar_api(run):
    """include/libecc_amd_compat.h through libsign_amd.so, driven by a libecc application (libecc_amd/compat/compat_check.c):
    prj_pt_mul_batch(prj_pt[], nn[], prj_pt[]), ecccdh_derive_secret_batch and ec_verify_batch -- called with const u8 **,
    const ec_pub_key ** arrays exactly as tests/ec_self_tests_core.c:373-383, 556-616 call it -- for ECDSA, DECDSA, the
    five EdDSA variants, BIP0340 and ECFSDSA; ec_sign_batch (same nonce hook as _ec_sign, RFC 6979, EdDSA: signature bytes equal),
    ec_key_pair_gen_batch / import / init_pubkey_from_privkey_batch for seven algorithms' key rules, x25519_batch / x448_batch --
    every result compared with libecc's own scalar function (the CPU code of the libecc the library was linked from) on
    the same structures.  One pytest case per compat_check run (COMPAT_RUNS), so that one bad row hides nothing else."""
    import subprocess
    exe = os.path.join(LIBDIR, "compat_check")
    if not os.path.exists(exe):
        pytest.skip("libecc_amd/lib/compat_check not built (needs the libecc sources at build time)")
    args, extra = COMPAT_RUNS[run]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=1500, env=dict(os.environ, **extra))
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "compat_check: all ok" in r.stdout and "FAILED" not in r.stdout and "MISMATCH" not in r.stdout
    if run == "schnorr_msm_forced":
        assert int(r.stdout.split("schnorr multi-scalar calls:")[1].split()[0]) >= 6, r.stdout[-600:]
        assert int(r.stdout.split("ed25519 whole-batch calls:")[1].split()[0]) >= 4, r.stdout[-600:]
    if run == "full_640":
        assert r.stdout.count(": ok") >= 48
        for row in ("ec_sign_batch ECDSA", "ec_sign_batch DECDSA", "ec_sign_batch EDDSA25519", "ec_sign_batch EDDSA448", "ec_key_pair_{gen,import}_batch",
                    "x25519_batch", "x448_batch", "ec_verify_batch BIP0340", "ec_verify_batch ECFSDSA", "foreign generator"):
            assert row in r.stdout, row
        sent = int(r.stdout.split("items sent to the GPU:")[1].split()[0])
        assert sent >= 640 * 20, sent          # the batch forms did run on the GPU (there is no CPU fallback to hide behind)


def test_libecc_self_tests_against_libsign_amd():
    """The drop-in check SURVEY.md section 8b names: libecc's OWN self-test program (tests/ec_self_tests.c, unmodified),
    linked against libsign_amd.so instead of libsign.so.  `vectors` runs every known-answer test of the snapshot; each
    signature case also calls ec_verify_batch (batch of 1) when is_verify_batch_mode_supported says so -- EdDSA, and with
    this library ECDSA / DECDSA -- and those calls run on the GPU."""
    import subprocess
    exe = os.path.join(LIBDIR, "ec_self_tests_amd")
    if not os.path.exists(exe):
        pytest.skip("libecc_amd/lib/ec_self_tests_amd not built (needs the libecc sources at build time)")
    r = subprocess.run([exe, "vectors"], capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "[-]" not in out and "failed" not in out, out[-4000:]
    assert out.count("[+]") >= 200, out[-2000:]     # 256 cases in this snapshot (the plain-Ed25519 vector file is absent)


def test_eddsa25519_zero_challenge(gpu_ctx):
    """h = 0 mod q (the caller supplies the hash, so this is reachable through the API): [h]A is the point at
    infinity / the Edwards neutral element, and the equation degenerates to 8([S]B - R) = infinity"""
    import oracles as O
    rng = np.random.default_rng(37)
    q = O.ED_Q
    t8 = O.ed_decode(O.ED_TORSION8)
    pubs, sigs, hram = b"", b"", b""
    for i in range(24):
        a_enc, _, _ = O.ed25519_sign(rand_bytes(rng, 32), b"x")
        S = int.from_bytes(rand_bytes(rng, 40), "little") % q
        R = O.ed_mul(S, O.ED_B) if S else (0, 1, 1, 0)
        if i % 4 == 1:
            R = O.ed_add(R, t8)                       # torsion-shifted R: still accepted
        if i % 4 == 2:
            R = O.ed_add(R, O.ED_B)                   # wrong R: rejected
        if i % 4 == 3:
            a_enc = O.ed_encode(O.ed_add(O.ed_decode(a_enc), t8))   # mixed-order key
        h = [0, q, 5 * q, (2**512 - 1) // q * q][i % 4] if i < 20 else [1, q - 1, q + 1, 2**512 - 1][i - 20]
        pubs += a_enc
        sigs += O.ed_encode(R) + S.to_bytes(32, "little")
        hram += h.to_bytes(64, "little")
    exp = Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
    assert exp[:20] == bytes([0, 0, 1, 0] * 5)
    cv = gpu_ctx.curve("WEI25519")
    try:
        assert cv.eddsa_verify(pubs, sigs, hram) == exp
    finally:
        cv.free()


def test_eddsa25519_exceptional_pairs(gpu_ctx):
    """signatures whose cofactored equation holds while one of the reference's two prj_pt_add calls meets its exceptional pair
    (reachable through the API: the caller supplies the hash), with accepted neighbours -- on the Edwards tail of round 4
    (k_ed_tail_c25519: batches of at least ECAMD_COMB_MIN_BATCH items), on the Weierstrass tail it replaces
    (ECAMD_NO_ED_TAIL) and on a batch too small for the comb table; the zero-challenge and edge families through the new tail too"""
    import oracles as O
    from test_ed_tail_model import ed_exceptional_cases
    from test_oracle import ed25519_cases
    rng = np.random.default_rng(43)
    pubs, sigs, hram, kinds = ed_exceptional_cases(rng, n_each=8)
    p2, s2, _, h2 = ed25519_cases(rng, nvalid=8)
    pubs, sigs, hram = pubs + p2, sigs + s2, hram + h2
    n = len(pubs) // 32
    assert n >= 64
    exp = Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
    for k, e in zip(kinds, exp):
        assert e == (1 if k in ("E1", "E2", "E1-wrongS") else 0), k
    # hashes with a huge partial quotient in the Euclidean algorithm on (q, h): k_ed_lat cannot shorten them within its
    # iteration budget and the item takes the full-length window loop (meta bit 1) -- valid signatures under such h (the
    # caller supplies the hash) and their corrupted neighbours
    q = O.ED_Q
    for j, h in enumerate([2**127, 2**127 + 1, 2**128, 2**130 + 5, 2**200, 2**220 - 1, int("8" * 64, 16) % q, q // 2**20, q - 2**127, 2**127 - 1, 5, q - 1]):
        a = int.from_bytes(rand_bytes(rng, 40), "little") % q or 1
        r = int.from_bytes(rand_bytes(rng, 40), "little") % q
        A, R = O.ed_mul(a, O.ED_B), (O.ed_mul(r, O.ED_B) if r else (0, 1, 1, 0))
        if j % 3 == 1:
            A = O.ed_add(A, O.ed_decode(O.ED_TORSION8))
        S = (r + h * a) % q
        for bad in (0, 1):
            pubs += O.ed_encode(A)
            sigs += O.ed_encode(R) + ((S + bad) % q).to_bytes(32, "little")
            hram += (h + q * (j % 2)).to_bytes(64, "little")
    exp = Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
    assert exp[n:] == bytes([0, 1] * 12)
    n = len(pubs) // 32
    cv = gpu_ctx.curve("WEI25519")
    try:
        assert cv.eddsa_verify(pubs, sigs, hram) == exp                      # half-length scalars, Edwards tail
        assert cv.eddsa_verify(pubs[:32 * 20], sigs[:64 * 20], hram[:64 * 20]) == exp[:20]   # below the comb threshold
        for off in ("ECAMD_NO_ED_LATTICE", "ECAMD_NO_ED_TAIL"):
            os.environ[off] = "1"
            try:
                assert cv.eddsa_verify(pubs, sigs, hram) == exp              # full-length loop + Edwards tail; Weierstrass tail
            finally:
                del os.environ[off]
        # a larger tiled batch (several waves, ragged end)
        reps = 4099 // n + 1
        assert cv.eddsa_verify((pubs * reps)[:32 * 4099], (sigs * reps)[:64 * 4099], (hram * reps)[:64 * 4099]) == (exp * reps)[:4099]
    finally:
        cv.free()


def test_host_pipeline_multi_chunk(gpu_ctx):
    """host-pointer entry points split a batch into chunks and overlap the copy of the next chunk with the
    kernels of the current one: a context with a tiny chunk (ECAMD_HOST_CHUNK) must give the same bytes as the
    session context, on ragged batch sizes, for scalar mult, ECDSA verify (incl. exceptional items), Ed25519
    verify and X25519"""
    import libecc_amd
    from test_oracle import ed25519_cases
    rng = np.random.default_rng(38)
    old = {k: os.environ.get(k) for k in ("ECAMD_HOST_CHUNK", "ECAMD_HOST_RAMP_MIN")}
    os.environ["ECAMD_HOST_CHUNK"] = "700"
    os.environ["ECAMD_HOST_RAMP_MIN"] = "100"    # the doubling start of a multi-chunk call (round 6): 100, 200, 400, then 700, and the last 823 as one
    try:
        ctx2 = libecc_amd.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        n = 3 * 700 + 123
        # the producer hook: asked for every range of the input arrays before it is read, in order, exactly once
        asked = []
        ctx2.set_host_ready_hook(lambda first, count: asked.append((first, count)))
        hk = ctx2.curve("SECP256R1")
        try:
            hk.scalar_mult(rand_bytes(rng, 32 * n))
        finally:
            hk.free()
            ctx2.set_host_ready_hook(None)
        assert asked == [(0, 100), (100, 200), (300, 400), (700, 700), (1400, 823)], asked
        a, b = gpu_ctx.curve("SECP256R1"), ctx2.curve("SECP256R1")
        try:
            sc = rand_bytes(rng, 32 * n)
            pa = a.scalar_mult(sc)
            assert b.scalar_mult(sc) == pa and set(pa[1]) == {0}
            sc2 = rand_bytes(rng, 32 * n)
            assert b.scalar_mult(sc2, pa[0]) == a.scalar_mult(sc2, pa[0])
            o, pubs, sigs, dg, hl, _ = make_sigs("SECP256R1", "SHA256", 64, rng)
            reps = n // 64 + 1
            P, S, D = (pubs * reps)[:64 * n], bytearray((sigs * reps)[:64 * n]), (dg * reps)[:32 * n]
            for i in range(0, n, 9):
                S[64 * i + 40] ^= 2
            c = CURVES["SECP256R1"]
            G = c["gx"].to_bytes(32, "big") + c["gy"].to_bytes(32, "big")
            kG, _ = o.scalar_mult((7).to_bytes(32, "big"))
            r = int.from_bytes(kG[:32], "big") % c["q"]
            sv = pow(7, c["q"] - 2, c["q"]) * (2 * r) % c["q"]
            for i in (5, 701, 2100):    # exceptional items (Q = G, u1 == u2) in different chunks
                P = P[:64 * i] + G + P[64 * i + 64:]
                S[64 * i:64 * i + 64] = r.to_bytes(32, "big") + sv.to_bytes(32, "big")
                D = D[:32 * i] + r.to_bytes(32, "big") + D[32 * i + 32:]
            ra = a.ecdsa_verify(P, bytes(S), D, 32)
            assert b.ecdsa_verify(P, bytes(S), D, 32) == ra and 0 in ra and 1 in ra and ra[5] == 0
        finally:
            a.free()
            b.free()
        a, b = gpu_ctx.curve("WEI25519"), ctx2.curve("WEI25519")
        try:
            pubs, sigs, msgs, hram = ed25519_cases(rng, nvalid=10)
            n0 = len(pubs) // 32
            reps = n // n0 + 1
            P, S, H = (pubs * reps)[:32 * n], (sigs * reps)[:64 * n], (hram * reps)[:64 * n]
            ra = a.eddsa_verify(P, S, H)
            assert b.eddsa_verify(P, S, H) == ra and 0 in ra and 1 in ra
            k = rand_bytes(rng, 32 * n)
            base = (9).to_bytes(32, "little") * n
            pa = a.xdh(k, base)
            assert b.xdh(k, base) == pa
            k2 = rand_bytes(rng, 32 * n)
            assert b.xdh(k2, pa[0]) == a.xdh(k2, pa[0])
        finally:
            a.free()
            b.free()
    finally:
        ctx2.close()


@pytest.mark.parametrize("env", ["ECAMD_NO_COMB", "ECAMD_NO_P25519", "ECAMD_NO_X25519_LADDER", "ECAMD_NO_EDWARDS_SMUL",
                                 "ECAMD_NO_FAST_PATH", "ECAMD_NO_ISO", "ECAMD_NO_K256", "ECAMD_NO_P448",
                                 "ECAMD_NO_ED_LATE_MAP", "ECAMD_NO_G448_DECODE", "ECAMD_NO_X448_LADDER", "ECAMD_NO_ED_FIN_G"])
def test_fallback_paths_stay_correct(env):
    """every fast path has a switch that routes around it (A/B measurements, fallbacks): the slower routes
    must give the same bytes -- fixed-base without the comb, WEI25519 on the dense field, X25519 and Ed25519
    through the Weierstrass scalar multiplication, everything on the complete-formula kernel"""
    import libecc_amd
    from test_oracle import ed25519_cases, xdh_edge_inputs
    rng = np.random.default_rng(39)
    old = os.environ.get(env)
    os.environ[env] = "1"
    try:
        ctx = libecc_amd.Context(0)
        try:
            cv = ctx.curve("WEI25519")
            try:
                pubs, sigs, msgs, hram = ed25519_cases(rng, nvalid=6)
                assert cv.eddsa_verify(pubs, sigs, hram) == Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
                ek, eu = xdh_edge_inputs(32, rng)
                assert cv.xdh(ek, eu) == Oracle("WEI25519").xdh(ek, eu)
            finally:
                cv.free()
            cv = ctx.curve("BRAINPOOLP256R1")       # computed on its a = -3 image unless ECAMD_NO_ISO
            try:
                o = Oracle("BRAINPOOLP256R1")
                sc = rand_bytes(rng, 32 * 80)
                pub = cv.scalar_mult(sc)
                assert pub == o.scalar_mult(sc)
                sc2 = rand_bytes(rng, 32 * 80)
                assert cv.scalar_mult(sc2, pub[0]) == o.scalar_mult(sc2, pub[0])
            finally:
                cv.free()
            cv = ctx.curve("WEI448")                # Goldilocks field unless ECAMD_NO_P448
            try:
                o = Oracle("WEI448")
                sc = rand_bytes(rng, 56 * 40)
                pub = cv.scalar_mult(sc)
                assert pub == o.scalar_mult(sc)
                sc2 = rand_bytes(rng, 56 * 40)
                assert cv.scalar_mult(sc2, pub[0]) == o.scalar_mult(sc2, pub[0])
                # X448 and Ed448 verification: front ends / ladder on the Goldilocks unit unless ECAMD_NO_G448_DECODE / _X448_LADDER
                ek, eu = xdh_edge_inputs(56, rng)
                assert cv.xdh(ek, eu) == o.xdh(ek, eu)
                from test_oracle import ed448_cases
                p4, s4, m4, h4 = ed448_cases(rng, 4)
                assert cv.eddsa_verify(p4, s4, h4) == o.eddsa_verify(p4, s4, h4)
            finally:
                cv.free()
            cv = ctx.curve("SECP256K1")             # pseudo-Mersenne field unless ECAMD_NO_K256
            try:
                o = Oracle("SECP256K1")
                sc = rand_bytes(rng, 32 * 80)
                pub = cv.scalar_mult(sc)
                assert pub == o.scalar_mult(sc)
                sc2 = rand_bytes(rng, 32 * 80)
                assert cv.scalar_mult(sc2, pub[0]) == o.scalar_mult(sc2, pub[0])
            finally:
                cv.free()
            cv = ctx.curve("SECP256R1")
            try:
                o = Oracle("SECP256R1")
                sc = rand_bytes(rng, 32 * 96)
                pub = cv.scalar_mult(sc)
                assert pub == o.scalar_mult(sc)
                sc2 = rand_bytes(rng, 32 * 96)
                assert cv.scalar_mult(sc2, pub[0]) == o.scalar_mult(sc2, pub[0])
                _, pubs, sigs, dg, hl, _ = make_sigs("SECP256R1", "SHA256", 64, rng)
                bad = bytearray(sigs)
                bad[64 * 3 + 1] ^= 1
                assert cv.ecdsa_verify(pubs, bytes(bad), dg, hl) == o.ecdsa_verify(pubs, bytes(bad), dg, hl)
            finally:
                cv.free()
        finally:
            ctx.close()
    finally:
        if old is None:
            del os.environ[env]
        else:
            os.environ[env] = old


def test_eddsa448_verify_vs_oracle_and_golden(gpu_ctx):
    """Ed448 batch verification on the WEI448 handle: the reference's RFC 8032 Ed448 / Ed448ph vectors, then valid,
    torsion-shifted and every class of rejected input against the oracle (device- and host-pointer forms)"""
    import torch
    from test_oracle import KAT_EDDSA448, ed448_cases, eddsa448_kat_inputs
    rng = np.random.default_rng(44)
    cv = gpu_ctx.curve("WEI448")
    o = Oracle("WEI448")
    try:
        pubs, sigs, hram = eddsa448_kat_inputs()
        n = len(KAT_EDDSA448)
        assert cv.eddsa_verify(pubs, sigs, hram) == bytes(n)
        bad = bytearray(sigs)
        bad[114 + 3] ^= 1
        assert cv.eddsa_verify(pubs, bytes(bad), hram) == bytes([0, 1] + [0] * (n - 2))
        pubs, sigs, msgs, hram = ed448_cases(rng, nvalid=16)
        exp = o.eddsa_verify(pubs, sigs, hram)
        assert 0 in exp and 1 in exp and exp[:16] == bytes(16)
        assert cv.eddsa_verify(pubs, sigs, hram) == exp
        assert cv.eddsa_verify(b"", b"", b"") == b""
        n = len(exp)
        dev = torch.device("cuda:0")
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        dp, ds, dh = t(pubs), t(sigs), t(hram)
        dr = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        cv.eddsa_verify_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), dr.data_ptr(), None, 114)
        gpu_ctx.synchronize()
        assert bytes(dr.cpu().numpy()) == exp
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "BRAINPOOLP384R1", "WEI25519"])
def test_structured_pub_keys(gpu_ctx, curve):
    """ec_structured_pub_key_import_from_buf as a batch: header, projective import, subgroup check; the affine keys
    it returns feed ECDSA verification"""
    import oracles as O
    from test_oracle import structured_key_cases
    rng = np.random.default_rng(61)
    keys = structured_key_cases(curve, rng, nrand=40)
    exp = O.structured_pub_expect(curve, keys, 1)
    cv = gpu_ctx.curve(curve)
    try:
        assert cv.structured_pub_keys(keys, 1) == exp
        assert cv.structured_pub_keys(keys, 6)[1][0] == 1      # keys made for another algorithm
    finally:
        cv.free()


def test_device_cores_chunked_by_max_chunk(gpu_ctx):
    """device-pointer forms bound their scratch by processing at most max_chunk items at a time: a context with a
    small max_chunk gives the same bytes (Ed25519, Ed448, X25519, ECDSA on a two-scalar-mult curve and on secp256r1)"""
    import torch
    import libecc_amd
    from test_oracle import ed25519_cases, ed448_cases
    rng = np.random.default_rng(40)
    dev = torch.device("cuda:0")
    ctx2 = libecc_amd.Context(0)
    ctx2.set_max_chunk(300)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    try:
        n = 1000
        for curve, gen, kl, sl, hl in (("WEI25519", ed25519_cases, 32, 64, 64), ("WEI448", ed448_cases, 57, 114, 114)):
            pubs, sigs, msgs, hram = gen(rng, 6)
            n0 = len(pubs) // kl
            reps = n // n0 + 1
            P, S, H = (pubs * reps)[:kl * n], (sigs * reps)[:sl * n], (hram * reps)[:hl * n]
            a, b = gpu_ctx.curve(curve), ctx2.curve(curve)
            try:
                exp = a.eddsa_verify(P, S, H)
                dp, ds, dh = t(P), t(S), t(H)
                dr = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                b.eddsa_verify_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), dr.data_ptr(), None, hl)
                ctx2.synchronize()
                assert bytes(dr.cpu().numpy()) == exp and 0 in exp and 1 in exp
                if curve == "WEI25519":
                    k = rand_bytes(rng, 32 * n)
                    pub, st = a.xdh(k, (9).to_bytes(32, "little") * n)
                    exp2 = a.xdh(k[::-1], pub)
                    dk, du = t(k[::-1]), t(pub)
                    do, dst = torch.empty(32 * n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
                    torch.cuda.synchronize()
                    b.xdh_dev(n, dk.data_ptr(), du.data_ptr(), do.data_ptr(), dst.data_ptr(), None)
                    ctx2.synchronize()
                    assert (bytes(do.cpu().numpy()), bytes(dst.cpu().numpy())) == exp2
            finally:
                a.free()
                b.free()
        for curve in ("BRAINPOOLP256R1", "SECP256R1"):
            o, pubs, sigs, dg, hl, _ = make_sigs(curve, "SHA256", 50, rng)
            reps = n // 50
            P, S, D = pubs * reps, bytearray(sigs * reps), dg * reps
            for i in range(0, n, 7):
                S[2 * o.qlen * i + 9] ^= 1
            a, b = gpu_ctx.curve(curve), ctx2.curve(curve)
            try:
                exp = a.ecdsa_verify(P, bytes(S), D, hl)
                dp, ds, dd = t(P), t(bytes(S)), t(D)
                dr = torch.full((n,), 0xAA, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                b.ecdsa_verify_dev(n, dp.data_ptr(), ds.data_ptr(), dd.data_ptr(), hl, dr.data_ptr(), None)
                ctx2.synchronize()
                assert bytes(dr.cpu().numpy()) == exp and 0 in exp and 1 in exp
            finally:
                a.free()
                b.free()
    finally:
        ctx2.close()


def test_ecdsa_crafted_fixture_gpu(gpu_ctx):
    """tests/golden/ecdsa_crafted.json: the verdicts the unmodified reference gave on the crafted ECDSA family (x(R) >= q,
    extreme r, s and digests, invalid twins) must come back from the GPU, item for item"""
    from test_oracle import crafted_fixture
    for curve, h, pubs, sigs, dgs, hl, ref in crafted_fixture():
        cv = gpu_ctx.curve(curve)
        try:
            assert cv.ecdsa_verify(pubs, sigs, dgs, hl) == ref, (curve, h)
        finally:
            cv.free()


def test_edge_fixtures_gpu(gpu_ctx):
    """tests/golden/edge_fixtures.json: the reference's recorded answers on the EdDSA verification, X25519 / X448 and Ed25519
    signing edge families must come back from the GPU byte for byte (same checks as tests/test_oracle.py::test_edge_fixtures)"""
    from test_oracle import check_edge_fixtures
    check_edge_fixtures(gpu_ctx.curve)



@pytest.mark.parametrize("curve", ["SECP384R1", "SECP521R1", "BRAINPOOLP256R1", "SECP224R1", "BRAINPOOLP512R1"])
def test_ecdsa_fused_verify_generic_curves(gpu_ctx, curve):
    """ECDSA verification on the generic radix-2^29 units: batches of >= 4096 items take the fused double-scalar loop
    (k_loop_g for [u2]Q, then k_comb_add_g adds [u1]G from the comb table: sig/ecdsa_common.c:786-796 without the second
    scalar multiplication).  4096 signatures made on the GPU, 10 % corrupted, plus the exceptional families -- key G / -G with u1 == u2 small, so
    that the first comb addition meets the doubling / the inverse case -- against the oracle, and the whole batch against the
    two-multiplication path ($ECAMD_NO_FUSED_VERIFY) and the construction."""
    c = CURVES[curve]
    q, p = c["q"], c["p"]
    cl, ql = clen(curve), qlen(curve)
    rng = np.random.default_rng(77)
    o = Oracle(curve)
    cv = gpu_ctx.curve(curve)
    try:
        n = 4096
        scal = lambda: b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
        d, ks = scal(), scal()
        hl = 32
        dg = rand_bytes(rng, hl * n)
        pubs, st = cv.scalar_mult(d)
        assert set(st) == {0}
        sigs, st = cv.ecdsa_sign(d, ks, dg, hl)
        assert set(st) == {0}
        S = np.frombuffer(sigs, dtype=np.uint8).copy().reshape(n, 2 * ql)
        D = np.frombuffer(dg, dtype=np.uint8).copy().reshape(n, hl)
        P = np.frombuffer(pubs, dtype=np.uint8).copy().reshape(n, 2 * cl)
        i = np.arange(n)
        bad = (i % 10) == 4
        kind = (i // 10) % 4
        S[bad & (kind == 0), ql - 1] ^= 0x02            # r
        S[bad & (kind == 1), 2 * ql - 2] ^= 0x10        # s
        D[bad & (kind == 2), 7] ^= 0x01                 # digest
        P[bad & (kind == 3), 2 * cl - 1] ^= 0x01        # key: off the curve
        # exceptional families (see test_ecdsa_verify_exceptional_pairs): u1 == u2 == k / 2 with the key G or -G
        G = c["gx"].to_bytes(cl, "big") + c["gy"].to_bytes(cl, "big")
        negG = c["gx"].to_bytes(cl, "big") + (p - c["gy"]).to_bytes(cl, "big")
        xp, xs, xd = b"", b"", b""
        hl_x = ql

        def dig_of(e):   # a digest of ql bytes whose leftmost qbits bits are e (bits2int of sig/ecdsa_common.c:404-412)
            return (e << (8 * ql - q.bit_length())).to_bytes(ql, "big")
        for k in (10, 0x2468, 2 * 0x7fff, 2 * 0x8000, 2 * 0x12345, 5, q - 3, (q + 1) // 2):
            kG, st = o.scalar_mult(k.to_bytes(ql, "big"))
            r = int.from_bytes(kG[:cl], "big") % q
            s_ = pow(k, q - 2, q) * (2 * r) % q
            for key, e in ((G, r), (negG, 3 * r % q), (negG, r)):
                xp += key
                xs += r.to_bytes(ql, "big") + s_.to_bytes(ql, "big")
                xd += dig_of(e)
        nx = len(xs) // (2 * ql)
        exp_x = o.ecdsa_verify(xp, xs, xd, hl_x)
        assert exp_x == bytes([0, 0, 1] * (nx // 3))
        # the digests of the two groups have different lengths: two calls, each of >= 4096 items (the exceptional ones tiled)
        got = cv.ecdsa_verify(P.tobytes(), S.tobytes(), D.tobytes(), hl)
        assert got == bytes(bad.astype(np.uint8))
        reps = 4096 // nx + 1
        got_x = cv.ecdsa_verify(xp * reps, xs * reps, xd * reps, hl_x)
        assert got_x == exp_x * reps
        idx = [int(x) for x in rng.choice(n, size=96, replace=False)]
        cut = lambda b, w: b"".join(b[w * j:w * j + w] for j in idx)
        assert o.ecdsa_verify(cut(P.tobytes(), 2 * cl), cut(S.tobytes(), 2 * ql), cut(D.tobytes(), hl), hl) == bytes(got[j] for j in idx)
        os.environ["ECAMD_NO_FUSED_VERIFY"] = "1"
        try:
            assert cv.ecdsa_verify(P.tobytes(), S.tobytes(), D.tobytes(), hl) == got
            assert cv.ecdsa_verify(xp * reps, xs * reps, xd * reps, hl_x) == got_x
        finally:
            del os.environ["ECAMD_NO_FUSED_VERIFY"]
    finally:
        cv.free()
