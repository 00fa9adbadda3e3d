"""GPU tests of the wire formats around the hot path: point decompression (aff_pt_y_from_x with the
reference's Tonelli-Shanks root order), structured signatures and structured private keys -> key pairs, each against the
unmodified reference binary, the restatement oracle and the recorded reference answers."""
import json
import os

import numpy as np
import pytest

import oracles as O
from oracles import CURVES, GOLDEN, Oracle, have_ref
from test_gpu_parity import rand_bytes

pytestmark = pytest.mark.gpu

DECOMP = json.load(open(os.path.join(GOLDEN, "decompress_fixture.json")))


@pytest.mark.parametrize("curve", sorted(DECOMP))
def test_y_from_x_recorded_reference_answers(gpu_ctx, curve):
    f = DECOMP[curve]
    cv = gpu_ctx.curve(curve)
    try:
        y1, y2, st = cv.y_from_x(bytes.fromhex(f["x"]))
        assert (y1.hex(), y2.hex(), st.hex()) == (f["y1"], f["y2"], f["status"])
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP224R1", "SECP256R1", "SECP256K1", "BRAINPOOLP320R1", "SECP384R1", "SECP521R1", "WEI25519", "WEI448"])
def test_y_from_x_and_decompression(gpu_ctx, curve):
    """4096 random x (about half of them on the curve) + edge values against the oracle and the reference binary; SEC 1
    decompression of compressed valid points gives the points back, bad prefixes / x >= p / non-residues are errors"""
    rng = np.random.default_rng(81)
    o = Oracle(curve)
    c = CURVES[curve]
    p, cl = c["p"], o.clen
    n = 4096
    raw = rng.integers(0, 256, size=(n, cl + 8), dtype=np.uint8)
    vals = [int.from_bytes(raw[i].tobytes(), "big") % p for i in range(n)]
    vals[:6] = [0, 1, p - 1, p, min(p + 5, (1 << (8 * cl)) - 1), c["gx"]]
    xs = b"".join(v.to_bytes(cl, "big") for v in vals)
    cv = gpu_ctx.curve(curve)
    try:
        got = cv.y_from_x(xs)
        assert got == o.y_from_x(xs)
        if have_ref():
            k = 512
            assert (got[0][:cl * k], got[1][:cl * k], got[2][:k]) == O.ref_y_from_x(curve, xs[:cl * k])
        assert 0.3 * n < got[2].count(0) < 0.7 * n
        # compress -> decompress
        sc = rand_bytes(rng, o.qlen * 600)
        pts, st = cv.scalar_mult(sc)
        pts = b"".join(pts[2 * cl * i:2 * cl * (i + 1)] for i in range(600) if st[i] == 0)
        m = len(pts) // (2 * cl)
        comp = b"".join(bytes([2 + (pts[2 * cl * i + 2 * cl - 1] & 1)]) + pts[2 * cl * i:2 * cl * i + cl] for i in range(m))
        out, st = cv.decompress(comp)
        assert st == bytes(m) and out == pts
        # the other parity gives the negated point; prefixes other than 02 / 03 and x >= p are errors
        flip = bytearray(comp)
        for i in range(m):
            flip[(cl + 1) * i] ^= 1
        out2, st2 = cv.decompress(bytes(flip))
        assert st2 == bytes(m)
        for i in range(0, m, 37):
            y = int.from_bytes(pts[2 * cl * i + cl:2 * cl * (i + 1)], "big")
            assert int.from_bytes(out2[2 * cl * i + cl:2 * cl * (i + 1)], "big") == (p - y) % p
        bad = bytearray(comp[:(cl + 1) * 8])
        bad[0] = 4
        bad[cl + 1] = 0
        bad[2 * (cl + 1)] = 0x82
        bad[3 * (cl + 1) + 1:4 * (cl + 1)] = p.to_bytes(cl, "big")
        out3, st3 = cv.decompress(bytes(bad))
        assert list(st3[:4]) == [1, 1, 1, 1] and list(st3[4:]) == [0] * 4 and out3[:4 * 2 * cl] == bytes(4 * 2 * cl)
    finally:
        cv.free()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
@pytest.mark.parametrize("curve,alg", [("SECP256R1", 1), ("BRAINPOOLP384R1", 1), ("SECP521R1", 14)])
def test_structured_private_keys_to_key_pairs(gpu_ctx, curve, alg):
    """ec_structured_key_pair_import_from_priv_key_buf in batch: header checks, x < q, Y = [x]G -- private scalars, public
    keys (compared in affine form) and status against the unmodified reference"""
    rng = np.random.default_rng(82)
    o = Oracle(curve)
    c = CURVES[curve]
    q, ql, cl = c["q"], o.qlen, o.clen
    ctype = O.curve_type(curve)
    n = 300
    xs = [int.from_bytes(rand_bytes(rng, ql + 8), "big") % q for _ in range(n)]
    xs[:6] = [0, 1, q - 1, q, q + 1, (1 << (8 * ql)) - 1]
    keys = bytearray()
    for i, x in enumerate(xs):
        keys += bytes([1, alg, ctype]) + x.to_bytes(ql, "big")
    klen = 3 + ql
    keys[10 * klen] = 0            # EC_PUBKEY where EC_PRIVKEY is expected
    keys[11 * klen + 1] ^= 2       # another algorithm
    keys[12 * klen + 2] ^= 1       # another curve
    cv = gpu_ctx.curve(curve)
    try:
        pv, pb, st = cv.structured_key_pairs(bytes(keys), klen, alg)
        rpv, rpb, rst = O.ref_structured_key_pairs(curve, bytes(keys), klen, alg)
        assert st == rst and pv == rpv
        assert list(st[:6]) == [2, 0, 0, 1, 1, 1] and list(st[10:13]) == [1, 1, 1]
        for i in range(n):
            rec = pb[3 * cl * i:3 * cl * (i + 1)]
            if st[i] == 0:
                assert rec[:2 * cl] == rpb[2 * cl * i:2 * cl * (i + 1)] and int.from_bytes(rec[2 * cl:], "big") == 1
            elif st[i] == 2:
                assert int.from_bytes(rec[:cl], "big") == 0 and int.from_bytes(rec[2 * cl:], "big") == 0 and any(rec[cl:2 * cl])
        # the exported keys verify what the exported scalars sign
        dg = rand_bytes(rng, 32 * n)
        ks = b"".join((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1) + 1).to_bytes(ql, "big") for _ in range(n))
        sigs, sst = cv.ecdsa_sign(pv, ks, dg, 32)
        res = cv.ecdsa_verify_fmt(pb, 1, sigs, dg, 32)
        for i in range(n):
            if st[i] == 0 and sst[i] == 0:
                assert res[i] == 0
    finally:
        cv.free()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
def test_structured_signatures(gpu_ctx):
    """the 3 header bytes of ec_structured_sig_export_to_buf (algorithm, hash, curve) checked and stripped, against what
    ec_structured_sig_import_from_buf hands back"""
    rng = np.random.default_rng(83)
    curve = "SECP256R1"
    ctype = O.curve_type(curve)
    n, sl = 64, 64
    recs = bytearray()
    for i in range(n):
        recs += bytes([1, 2, ctype]) + rand_bytes(rng, sl)      # ECDSA = 1, SHA256 = 2 in libecc's enums
    recs[5 * (sl + 3)] = 3
    recs[6 * (sl + 3) + 1] = 5
    recs[7 * (sl + 3) + 2] ^= 1
    cv = gpu_ctx.curve(curve)
    try:
        raw, st = cv.structured_sigs(bytes(recs), sl + 3, 1, 2)
        rraw, hdr, rst = O.ref_structured_sigs(bytes(recs), sl + 3)
        for i in range(n):
            same = rst[i] == 0 and hdr[3 * i:3 * i + 3] == bytes([1, 2, ctype])
            assert st[i] == (0 if same else 1)
            if same:
                assert raw[sl * i:sl * (i + 1)] == rraw[sl * i:sl * (i + 1)]
        assert list(st[5:8]) == [1, 1, 1] and st.count(1) == 3
    finally:
        cv.free()


def test_eddsa448_sign_steps(gpu_ctx):
    """Ed448 signing as the two device-side steps around the caller's SHAKE256 hashes (the split Ed25519 signing uses):
    R = encode([r / 4]G through the 4-isogeny), S = (r + h a) mod q -- signature bytes equal those of an RFC 8032 signer
    (python integers) and of the unmodified reference's ec_sign (EDDSA448) from the same seeds; the signatures verify on the
    GPU; edge values of the hashes (r = 0 mod q: the neutral element; all ones) go through"""
    import hashlib
    rng = np.random.default_rng(84)
    n, ml = 64, 21
    seeds = [rand_bytes(rng, 57) for _ in range(n)]
    msgs = [rand_bytes(rng, ml) for _ in range(n)]
    H = lambda x: hashlib.shake_256(x).digest(114)
    dom = O.ed_dom4(0, b"")
    A, sigs, a_all, rh = b"", b"", b"", b""
    for sd, m in zip(seeds, msgs):
        hk = H(sd)
        ab = bytearray(hk[:57])
        ab[0] &= 0xFC
        ab[55] |= 0x80
        ab[56] = 0
        a_all += bytes(ab)
        rh += H(dom + hk[57:] + m)
        pk, sg, _ = O.ed448_sign(sd, m)
        A += pk
        sigs += sg
    cv = gpu_ctx.curve("WEI448")
    try:
        R, st = cv.eddsa_sign_R(rh)
        assert st == bytes(n)
        assert R == b"".join(sigs[114 * i:114 * i + 57] for i in range(n))
        hram = b"".join(H(dom + R[57 * i:57 * (i + 1)] + A[57 * i:57 * (i + 1)] + msgs[i]) for i in range(n))
        S = cv.eddsa_sign_S(rh, hram, a_all)
        assert S == b"".join(sigs[114 * i + 57:114 * (i + 1)] for i in range(n))
        full = b"".join(R[57 * i:57 * (i + 1)] + S[57 * i:57 * (i + 1)] for i in range(n))
        assert cv.eddsa_verify(A, full, hram) == bytes(n)
        if have_ref():
            rp, rs, rst = O.ref_ed448_sign(b"".join(seeds), b"".join(msgs), ml)
            assert set(rst) == {0} and rp == A and rs == full
        # edge hashes: r = 0 mod q encodes the neutral element (y = 1); r = q, 2^912 - 1
        q = O.E4_Q
        edge = [(0).to_bytes(114, "little"), q.to_bytes(114, "little"), b"\xff" * 114, (q + 1).to_bytes(114, "little"),
                (4).to_bytes(114, "little")]
        R2, st2 = cv.eddsa_sign_R(b"".join(edge))
        assert st2 == bytes(len(edge))
        neutral = (1).to_bytes(57, "little")
        assert R2[:57] == neutral and R2[57:114] == neutral
        exp = [O.e4_encode(O.e4_mul(int.from_bytes(e, "little") % q, O.E4_B)) for e in edge]
        assert R2 == b"".join(exp)
    finally:
        cv.free()


@pytest.mark.gpu
def test_eddsa_encode_point_batch(gpu_ctx):
    """ec_eddsa_encode_point_batch = eddsa_export_pub_key in batch: projective Weierstrass points of WEI25519 -> the RFC 8032
    encodings (python integers: [a]B encoded), scaled representatives included; the point at infinity encodes the neutral
    element, a point off the curve is an error"""
    import oracles as O
    rng = np.random.default_rng(31)
    n = 70
    p = O.ED_P
    cv = gpu_ctx.curve("WEI25519")
    try:
        scal = [int.from_bytes(rng.integers(0, 256, size=32, dtype=np.uint8).tobytes(), "little") % O.ED_Q or 1 for _ in range(n)]
        aff, st = cv.scalar_mult(b"".join(a.to_bytes(32, "big") for a in scal))
        assert set(st) == {0}
        prj = bytearray()
        for i in range(n):
            x, y = int.from_bytes(aff[64 * i:64 * i + 32], "big"), int.from_bytes(aff[64 * i + 32:64 * i + 64], "big")
            z = int.from_bytes(rng.integers(0, 256, size=40, dtype=np.uint8).tobytes(), "big") % p or 1 if i % 2 else 1
            prj += (x * z % p).to_bytes(32, "big") + (y * z % p).to_bytes(32, "big") + z.to_bytes(32, "big")
        prj += bytes(32) + (1).to_bytes(32, "big") + bytes(32)                         # (0 : 1 : 0)
        prj += (5).to_bytes(32, "big") + (7).to_bytes(32, "big") + (1).to_bytes(32, "big")   # not on the curve
        enc, st = cv.eddsa_encode_points(bytes(prj))
        for i in range(n):
            assert enc[32 * i:32 * i + 32] == O.ed_encode(O.ed_mul(scal[i], O.ED_B)), i
        assert list(st) == [0] * (n + 1) + [1]
        assert enc[32 * n:32 * n + 32] == (1).to_bytes(32, "little")
    finally:
        cv.free()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["SECP256R1", "WEI25519", "SECP384R1", "BRAINPOOLP256R1", "SECP521R1"])
def test_group_law_and_unprotected_mult(gpu_ctx, curve):
    """ec_prj_pt_op_batch_fmt (prj_pt_add / prj_pt_dbl / prj_pt_is_on_curve / prj_pt_neg / prj_pt_cmp / prj_pt_eq_or_opp) and ec_prj_pt_unprotected_mult_batch
    (_prj_pt_unprotected_mult statement for statement; one scalar for all = check_prj_pt_order) against the oracle -- itself pinned
    on the unmodified reference for exactly these cases, tests/test_oracle.py -- in both wire formats: exceptional pairs of the
    cofactor curve, infinity in its spellings, (0 : 0 : 0), off-curve and out-of-range triples, P + P and P - P through the addition"""
    from test_oracle import group_law_cases
    from oracles import clen
    rng = np.random.default_rng(61)
    o = Oracle(curve)
    cv = gpu_ctx.curve(curve)
    cl = clen(curve)
    try:
        p1, p2, scal, slen = group_law_cases(curve, rng, n=40)
        n = len(p1) // (3 * cl)
        for op in (0, 1, 2, 3, 4, 5):       # add, dbl, on-curve, neg, cmp, eq_or_opp
            for out_fmt in (0, 1):
                assert cv.pt_op_fmt(op, p1, p2, 1, out_fmt) == o.pt_op_fmt(op, p1, p2, 1, out_fmt), (op, out_fmt)
        cmpb = cv.pt_op_fmt(4, p1, p2, 1, 0)[0]
        eqb = cv.pt_op_fmt(5, p1, p2, 1, 0)[0]
        assert 0 in cmpb and 1 in cmpb and 0 in eqb and 1 in eqb
        aff1, s1 = o.pt_op_fmt(1, p1, None, 1, 0)
        keep = [i for i in range(n) if s1[i] == 0]
        a1 = b"".join(aff1[2 * cl * i:2 * cl * (i + 1)] for i in keep)
        a2 = b"".join(aff1[2 * cl * i:2 * cl * (i + 1)] for i in reversed(keep))
        for op in (0, 1, 2, 3, 4, 5):
            assert cv.pt_op_fmt(op, a1, a2, 0, 1) == o.pt_op_fmt(op, a1, a2, 0, 1)
        # the old affine-only entry points are the same operation
        assert cv.pt_add(a1, a2) == o.pt_op_fmt(0, a1, a2, 0, 0)
        got = cv.unprotected_mult(scal, slen, p1, 1, 1)
        assert got == o.unprotected_mult(scal, slen, p1, 1, 1)
        assert 0 in got[1] and 1 in got[1] and 2 in got[1]
        for k in (CURVES[curve]["q"], CURVES[curve]["order"], 8, 0):
            kb = k.to_bytes(slen, "big")
            assert cv.unprotected_mult(kb, slen, p1, 1, 0, broadcast=True) == o.unprotected_mult(kb, slen, p1, 1, 0, broadcast=True)
        # a larger batch (several waves): the cases tiled
        reps = 300 // n + 1
        P1, P2, SC = p1 * reps, p2 * reps, scal * reps
        exp = o.pt_op_fmt(0, p1, p2, 1, 1)
        assert cv.pt_op_fmt(0, P1, P2, 1, 1) == (exp[0] * reps, exp[1] * reps)
        for op in (3, 4, 5):
            exp = o.pt_op_fmt(op, p1, p2, 1, 1)
            assert cv.pt_op_fmt(op, P1, P2, 1, 1) == (exp[0] * reps, exp[1] * reps)
        exp = o.unprotected_mult(scal, slen, p1, 1, 0)
        assert cv.unprotected_mult(SC, slen, P1, 1, 0) == (exp[0] * reps, exp[1] * reps)
        assert cv.pt_op_fmt(0, b"", b"", 1, 1) == (b"", b"")
    finally:
        cv.free()
