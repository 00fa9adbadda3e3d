"""CPU tests behind k_ed_tail_c25519 (round 4: the tail of an Ed25519 verification on the Edwards curve).

1. The characterisation the kernel rests on: RCB15 Algorithm 1 (prj_pt_add, curves/prj_pt.c:971-1071) on WEI25519 -- a curve of
   even order -- returns (0 : 0 : 0), which libecc turns into -1 (:1058-1060), exactly when the DIFFERENCE of its two inputs is the
   point of order two, and the correct sum otherwise.  Checked on every torsion coset, with infinity and equal / opposite inputs.
2. The decision procedure of the kernel, restated over Python integers (E1, E2, [8]W2 == neutral), against the oracle's
   restatement of the reference (two prj_pt_add calls with their -1) on valid, invalid, torsion-shifted and CRAFTED inputs: the
   batch verifiers take the hash from the caller, so an exceptional pair together with a valid equation is reachable through the
   API (h chosen freely), and only the restated failures reject those.

ed_exceptional_cases() is shared with the GPU test (tests/test_gpu_parity.py::test_eddsa25519_exceptional_pairs)."""
import numpy as np

import oracles as O
from oracles import Oracle

P = O.ED_P
Q = O.ED_Q
A_M = 486662
inv = lambda x: pow(x % P, P - 2, P)


def _wei25519():
    a3 = A_M * inv(3) % P
    a = (3 - A_M * A_M) * inv(3) % P
    b = (2 * A_M**3 - 9 * A_M) * inv(27) % P
    c = pow(-(A_M + 2) % P, (P + 3) // 8, P)
    if c * c % P != -(A_M + 2) % P:
        c = c * O.ED_I % P
    assert c * c % P == -(A_M + 2) % P
    return a3, a, b, c


def _rcb_add(Pt, Qt, a, b):
    X1, Y1, Z1 = Pt
    X2, Y2, Z2 = Qt
    b3 = 3 * b
    t0, t1, t2 = X1 * X2 % P, Y1 * Y2 % P, Z1 * Z2 % P
    t3 = ((X1 + Y1) * (X2 + Y2) - t0 - t1) % P
    t4 = ((X1 + Z1) * (X2 + Z2) - t0 - t2) % P
    t5 = ((Y1 + Z1) * (Y2 + Z2) - t1 - t2) % P
    z3 = (b3 * t2 + a * t4) % P
    x3, z3 = (t1 - z3) % P, (t1 + z3) % P
    y3 = x3 * z3 % P
    t1 = (3 * t0 + a * t2) % P
    t4 = (b3 * t4 + a * (t0 - a * t2)) % P
    return ((t3 * x3 - t5 * t4) % P, (y3 + t1 * t4) % P, (t5 * z3 + t3 * t1) % P)


def _aff(Pt):
    zi = inv(Pt[2])
    return Pt[0] * zi % P, Pt[1] * zi % P


def _to_w(Pt, a3, alpha):
    x, y = _aff(Pt)
    if (x, y) == (0, 1):
        return (0, 1, 0)
    if (x, y) == (0, P - 1):
        return (a3, 0, 1)
    u = (1 + y) * inv(1 - y) % P
    return ((u + a3) % P, alpha * u * inv(x) % P, 1)


def _neg(Pt):
    return ((-Pt[0]) % P, Pt[1], Pt[2], (-Pt[3]) % P)


def _rand_point(rng):
    while True:
        enc = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        enc = enc[:31] + bytes([enc[31] & 0x7f])
        try:
            Pt = O.ed_decode(enc)
        except Exception:
            continue
        if Pt is not None and Pt[0] != 0:
            return Pt


def test_rcb_addition_fails_exactly_when_the_difference_has_order_two():
    a3, a, b, alpha = _wei25519()
    rng = np.random.default_rng(41)
    t8 = O.ed_decode(O.ED_TORSION8)
    tors = [O.ed_mul(k, t8) if k else (0, 1, 1, 0) for k in range(8)]
    T2 = (0, P - 1)
    assert T2 in [_aff(t) for t in tors]
    hits = 0
    for trial in range(40):
        base = _rand_point(rng) if trial % 5 else tors[trial % 8]
        for t in tors:
            for U in (base, (0, 1, 1, 0)):
                V = O.ed_add(U, t)
                for (X, Y) in ((U, V), (V, U)):
                    R = _rcb_add(_to_w(X, a3, alpha), _to_w(Y, a3, alpha), a, b)
                    exceptional = R[1] == 0 and R[2] == 0
                    assert exceptional == (_aff(O.ed_add(X, _neg(Y))) == T2)
                    if exceptional:
                        assert R[0] == 0
                        hits += 1
                    else:
                        S = _to_w(O.ed_add(X, Y), a3, alpha)
                        assert (R[0] * S[2] - S[0] * R[2]) % P == 0 and (R[1] * S[2] - S[1] * R[2]) % P == 0
    assert hits > 0


def ed_exceptional_cases(rng, n_each=6):
    """(pubs, sigs, hram): signatures whose cofactored EQUATION holds -- so an implementation that only evaluates the group
    equation accepts them -- while one of the reference's two prj_pt_add calls meets its exceptional pair and returns -1:
      E1  R = T2 - [S]B and h = 2 S / a:   [S]G + R = T2 at the first addition;
      E2  h = 0 mod q ([h mod q]A = infinity whatever the key) and R = [S]B + T2:  W1 = T2 meets infinity at the second addition;
    mixed with their accepted neighbours (the same constructions shifted by another torsion point, plain valid ones)."""
    t8 = O.ed_decode(O.ED_TORSION8)
    T2 = O.ed_mul(4, t8)
    pubs, sigs, hram, kinds = b"", b"", b"", []

    def rnd(k):
        return int.from_bytes(bytes(rng.integers(0, 256, k, dtype=np.uint8)), "little")

    def emit(A, R, S, h, kind):
        nonlocal pubs, sigs, hram
        pubs += O.ed_encode(A)
        sigs += O.ed_encode(R) + S.to_bytes(32, "little")
        hram += h.to_bytes(64, "little")
        kinds.append(kind)

    for i in range(n_each):
        a = rnd(40) % Q or 1
        A = O.ed_mul(a, O.ED_B)
        if i % 2:
            A = O.ed_add(A, t8)                              # mixed-order key
        S = rnd(40) % Q or 1
        SB = O.ed_mul(S, O.ED_B)
        h = 2 * S * pow(a, Q - 2, Q) % Q + Q * (i % 3)        # any representative of h mod q
        # E1 with a valid equation: 8 (SB - R - hA) = 8 (2 SB - T2 - 2 S B) = neutral
        emit(A, O.ed_add(T2, _neg(SB)), S, h, "E1")
        # the neighbour shifted by a point of order four instead: no exceptional pair, equation still holds: accepted
        emit(A, O.ed_add(O.ed_mul(2, t8), _neg(SB)), S, h, "E1-shift4")
        # the same R with another S: plain rejection
        emit(A, O.ed_add(T2, _neg(SB)), (S + 1) % Q, h, "E1-wrongS")
    for i in range(n_each):
        a = rnd(40) % Q or 1
        A = O.ed_mul(a, O.ED_B)
        if i % 2:
            A = O.ed_add(A, t8)
        S = rnd(40) % Q
        SB = O.ed_mul(S, O.ED_B) if S else (0, 1, 1, 0)
        h = Q * (i + 1)                                       # the verifier multiplies by h mod q = 0: [h]A = infinity for every key
        # E2 with a valid equation: W1 = SB - R = T2 meets [h]A = infinity at the second addition, 8 (W1 - 0) = neutral
        emit(A, O.ed_add(SB, T2), S, h, "E2")
        # neighbours: a shift by a point of order four (no exceptional pair, the equation still holds) and none at all
        emit(A, O.ed_add(SB, O.ed_mul(2, t8)), S, h, "E2-shift4")
        emit(A, SB, S, h, "E2-plain")
    for i in range(n_each):
        seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
        a_enc, sg, hr = O.ed25519_sign(seed, b"m%d" % i)
        pubs += a_enc
        sigs += sg
        hram += hr
        kinds.append("valid")
    return pubs, sigs, hram, kinds


def _tail_model(pub, sig, hr):
    """the decision procedure of k_ed_decode_ed_c25519 + k_ed_scal + k_ed_tail_c25519 over Python integers; 0 accept / 1 reject"""
    def dec(enc):
        y = int.from_bytes(enc, "little")
        sign, y = y >> 255, y & ((1 << 255) - 1)
        if y >= P:
            return None
        u, v = (1 - y * y) % P, (-1 - O.ED_D * y * y) % P
        x = pow(u * inv(v) % P, (P + 3) // 8, P)
        if (v * x * x - u) % P:
            x = x * O.ED_I % P
        if (v * x * x - u) % P:
            return None
        if x == 0:
            return "neutral" if (y == 1 and sign == 0) else None
        if (x & 1) != sign:
            x = P - x
        return (x, y, 1, x * y % P)
    A, R = dec(pub), dec(sig[:32])
    S = int.from_bytes(sig[32:], "little")
    if A is None or A == "neutral" or R is None or S >= Q:
        return 1
    if _aff(O.ed_mul(8, A)) == (0, 1):
        return 1
    if R == "neutral":
        R = (0, 1, 1, 0)
    h = int.from_bytes(hr, "little") % Q
    SB = O.ed_mul(S, O.ED_B) if S else (0, 1, 1, 0)
    hA = O.ed_mul(h, A) if h else (0, 1, 1, 0)
    xr, yr = _aff(R)
    e1 = _aff(SB) == (xr, (-yr) % P)
    W1 = O.ed_add(SB, _neg(R))
    xh, yh = _aff(hA)
    e2 = _aff(W1) == (xh, (-yh) % P)
    W2 = O.ed_mul(8, O.ed_add(W1, _neg(hA)))
    return 0 if (not e1 and not e2 and _aff(W2) == (0, 1)) else 1


def test_tail_model_matches_the_oracle():
    from test_oracle import ed25519_cases
    rng = np.random.default_rng(42)
    o = Oracle("WEI25519")
    pubs, sigs, hram, kinds = ed_exceptional_cases(rng)
    exp = o.eddsa_verify(pubs, sigs, hram)
    by_kind = {}
    for k, e in zip(kinds, exp):
        by_kind.setdefault(k, set()).add(e)
    # the constructions do what they say: exceptional pairs reject although the equation holds, their neighbours are accepted
    assert by_kind["E1"] == {1} and by_kind["E2"] == {1} and by_kind["E1-wrongS"] == {1}
    assert by_kind["E1-shift4"] == {0} and by_kind["E2-shift4"] == {0} and by_kind["E2-plain"] == {0} and by_kind["valid"] == {0}
    n = len(kinds)
    got = bytes(_tail_model(pubs[32 * i:32 * i + 32], sigs[64 * i:64 * i + 64], hram[64 * i:64 * i + 64]) for i in range(n))
    assert got == exp
    p2, s2, _, h2 = ed25519_cases(rng, nvalid=6)
    n2 = len(p2) // 32
    got = bytes(_tail_model(p2[32 * i:32 * i + 32], s2[64 * i:64 * i + 64], h2[64 * i:64 * i + 64]) for i in range(n2))
    assert got == o.eddsa_verify(p2, s2, h2)
