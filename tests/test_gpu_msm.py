"""Ed25519 whole-batch verification as one multi-scalar multiplication on the GPU:
ec_eddsa_verify_all_batch / ecamd_debug_eddsa_msm against python integers (the combination itself, with the z_i the
device drew), against the oracle's per-item verdicts and against the unmodified reference's ec_verify_batch."""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import oracles as O  # noqa: E402
from oracles import Oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["straus", "bucket"])
def ed_msm_algo(request):
    """every test of this module runs on both evaluations of the combination: the Straus loop (round 2) and the bucket form (round 6);
    ECAMD_ED_MSM_ALGO is read by the library at every call"""
    old = os.environ.get("ECAMD_ED_MSM_ALGO")
    os.environ["ECAMD_ED_MSM_ALGO"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("ECAMD_ED_MSM_ALGO", None)
    else:
        os.environ["ECAMD_ED_MSM_ALGO"] = old


def chacha20_block(key, counter, nonce):
    """RFC 8439 section 2.3 (key 32 bytes, counter u32, nonce 3 x u32) -> 64 bytes"""
    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xffffffff

    def qr(s, a, b, c, d):
        s[a] = (s[a] + s[b]) & 0xffffffff; s[d] = rotl(s[d] ^ s[a], 16)
        s[c] = (s[c] + s[d]) & 0xffffffff; s[b] = rotl(s[b] ^ s[c], 12)
        s[a] = (s[a] + s[b]) & 0xffffffff; s[d] = rotl(s[d] ^ s[a], 8)
        s[c] = (s[c] + s[d]) & 0xffffffff; s[b] = rotl(s[b] ^ s[c], 7)
    init = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)] + \
        [counter] + list(nonce)
    s = list(init)
    for _ in range(10):
        qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15)
        qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14)
    return b"".join(((s[i] + init[i]) & 0xffffffff).to_bytes(4, "little") for i in range(16))


def test_chacha_vector():
    """the python block function on the RFC 8439 2.3.2 vector (it pins the device's z_i below)"""
    key = bytes(range(32))
    out = chacha20_block(key, 1, [0x09000000, 0x4a000000, 0x00000000])
    assert out[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"


def python_combination(pubs, sigs, hram, zs):
    """T = [q - sum z S]B + sum [z h mod q]A + [z]R over python integers; None if an item does not decode"""
    n = len(pubs) // 32
    T = (0, 1, 1, 0)
    ssum = 0
    for i in range(n):
        A = O.ed_decode(pubs[32 * i:32 * i + 32])
        R = O.ed_decode(sigs[64 * i:64 * i + 32])
        if A is None or R is None:
            return None
        S = int.from_bytes(sigs[64 * i + 32:64 * i + 64], "little")
        h = int.from_bytes(hram[64 * i:64 * i + 64], "little") % O.ED_Q
        z = int.from_bytes(zs[16 * i:16 * i + 16], "little")
        ssum = (ssum + z * S) % O.ED_Q
        T = O.ed_add(T, O.ed_mul(z * h % O.ED_Q, A))
        T = O.ed_add(T, O.ed_mul(z, R))
    return O.ed_add(T, O.ed_mul((O.ED_Q - ssum) % O.ED_Q, O.ED_B))


def same_point(P, Q):
    p = O.ED_P
    return (P[0] * Q[2] - Q[0] * P[2]) % p == 0 and (P[1] * Q[2] - Q[1] * P[2]) % p == 0 and P[2] % p and Q[2] % p


def make_items(rng, n, wrong_s=()):
    pubs, sigs, hram = bytearray(), bytearray(), bytearray()
    for i in range(n):
        seed = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
        msg = rng.integers(0, 256, size=24, dtype=np.uint8).tobytes()
        A, sig, h = O.ed25519_sign(seed, msg)
        if i in wrong_s:   # a random S below q: the item fails and the combination is a generic point
            s = int.from_bytes(rng.integers(0, 256, size=40, dtype=np.uint8).tobytes(), "little") % O.ED_Q
            sig = sig[:32] + s.to_bytes(32, "little")
        pubs += A
        sigs += sig
        hram += h
    return bytes(pubs), bytes(sigs), bytes(hram)


@pytest.mark.parametrize("n,k", [(1, 1), (5, 1), (37, 3), (64, 8), (130, 4), (200, 64)])
def test_combination_vs_python(gpu_ctx, n, k):
    """the device's z_i are ChaCha20(seed; item), and the sum it forms -- valid items and items with a wrong S mixed, so
    that it is a generic point -- is the python combination; the accept bit follows [8]T = neutral"""
    rng = np.random.default_rng(1000 + n)
    cv = gpu_ctx.curve("WEI25519")
    try:
        for wrong in ((), tuple(range(0, n, 3))):
            pubs, sigs, hram = make_items(rng, n, wrong)
            seed = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
            gpu_ctx.set_eddsa_msm(2, 0, k)
            acc, zs, T = cv.debug_eddsa_msm(pubs, sigs, hram, seed)
            for i in range(n):
                assert zs[16 * i:16 * i + 16] == chacha20_block(seed, i, [0, 0, 0])[:16], i
            exp = python_combination(pubs, sigs, hram, zs)
            assert same_point(T, exp), (n, k, wrong)
            e8 = O.ed_mul(8, exp)
            assert acc == (e8[0] % O.ED_P == 0 and (e8[1] - e8[2]) % O.ED_P == 0)
            assert acc == (len(wrong) == 0)
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()


@pytest.mark.parametrize("fold", [2, 4, 16])
def test_bucket_reduction_folds(gpu_ctx, fold, ed_msm_algo):
    """$ECAMD_BKT_FOLD (entries per lane and level of the bucket reduction; 8 by default): the device's sum is the python combination for every fold"""
    if ed_msm_algo != "bucket":
        pytest.skip("the Straus evaluation has no bucket reduction")
    old = os.environ.get("ECAMD_BKT_FOLD")
    os.environ["ECAMD_BKT_FOLD"] = str(fold)
    rng = np.random.default_rng(2000 + fold)
    cv = gpu_ctx.curve("WEI25519")
    try:
        n = 70
        for wrong in ((), (1, 33, 69)):
            pubs, sigs, hram = make_items(rng, n, wrong)
            seed = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
            acc, zs, T = cv.debug_eddsa_msm(pubs, sigs, hram, seed)
            assert same_point(T, python_combination(pubs, sigs, hram, zs)), (fold, wrong)
            assert acc == (len(wrong) == 0)
    finally:
        if old is None:
            os.environ.pop("ECAMD_BKT_FOLD", None)
        else:
            os.environ["ECAMD_BKT_FOLD"] = old
        cv.free()


def test_msm_verdict_on_the_edge_families(gpu_ctx):
    """the accept bit over the case families of tests/test_oracle.py (torsion-shifted R and A, non-canonical and undecodable
    encodings, S >= q, small-order keys, R = neutral ...): every subset the per-item oracle accepts is accepted, a subset with
    one rejected item is rejected, and where the reference library is here its ec_verify_batch says the same"""
    from test_oracle import ed25519_cases, eddsa_subset, ED_MSG_LEN
    rng = np.random.default_rng(72)
    pubs, sigs, msgs, hram = ed25519_cases(rng, 10)
    n = len(pubs) // 32
    one = Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
    good = [i for i in range(n) if one[i] == 0]
    bad = [i for i in range(n) if one[i]]
    assert good and bad
    cv = gpu_ctx.curve("WEI25519")
    try:
        for k in (1, 4, 8):
            gpu_ctx.set_eddsa_msm(2, 0, k)

            def run(idx):
                P, S, M, H = eddsa_subset(idx, pubs, sigs, msgs, hram, 32, 64, ED_MSG_LEN, 64)
                seed = hashlib.sha256(bytes(idx[:8]) + bytes([k])).digest()
                acc = cv.debug_eddsa_msm(P, S, H, seed)[0]
                assert cv.eddsa_verify_all(P, S, H)[0] == acc
                if O.have_ref() and k == 1:
                    assert O.ref_eddsa_verify_all(P, S, M, ED_MSG_LEN) == acc, idx
                return acc
            assert run(good)
            assert run(good * 7)
            for g in good:
                assert run([g]), g
            for b in bad:
                assert not run([b]), b
                assert not run(good[:3] + [b] + good[3:]), b
            P, S, M, H = eddsa_subset(good * 3 + [bad[0]] + good, pubs, sigs, msgs, hram, 32, 64, ED_MSG_LEN, 64)
            assert cv.eddsa_verify_all(P, S, H) == (False, 3 * len(good))
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()


def test_msm_large_batch(gpu_ctx):
    """2^ECAMD_TEST_MSM_LOG2 (default 17) signatures, lanes chosen by the library: accepted; one flipped hash bit anywhere
    rejects and the item-by-item pass names the item; pieces of max_chunk items each carry their own combination"""
    import libecc_amd
    log2 = int(os.environ.get("ECAMD_TEST_MSM_LOG2", "17"))
    n = 1 << log2
    rng = np.random.default_rng(5)
    base = 509
    pubs, sigs, hram = make_items(rng, base)
    reps = (n + base - 1) // base
    P, S, H = (pubs * reps)[:32 * n], (sigs * reps)[:64 * n], bytearray((hram * reps)[:64 * n])
    cv = gpu_ctx.curve("WEI25519")
    try:
        gpu_ctx.set_eddsa_msm(2, 0, 0)
        assert cv.eddsa_verify_all(P, S, bytes(H)) == (True, n)
        for idx in (0, n // 3, n - 1):
            H[64 * idx + 5] ^= 4
            assert cv.eddsa_verify_all(P, S, bytes(H)) == (False, idx)
            H[64 * idx + 5] ^= 4
        ctx2 = libecc_amd.Context(0)
        try:
            ctx2.set_max_chunk(n // 4 + 3)
            ctx2.set_eddsa_msm(2, 0, 0)
            c2 = ctx2.curve("WEI25519")
            assert c2.eddsa_verify_all(P, S, bytes(H)) == (True, n)
            H[64 * (n - 2)] ^= 1
            assert c2.eddsa_verify_all(P, S, bytes(H)) == (False, n - 2)
            c2.free()
        finally:
            ctx2.close()
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()


def test_device_pointer_form(gpu_ctx):
    """ec_eddsa_verify_all_batch_dev: device pointers, a caller's stream, only enqueues; verdict byte 0 / 1; pieces of
    max_chunk items share the byte"""
    import torch
    import libecc_amd
    rng = np.random.default_rng(6)
    n = 700
    pubs, sigs, hram = make_items(rng, n)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)

    def t(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    ctx2 = libecc_amd.Context(0)
    try:
        ctx2.set_max_chunk(256)
        cv = ctx2.curve("WEI25519")
        dp, ds, dh = t(pubs), t(sigs), t(hram)
        verdict = torch.full((1,), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        cv.eddsa_verify_all_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), verdict.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        assert int(verdict.item()) == 0
        for bad in (0, 255, 256, n - 1):
            h2 = bytearray(hram)
            h2[64 * bad + 1] ^= 8
            dh2 = t(bytes(h2))
            torch.cuda.synchronize()
            cv.eddsa_verify_all_dev(n, dp.data_ptr(), ds.data_ptr(), dh2.data_ptr(), verdict.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            assert int(verdict.item()) == 1, bad
        # (the WEI448 handle has the form too since round 6: tests/test_gpu_ed448_msm.py; a handle of another curve is refused)
        with pytest.raises(libecc_amd.EcamdError):
            c256 = ctx2.curve("SECP256R1")
            try:
                c256.eddsa_verify_all_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), verdict.data_ptr(), None)
            finally:
                c256.free()
        cv.free()
    finally:
        ctx2.close()


def test_whole_batch_bit_from_projective_keys_and_messages(gpu_ctx):
    """ec_eddsa_verify_msg_prj_all_batch (round 6): ec_verify_batch's one bit from what libsign_amd.so holds -- the key as a projective
    Weierstrass point (random Z), R || S, and the hash input R || <blank for A> || M; the device imports, encodes, hashes and evaluates the
    batch equation.  Valid batch: True; one damaged signature, a damaged message, a key off the curve or at infinity: False (not decided).
    Staged in three chunks.  Signatures from Python integers on the keys [a]G the library computes (as in test_gpu_hash.py)."""
    rng = np.random.default_rng(77)
    n = 600
    c = O.CURVES["WEI25519"]
    p, q = c["p"], c["q"]
    rb = lambda k: rng.integers(0, 256, size=k, dtype=np.uint8).tobytes()
    cv = gpu_ctx.curve("WEI25519")
    old = os.environ.get("ECAMD_HOST_SCHEDULE")
    os.environ["ECAMD_HOST_SCHEDULE"] = "100,200"
    try:
        a = [(int.from_bytes(rb(40), "big") % (q - 1)) + 1 for _ in range(n)]
        r = [(int.from_bytes(rb(40), "big") % (q - 1)) + 1 for _ in range(n)]
        msgs = [rb(20 + i % 9) for i in range(n)]
        Aw, st = cv.scalar_mult(b"".join(x.to_bytes(32, "big") for x in a))
        Rw, st2 = cv.scalar_mult(b"".join(x.to_bytes(32, "big") for x in r))
        assert set(st) == {0} and set(st2) == {0}

        def prj(aff, i, lam):
            x, y = int.from_bytes(aff[64 * i:64 * i + 32], "big"), int.from_bytes(aff[64 * i + 32:64 * i + 64], "big")
            return b"".join((v % p).to_bytes(32, "big") for v in (x * lam, y * lam, lam))
        keys = b"".join(prj(Aw, i, 1 if i % 3 == 0 else int.from_bytes(rb(40), "big") % (p - 1) + 1) for i in range(n))
        Aenc, est = cv.eddsa_encode_points(keys)
        Renc, est2 = cv.eddsa_encode_points(b"".join(prj(Rw, i, 1) for i in range(n)))
        assert set(est) == {0} and set(est2) == {0}
        stride = (4 + 64 + 28 + 3) & ~3
        sigs, slots = bytearray(), bytearray(stride * n)
        for i in range(n):
            Ri, Ai = Renc[32 * i:32 * i + 32], Aenc[32 * i:32 * i + 32]
            hd = hashlib.sha512(Ri + Ai + msgs[i]).digest()
            sigs += Ri + ((r[i] + (int.from_bytes(hd, "little") % q) * a[i]) % q).to_bytes(32, "little")
            inp = Ri + bytes(32) + msgs[i]                      # the blank the device fills
            slots[stride * i:stride * i + 4] = len(inp).to_bytes(4, "little")
            slots[stride * i + 4:stride * i + 4 + len(inp)] = inp
        sigs, slots = bytes(sigs), bytes(slots)
        assert cv.eddsa_verify_msg_prj_all(keys, sigs, slots, stride, 32)
        for k in (0, 299, n - 1):
            bad = bytearray(sigs)
            bad[64 * k + 40] ^= 2
            assert not cv.eddsa_verify_msg_prj_all(keys, bytes(bad), slots, stride, 32)
            bs = bytearray(slots)
            bs[stride * k + 4 + 64 + 5] ^= 1
            assert not cv.eddsa_verify_msg_prj_all(keys, sigs, bytes(bs), stride, 32)
        bk = bytearray(keys)
        bk[96 * 7 + 31] ^= 1                                    # off the curve
        assert not cv.eddsa_verify_msg_prj_all(bytes(bk), sigs, slots, stride, 32)
        bk = bytearray(keys)
        bk[96 * 9:96 * 10] = bytes(32) + (1).to_bytes(32, "big") + bytes(32)   # the point at infinity
        assert not cv.eddsa_verify_msg_prj_all(bytes(bk), sigs, slots, stride, 32)
    finally:
        if old is None:
            os.environ.pop("ECAMD_HOST_SCHEDULE", None)
        else:
            os.environ["ECAMD_HOST_SCHEDULE"] = old
        cv.free()
