"""BASELINE.json configs[3] and [4] at their full size (2^20 items in one call) -- ECDSA verification on secp256r1,
Ed25519 verification and X25519 -- with (i) size-independent properties over every item and (ii) 2^16 items of each
result (SURVEY.md section 8d's sample size; $ECAMD_TEST_REF_ITEMS) compared with the UNMODIFIED reference binary
(oracle/_ref: ec_verify, ec_sign, x25519 of libecc itself, run on all host threads), not with the restatement.  configs[1] / [2] at full size are tests/test_gpu_parity.py::test_full_batch_properties."""
import hashlib
import os

import numpy as np
import pytest

import oracles as O
from oracles import CURVES, RefLib, have_ref
from test_gpu_parity import rand_bytes

pytestmark = pytest.mark.gpu

LOG2N = int(os.environ.get("ECAMD_TEST_FULL_LOG2", "20"))
NREF = int(os.environ.get("ECAMD_TEST_REF_ITEMS", str(1 << 16)))


def cut(b, w, idx):
    return b"".join(b[w * i:w * i + w] for i in idx)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
def test_ecdsa_secp256r1_full_size_vs_reference_binary(gpu_ctx):
    """configs[3]: 2^20 (public key, signature, digest) triples, 10 % corrupted in r, s, the digest or the key.
    Every valid item is accepted and every corrupted one rejected; 2^12 items against libecc's ec_verify (which hashes the
    messages itself) and 2^11 signatures against libecc's ec_sign with the same nonces."""
    curve, h, ml = "SECP256R1", "SHA256", 24
    rng = np.random.default_rng(401)
    cv = gpu_ctx.curve(curve)
    r = RefLib(curve)
    try:
        n, q = 1 << LOG2N, CURVES[curve]["q"]
        raw = rng.integers(0, 256, size=(2, n, 40), dtype=np.uint8)
        scal = lambda a: b"".join(((int.from_bytes(a[i].tobytes(), "big") % (q - 1)) + 1).to_bytes(32, "big") for i in range(n))
        d, ks = scal(raw[0]), scal(raw[1])
        msgs = rand_bytes(rng, ml * n)
        dg = b"".join(hashlib.sha256(msgs[ml * i:ml * (i + 1)]).digest() for i in range(n))
        pubs, st = cv.scalar_mult(d)
        assert set(st) == {0}
        sigs, st = cv.ecdsa_sign(d, ks, dg, 32)
        assert set(st) == {0}
        assert cv.ecdsa_verify(pubs, sigs, dg, 32) == bytes(n)
        # corrupt 10 %: r, s, digest (= another message), key
        S = np.frombuffer(sigs, dtype=np.uint8).copy().reshape(n, 64)
        D = np.frombuffer(dg, dtype=np.uint8).copy().reshape(n, 32)
        M = np.frombuffer(msgs, dtype=np.uint8).copy().reshape(n, ml)
        P = np.frombuffer(pubs, dtype=np.uint8).copy().reshape(n, 64)
        i = np.arange(n)
        bad = (i % 10) == 7
        kind = (i // 10) % 4
        S[bad & (kind == 0), 5] ^= 0x40
        S[bad & (kind == 1), 40] ^= 0x02
        m2 = bad & (kind == 2)
        M[m2, 3] ^= 0x80
        for j in np.nonzero(m2)[0]:
            D[j] = np.frombuffer(hashlib.sha256(M[j].tobytes()).digest(), dtype=np.uint8)
        P[bad & (kind == 3), 63] ^= 0x01          # off the curve (or, rarely, another valid key): rejected either way
        res = cv.ecdsa_verify(P.tobytes(), S.tobytes(), D.tobytes(), 32)
        assert res == bytes(bad.astype(np.uint8))
        # against the reference binary
        idx = [int(x) for x in np.sort(rng.choice(n, size=min(n, NREF), replace=False))]
        sp, ss, sm = cut(P.tobytes(), 64, idx), cut(S.tobytes(), 64, idx), cut(M.tobytes(), ml, idx)
        exp = O.join_slices(O.in_slices(lambda lo, hi: r.ecdsa_verify(h, sp[64 * lo:64 * hi], ss[64 * lo:64 * hi], sm[ml * lo:ml * hi], ml), len(idx)))
        assert exp == bytes(res[j] for j in idx) and 0 < sum(exp) < len(idx)
        idx2 = idx[:len(idx) // 2]
        sd, sk, sm2 = cut(d, 32, idx2), cut(ks, 32, idx2), cut(msgs, ml, idx2)
        rs, rp, rst = O.join_slices(O.in_slices(lambda lo, hi: r.ecdsa_sign(h, sd[32 * lo:32 * hi], sk[32 * lo:32 * hi], sm2[ml * lo:ml * hi], ml), len(idx2)))
        assert set(rst) == {0} and rs == cut(sigs, 64, idx2) and rp == cut(pubs, 64, idx2)
    finally:
        cv.free()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
def test_ed25519_full_size_vs_reference_binary(gpu_ctx):
    """configs[4], Ed25519: 2^20 signatures made through the device-side signing steps (keys, R and S on the GPU, the three
    SHA-512 per item on the host), verified on the GPU with 10 % corrupted; 2^12 verdicts against libecc's ec_verify and
    2^10 (key, signature) pairs byte for byte against libecc's key derivation + ec_sign from the same seeds"""
    rng = np.random.default_rng(402)
    ml = 16
    cv = gpu_ctx.curve("WEI25519")
    try:
        n = 1 << LOG2N
        seeds = rand_bytes(rng, 32 * n)
        msgs = rand_bytes(rng, ml * n)
        hk = [hashlib.sha512(seeds[32 * i:32 * i + 32]).digest() for i in range(n)]
        a = bytearray()
        for x in hk:
            v = (int.from_bytes(x[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
            a += v.to_bytes(32, "little")
        a = bytes(a)
        # public keys: [a]B encoded -- the R step applied to the secret scalar (B has order q)
        A, st = cv.eddsa_sign_R(b"".join(a[32 * i:32 * i + 32] + bytes(32) for i in range(n)))
        assert set(st) == {0}
        r_hash = b"".join(hashlib.sha512(hk[i][32:] + msgs[ml * i:ml * (i + 1)]).digest() for i in range(n))
        R, st = cv.eddsa_sign_R(r_hash)
        assert set(st) == {0}
        hram = b"".join(hashlib.sha512(R[32 * i:32 * i + 32] + A[32 * i:32 * i + 32] + msgs[ml * i:ml * (i + 1)]).digest() for i in range(n))
        Sb = cv.eddsa_sign_S(r_hash, hram, a)
        sigs = np.concatenate([np.frombuffer(R, dtype=np.uint8).reshape(n, 32), np.frombuffer(Sb, dtype=np.uint8).reshape(n, 32)], axis=1)
        assert cv.eddsa_verify(A, sigs.tobytes(), hram) == bytes(n)
        # ec_verify_batch's one bit for the whole batch: at this size the library decides it with the multi-scalar
        # multiplication of the reference's batch equation (tests/test_gpu_msm.py pins that kernel on small batches)
        assert cv.eddsa_verify_all(A, sigs.tobytes(), hram) == (True, n)
        spoiled = bytearray(hram)
        spoiled[64 * (n - 7) + 9] ^= 2
        assert cv.eddsa_verify_all(A, sigs.tobytes(), bytes(spoiled)) == (False, n - 7)
        i = np.arange(n)
        bad = (i % 10) == 3
        kind = (i // 10) % 3
        S = sigs.copy()
        S[bad & (kind == 0), 7] ^= 0x10           # R
        S[bad & (kind == 1), 33] ^= 0x01          # S
        H = np.frombuffer(hram, dtype=np.uint8).copy().reshape(n, 64)
        M = np.frombuffer(msgs, dtype=np.uint8).copy().reshape(n, ml)
        m2 = bad & (kind == 2)
        M[m2, 0] ^= 1                               # another message
        # the hash binds R: recompute it where R or the message changed
        for j in np.nonzero(bad)[0]:
            H[j] = np.frombuffer(hashlib.sha512(S[j, :32].tobytes() + A[32 * j:32 * j + 32] + M[j].tobytes()).digest(), dtype=np.uint8)
        res = cv.eddsa_verify(A, S.tobytes(), H.tobytes())
        assert res == bytes(bad.astype(np.uint8))
        idx = [int(x) for x in np.sort(rng.choice(n, size=min(n, NREF), replace=False))]
        sa, ss, sm = cut(A, 32, idx), cut(S.tobytes(), 64, idx), cut(M.tobytes(), ml, idx)
        exp = O.join_slices(O.in_slices(lambda lo, hi: O.ref_ed25519_verify(sa[32 * lo:32 * hi], ss[64 * lo:64 * hi], sm[ml * lo:ml * hi], ml), len(idx)))
        assert exp == bytes(res[j] for j in idx) and 0 < sum(exp) < len(idx)
        idx2 = idx[:len(idx) // 4]
        se, sm2 = cut(seeds, 32, idx2), cut(msgs, ml, idx2)
        rp, rs, rst = O.join_slices(O.in_slices(lambda lo, hi: O.ref_ed25519_sign(se[32 * lo:32 * hi], sm2[ml * lo:ml * hi], ml), len(idx2)))
        assert set(rst) == {0} and rp == cut(A, 32, idx2) and rs == cut(sigs.tobytes(), 64, idx2)
    finally:
        cv.free()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
def test_x25519_full_size_vs_reference_binary(gpu_ctx):
    """configs[4], X25519: 2^20 key agreements -- X25519(a, X25519(b, 9)) == X25519(b, X25519(a, 9)) for every pair -- and
    2^20 random (k, u) inputs (about half of them on the twist, which libecc rejects), 2^12 of each against libecc's x25519()"""
    rng = np.random.default_rng(403)
    cv = gpu_ctx.curve("WEI25519")
    try:
        n = 1 << LOG2N
        ka, kb = rand_bytes(rng, 32 * n), rand_bytes(rng, 32 * n)
        nine = (9).to_bytes(32, "little") * n
        pa, sa = cv.xdh(ka, nine)
        pb, sb = cv.xdh(kb, nine)
        assert set(sa) == {0} and set(sb) == {0}
        s1, st1 = cv.xdh(ka, pb)
        s2, st2 = cv.xdh(kb, pa)
        assert set(st1) == {0} and (s1, st1) == (s2, st2)
        u = np.frombuffer(rand_bytes(rng, 32 * n), dtype=np.uint8).copy().reshape(n, 32)
        u[:, 31] &= 0x7f
        u[::97] = 0xff                              # non-canonical: >= p (top bit set as well)
        u[5::97] = 0                                # u = 0: small order
        out, st = cv.xdh(ka, u.tobytes())
        assert 0.3 * n < st.count(1) < 0.7 * n
        idx = [int(x) for x in np.sort(rng.choice(n, size=min(n, NREF), replace=False))]
        sk, su, sp = cut(ka, 32, idx), cut(u.tobytes(), 32, idx), cut(pb, 32, idx)
        assert O.join_slices(O.in_slices(lambda lo, hi: O.ref_xdh(32, sk[32 * lo:32 * hi], su[32 * lo:32 * hi]), len(idx))) == \
            (cut(out, 32, idx), bytes(st[j] for j in idx))
        assert O.join_slices(O.in_slices(lambda lo, hi: O.ref_xdh(32, sk[32 * lo:32 * hi], sp[32 * lo:32 * hi]), len(idx))) == \
            (cut(s1, 32, idx), bytes(len(idx)))
    finally:
        cv.free()


# ---- row f4 at the reference's own batch function ("What's weak" 1 / item 2) ----
def _pieces(n, size, count, rng, must_hold=None):
    """`count` random contiguous pieces of `size` items of range(n) (aligned to `size`), always including the one that holds must_hold"""
    slots = max(1, n // size)
    pick = set(int(x) for x in rng.choice(slots, size=min(count, slots), replace=False))
    if must_hold is not None:
        pick.add(min(slots - 1, must_hold // size))
    return [(size * x, min(n, size * x + size)) for x in sorted(pick)]


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
@pytest.mark.parametrize("log2n", [17, 20])
def test_bip0340_whole_batch_vs_the_reference_batch_verifier(gpu_ctx, log2n):
    """ec_schnorr_verify_all_batch (the form libsign_amd.so's ec_verify_batch switches to from 2^17 items per device) against
    libecc's OWN batch function on the same batch: ec_verify_batch -> bip0340_verify_batch (sig/sig_algs.c:675, sig/bip0340.c:1296).
    2^n distinct BIP0340 signatures are made from Python integers (oracles.make_bip0340_batch); the reference must accept them and
    reject the batch once one item, at a random index, is damaged -- and the GPU form must say the same both times.  libecc's verifier
    is one thread at 4 ms per item (and its Bos-Coster variant is quadratic in the batch), so the reference runs on contiguous PIECES
    on every host thread: the whole batch at 2^17 (pieces of 2^12), random pieces totalling 2^15 items plus the piece holding the
    damaged item at 2^20."""
    curve = "SECP256K1"
    n = 1 << min(log2n, LOG2N)
    rng = np.random.default_rng(1700 + log2n)
    cv = gpu_ctx.curve(curve)
    try:
        it = O.make_bip0340_batch(lambda sc: cv.scalar_mult(sc), curve, n, rng)
        cl, ql, sl = it["cl"], it["ql"], it["sig_len"]
        assert cv.schnorr_verify_all(it["s"], it["ne"], it["keys"], it["rx"], 1)
        k = int(rng.integers(0, n))
        psize = 1 << 12 if log2n <= 17 else 1 << 9
        pieces = _pieces(n, psize, n // psize if log2n <= 17 else (1 << 15) // psize, rng, must_hold=k)

        def ref(sigs, pcs):
            def one(lo, hi):
                return [O.ref_sig_verify_all(curve, "BIP0340", "SHA256", it["pubs"][2 * cl * a:2 * cl * b], sigs[sl * a:sl * b], sl,
                                             it["msgs"][32 * a:32 * b], 32) for a, b in pcs[lo:hi]]
            return [v for part in O.in_slices(one, len(pcs)) for v in part]
        assert all(ref(it["sigs"], pieces)), "the unmodified reference rejects a piece of the batch the GPU accepted"
        # one damaged item: s_k + 1 in the signature the reference sees and in the scalar the multi-scalar form sees
        s_k = (int.from_bytes(it["s"][ql * k:ql * (k + 1)], "big") + 1) % it["q"]
        bad_s = it["s"][:ql * k] + s_k.to_bytes(ql, "big") + it["s"][ql * (k + 1):]
        bad_sigs = it["sigs"][:sl * k + cl] + s_k.to_bytes(ql, "big") + it["sigs"][sl * (k + 1):]
        assert not cv.schnorr_verify_all(bad_s, it["ne"], it["keys"], it["rx"], 1)
        # (every other piece is byte-identical to the run above: the piece that holds item k and a few neighbours are run again)
        again = [p for p in pieces if p[0] <= k < p[1]] + [p for p in pieces if not (p[0] <= k < p[1])][:O.host_threads() - 1]
        verdicts = ref(bad_sigs, again)
        assert not verdicts[0] and all(verdicts[1:])
    finally:
        cv.free()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
@pytest.mark.parametrize("log2n", [17, 20])
def test_ed25519_whole_batch_vs_the_reference_batch_verifier(gpu_ctx, log2n):
    """the same for ec_eddsa_verify_all_batch against ec_verify_batch -> eddsa_verify_batch (sig/eddsa.c:2904) of the unmodified
    reference: 2^n distinct signatures (signed on the GPU by ec_eddsa_sign_R/S_batch, hashes by hashlib), accepted by both; one damaged
    item at a random index: rejected by both (the reference on the piece that holds it, every other sampled piece still accepted)."""
    n = 1 << min(log2n, LOG2N)
    rng = np.random.default_rng(2500 + log2n)
    cv = gpu_ctx.curve("WEI25519")
    try:
        seeds, msgs = rand_bytes(rng, 32 * n), rand_bytes(rng, 32 * n)
        hk = [hashlib.sha512(seeds[32 * i:32 * i + 32]).digest() for i in range(n)]
        a_np = np.frombuffer(b"".join(h[:32] for h in hk), dtype=np.uint8).reshape(n, 32).copy()
        a_np[:, 0] &= 248
        a_np[:, 31] &= 127
        a_np[:, 31] |= 64
        wide = np.zeros((n, 64), dtype=np.uint8)
        wide[:, :32] = a_np
        pubs, st = cv.eddsa_sign_R(wide.tobytes())
        assert set(st) == {0}
        r_hash = b"".join(hashlib.sha512(hk[i][32:] + msgs[32 * i:32 * i + 32]).digest() for i in range(n))
        Renc, st = cv.eddsa_sign_R(r_hash)
        assert set(st) == {0}
        hram = b"".join(hashlib.sha512(Renc[32 * i:32 * i + 32] + pubs[32 * i:32 * i + 32] + msgs[32 * i:32 * i + 32]).digest() for i in range(n))
        Sb = cv.eddsa_sign_S(r_hash, hram, a_np.tobytes())
        sg = np.empty((n, 64), dtype=np.uint8)
        sg[:, :32] = np.frombuffer(Renc, dtype=np.uint8).reshape(n, 32)
        sg[:, 32:] = np.frombuffer(Sb, dtype=np.uint8).reshape(n, 32)
        sigs = sg.tobytes()
        gpu_ctx.set_eddsa_msm(2, 0, 0)                  # the multi-scalar form whatever the size
        ok, first = cv.eddsa_verify_all(pubs, sigs, hram)
        assert ok and first == n

        def combination_alone(sg_bytes):
            """the verdict byte of the multi-scalar multiplication ITSELF (the host-pointer form above follows a rejected combination with the
            item-by-item pass and would hide a combination that wrongly rejects): 0 = the batch equation holds"""
            import torch
            dev = torch.device("cuda:0")
            t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
            dp, ds, dh = t(pubs), t(sg_bytes), t(hram)
            verdict = torch.full((1,), 7, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            cv.eddsa_verify_all_dev(n, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), verdict.data_ptr(), None)
            torch.cuda.synchronize()
            return int(verdict.item())
        assert combination_alone(sigs) == 0
        k = int(rng.integers(0, n))
        psize = 1 << 12 if log2n <= 17 else 1 << 9
        pieces = _pieces(n, psize, n // psize if log2n <= 17 else (1 << 15) // psize, rng, must_hold=k)

        def ref(sg_bytes, pcs):
            def one(lo, hi):
                return [O.ref_eddsa_verify_all(pubs[32 * a:32 * b], sg_bytes[64 * a:64 * b], msgs[32 * a:32 * b], 32) for a, b in pcs[lo:hi]]
            return [v for part in O.in_slices(one, len(pcs)) for v in part]
        assert all(ref(sigs, pieces))
        bad = bytearray(sigs)
        bad[64 * k + 33] ^= 0x04                      # S_k damaged (the hash binds R, A and M, not S)
        bad = bytes(bad)
        ok, first = cv.eddsa_verify_all(pubs, bad, hram)
        assert not ok and first == k
        assert combination_alone(bad) == 1
        again = [p for p in pieces if p[0] <= k < p[1]] + [p for p in pieces if not (p[0] <= k < p[1])][:O.host_threads() - 1]
        verdicts = ref(bad, again)
        assert not verdicts[0] and all(verdicts[1:])
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()
