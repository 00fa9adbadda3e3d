"""GPU tests of the verification entry points that take MESSAGES and hash on the device (ec_ecdsa_verify_msg_batch_fmt,
ec_eddsa_verify_msg_batch; libecc_amd/csrc/ecamd_hash.hip): the same accept / reject bytes as the digest-taking entry points fed with
hashlib's digests, on messages of every length class (empty, one block, the padding boundaries, two blocks), valid and corrupted."""
import hashlib

import numpy as np
import pytest

import oracles as O
from oracles import Oracle, clen, qlen
from test_gpu_parity import rand_bytes

pytestmark = pytest.mark.gpu

CASES = [("SECP256R1", 2, hashlib.sha256), ("SECP384R1", 3, hashlib.sha384), ("SECP521R1", 4, hashlib.sha512), ("SECP224R1", 1, hashlib.sha224),
         ("SECP256R1", 4, hashlib.sha512), ("BRAINPOOLP256R1", 2, hashlib.sha256)]


@pytest.mark.parametrize("curve,hash_type,hf", CASES)
def test_ecdsa_verify_from_messages(gpu_ctx, curve, hash_type, hf):
    rng = np.random.default_rng(90 + hash_type)
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    cl, ql = clen(curve), qlen(curve)
    q = O.CURVES[curve]["q"]
    try:
        lens = [0, 1, 31, 32, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 200] * 4
        n = len(lens)
        msgs = [rand_bytes(rng, k) for k in lens]
        dg = b"".join(hf(m).digest() for m in msgs)
        hl = hf().digest_size
        privs = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
        nonces = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
        pubs, st = cv.scalar_mult(privs)
        assert set(st) == {0}
        sigs, st = cv.ecdsa_sign(privs, nonces, dg, hl)
        assert set(st) == {0}
        sigs = bytearray(sigs)
        for i in range(0, n, 5):
            sigs[2 * ql * i + ql + 3] ^= 1
        sigs = bytes(sigs)
        exp = cv.ecdsa_verify(pubs, sigs, dg, hl)
        assert exp == o.ecdsa_verify(pubs, sigs, dg, hl) and 0 in exp and 1 in exp
        assert cv.ecdsa_verify_msgs(pubs, 0, sigs, hash_type, msgs) == exp
        # projective keys (what an ec_pub_key holds), and a flipped message byte
        prj = b"".join(pubs[2 * cl * i:2 * cl * (i + 1)] + (1).to_bytes(cl, "big") for i in range(n))
        assert cv.ecdsa_verify_msgs(prj, 1, sigs, hash_type, msgs) == exp
        bad = [bytes([m[0] ^ 1]) + m[1:] if m else b"x" for m in msgs]
        assert cv.ecdsa_verify_msgs(pubs, 0, sigs, hash_type, bad) == bytes([1]) * n
        # a larger tiled batch with a wider stride
        reps = 40
        slots, stride = cv.msg_slots(msgs * reps, 256)
        import ctypes as C
        res = C.create_string_buffer(n * reps)
        assert cv.L.ec_ecdsa_verify_msg_batch_fmt(cv.ctx.h, cv.h, n * reps, pubs * reps, 0, sigs * reps, hash_type, slots, stride, res) == 0
        assert res.raw == exp * reps
        assert cv.ecdsa_verify_msgs(b"", 0, b"", hash_type, []) == b""
    finally:
        cv.free()


def test_eddsa_verify_from_hash_inputs(gpu_ctx):
    from test_oracle import ed25519_cases, ED_MSG_LEN
    rng = np.random.default_rng(95)
    cv = gpu_ctx.curve("WEI25519")
    try:
        pubs, sigs, msgs, hram = ed25519_cases(rng, nvalid=30)
        n = len(pubs) // 32
        exp = cv.eddsa_verify(pubs, sigs, hram)
        assert 0 in exp and 1 in exp
        inputs = [sigs[64 * i:64 * i + 32] + pubs[32 * i:32 * i + 32] + msgs[ED_MSG_LEN * i:ED_MSG_LEN * (i + 1)] for i in range(n)]
        assert hashlib.sha512(inputs[0]).digest() == hram[:64]
        assert cv.eddsa_verify_msgs(pubs, sigs, inputs) == exp
        reps = 4200 // n + 1
        assert cv.eddsa_verify_msgs(pubs * reps, sigs * reps, inputs * reps) == exp * reps
    finally:
        cv.free()


def test_eddsa_verify_from_projective_keys_and_messages(gpu_ctx):
    """ec_eddsa_verify_msg_prj_batch: the projective Weierstrass key an ec_pub_key holds goes in, the device imports it, encodes it
    (eddsa_export_pub_key) and writes the 32 octets into the blank of the hash input R || A || M before hashing.  Expected verdicts:
    ec_eddsa_verify_batch on the encodings ec_eddsa_encode_point_batch gives for the same keys and hashlib's hashes -- both pinned
    against the reference elsewhere -- with signatures made here from Python integers; keys that do not import, keys at infinity,
    corrupted signatures and messages, ragged message lengths, a multi-chunk batch."""
    rng = np.random.default_rng(97)
    cv = gpu_ctx.curve("WEI25519")
    c = O.CURVES["WEI25519"]
    p, q = c["p"], c["q"]
    try:
        n = 96
        lens = [0, 1, 31, 47, 48, 63, 64, 65, 111, 150] * 10
        msgs = [rand_bytes(rng, lens[i]) for i in range(n)]
        a = [(int.from_bytes(rand_bytes(rng, 40), "big") % (q - 1)) + 1 for _ in range(n)]
        r = [(int.from_bytes(rand_bytes(rng, 40), "big") % (q - 1)) + 1 for _ in range(n)]
        Aw, st = cv.scalar_mult(b"".join(x.to_bytes(32, "big") for x in a))
        Rw, st2 = cv.scalar_mult(b"".join(x.to_bytes(32, "big") for x in r))
        assert set(st) == {0} and set(st2) == {0}

        def prj(aff, i, lam):
            x, y = int.from_bytes(aff[64 * i:64 * i + 32], "big"), int.from_bytes(aff[64 * i + 32:64 * i + 64], "big")
            return b"".join((v % p).to_bytes(32, "big") for v in (x * lam, y * lam, lam))
        lams = [1 if i % 3 == 0 else int.from_bytes(rand_bytes(rng, 40), "big") % (p - 1) + 1 for i in range(n)]
        keys = b"".join(prj(Aw, i, lams[i]) for i in range(n))
        Aenc, est = cv.eddsa_encode_points(keys)
        Renc, est2 = cv.eddsa_encode_points(b"".join(prj(Rw, i, 1) for i in range(n)))
        assert set(est) == {0} and set(est2) == {0}
        sigs, hram, inputs = bytearray(), b"", []
        for i in range(n):
            Ri, Ai = Renc[32 * i:32 * i + 32], Aenc[32 * i:32 * i + 32]
            hd = hashlib.sha512(Ri + Ai + msgs[i]).digest()
            S = (r[i] + (int.from_bytes(hd, "little") % q) * a[i]) % q
            sigs += Ri + S.to_bytes(32, "little")
            hram += hd
            inputs.append(Ri + bytes(32) + msgs[i])          # the blank the device fills
        for i in range(0, n, 7):
            sigs[64 * i + 40] ^= 4                           # corrupted S
        for i in range(3, n, 11):
            sigs[64 * i + 5] ^= 1                            # corrupted R (the hash input keeps the original: also a rejection)
        sigs = bytes(sigs)
        exp = cv.eddsa_verify(Aenc, sigs, hram)
        assert 0 in exp and 1 in exp
        assert cv.eddsa_verify_msgs_prj(keys, sigs, inputs, 32) == exp
        # keys that do not import / are at infinity / are not on the curve reject their item and nothing else
        kb = bytearray(keys)
        kb[96 * 1:96 * 1 + 32] = p.to_bytes(32, "big")                        # X = p
        kb[96 * 2 + 64:96 * 3] = bytes(32)                                   # Z = 0 (with X, Y left): not on the curve
        kb[96 * 4:96 * 5] = bytes(32) + (1).to_bytes(32, "big") + bytes(32)  # (0 : 1 : 0): the point at infinity
        kb[96 * 5 + 31] ^= 1                                                 # off the curve
        got = cv.eddsa_verify_msgs_prj(bytes(kb), sigs, inputs, 32)
        for i in range(n):
            assert got[i] == (1 if i in (1, 2, 4, 5) else exp[i]), i
        # a flipped message byte
        bad = [x[:64] + (bytes([x[64] ^ 1]) + x[65:] if len(x) > 64 else b"y") for x in inputs]
        assert cv.eddsa_verify_msgs_prj(keys, sigs, bad, 32) == bytes([1]) * n
        # several chunks of the host pipeline (short first chunk included)
        reps = 4200 // n + 1
        assert cv.eddsa_verify_msgs_prj(keys * reps, sigs * reps, inputs * reps, 32) == exp * reps
        # the context and pre-hashed variants: dom2(phflag, ctx) in front of R; for Ed25519ph the device also computes PH(M) = SHA-512(M)
        # into the second blank (ec_eddsa_verify_ph_prj_batch)
        ctxb = b"the context of this batch"
        for ph in (0, 1):
            dom = b"SigEd25519 no Ed25519 collisions" + bytes([ph, len(ctxb)]) + ctxb
            sg2, hr2, in2 = bytearray(), b"", []
            for i in range(n):
                Ri, Ai = Renc[32 * i:32 * i + 32], Aenc[32 * i:32 * i + 32]
                body = hashlib.sha512(msgs[i]).digest() if ph else msgs[i]
                hd = hashlib.sha512(dom + Ri + Ai + body).digest()
                S = (r[i] + (int.from_bytes(hd, "little") % q) * a[i]) % q
                sg2 += Ri + S.to_bytes(32, "little")
                hr2 += hd
                in2.append(dom + Ri + bytes(96 if ph else 32) + (b"" if ph else msgs[i]))
            for i in range(0, n, 6):
                sg2[64 * i + 33] ^= 8
            sg2 = bytes(sg2)
            exp2 = cv.eddsa_verify(Aenc, sg2, hr2)
            assert 0 in exp2 and 1 in exp2
            off = len(dom) + 32
            if ph:
                assert cv.eddsa_verify_ph_prj(keys, sg2, in2, off, msgs) == exp2
                assert cv.eddsa_verify_ph_prj(keys * reps, sg2 * reps, in2 * reps, off, msgs * reps) == exp2 * reps
                assert cv.eddsa_verify_ph_prj(keys, sg2, in2, off, [m + b"!" for m in msgs]) == bytes([1]) * n
            else:
                assert cv.eddsa_verify_msgs_prj(keys, sg2, in2, off) == exp2
    finally:
        cv.free()
