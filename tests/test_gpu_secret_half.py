"""GPU tests of the round-4 entry points of the secret-key half that take RAW random material (ec_nn_random_mod_batch,
ec_ecdsa_sign_msg_batch, ec_key_pair_gen_raw_batch): the nonce / private scalar is nn_get_random_mod's value for the item's 2 * qlen
random bytes -- LE(raw) mod (q - 1) + 1, pinned on the unmodified reference by tests/test_oracle.py::test_random_mod_vs_reference --
computed on the device, and H(m) comes from the device too.  Expected bytes: the digest- and nonce-taking entry points (themselves
pinned against the reference) fed with Python's reductions and hashlib's digests."""
import hashlib

import numpy as np
import pytest

import oracles as O
from oracles import Oracle, clen, qlen
from test_gpu_parity import rand_bytes
from test_randmod_host import randmod_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "SECP521R1", "SECP224R1", "WEI25519", "SECP192R1"])
def test_random_mod_on_the_device(gpu_ctx, curve):
    q = O.CURVES[curve]["q"]
    cv = gpu_ctx.curve(curve)
    try:
        ql, vals = randmod_cases(q, np.random.default_rng(11), nrand=2000)
        raw = b"".join(v.to_bytes(2 * ql, "little") for v in vals)
        exp = b"".join((v % (q - 1) + 1).to_bytes(ql, "big") for v in vals)
        assert cv.random_mod(raw) == exp == Oracle(curve).random_mod(raw)
        assert cv.random_mod(b"") == b""
    finally:
        cv.free()


@pytest.mark.parametrize("curve,hash_type,hf", [("SECP256R1", 2, hashlib.sha256), ("SECP384R1", 3, hashlib.sha384), ("SECP521R1", 4, hashlib.sha512)])
def test_ecdsa_sign_from_messages_and_raw_nonces(gpu_ctx, curve, hash_type, hf):
    rng = np.random.default_rng(300 + hash_type)
    q = O.CURVES[curve]["q"]
    ql, cl = qlen(curve), clen(curve)
    cv = gpu_ctx.curve(curve)
    try:
        lens = [0, 1, 31, 55, 56, 64, 100, 119, 120, 200] * 13
        n = len(lens)
        msgs = [rand_bytes(rng, k) for k in lens]
        privs = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(n))
        raw = bytearray(rand_bytes(rng, 2 * ql * n))
        raw[0:2 * ql] = bytes(2 * ql)                                   # raw = 0: k = 1
        raw[2 * ql:4 * ql] = (q - 2).to_bytes(2 * ql, "little")         # k = q - 1
        raw = bytes(raw)
        nonces = b"".join(((int.from_bytes(raw[2 * ql * i:2 * ql * (i + 1)], "little") % (q - 1)) + 1).to_bytes(ql, "big") for i in range(n))
        dg = b"".join(hf(m).digest() for m in msgs)
        hl = hf().digest_size
        exp = cv.ecdsa_sign(privs, nonces, dg, hl)
        assert set(exp[1]) == {0}
        assert cv.ecdsa_sign_msgs(privs, raw, hash_type, msgs) == exp
        # the same with the digests handed over (hash_type 0), and a private key >= q failing its item only
        assert cv.ecdsa_sign_msgs(privs, raw, 0, [dg[hl * i:hl * (i + 1)] for i in range(n)]) == exp
        bad = q.to_bytes(ql, "big") + privs[ql:]
        sg, st = cv.ecdsa_sign_msgs(bad, raw, hash_type, msgs)
        assert st[0] == 1 and set(st[1:]) == {0} and sg[2 * ql:] == exp[0][2 * ql:]
        # the signatures verify (device hashing on that side too), several chunks of the host pipeline
        pubs, st = cv.scalar_mult(privs)
        assert set(st) == {0}
        reps = 4200 // n + 1
        sg, st = cv.ecdsa_sign_msgs(privs * reps, raw * reps, hash_type, msgs * reps)
        assert sg == exp[0] * reps and set(st) == {0}
        assert cv.ecdsa_verify_msgs(pubs * reps, 0, sg, hash_type, msgs * reps) == bytes(n * reps)
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256R1", "SECP384R1", "BRAINPOOLP256R1", "SECP256K1"])
def test_key_pairs_from_raw_random_bytes(gpu_ctx, curve):
    rng = np.random.default_rng(41)
    q = O.CURVES[curve]["q"]
    ql = qlen(curve)
    cv = gpu_ctx.curve(curve)
    gpu_ctx.set_secret_scalars(True)
    try:
        n = 1500
        raw = bytearray(rand_bytes(rng, 2 * ql * n))
        raw[0:2 * ql] = bytes(2 * ql)
        raw[2 * ql:4 * ql] = (q - 2).to_bytes(2 * ql, "little")
        raw = bytes(raw)
        exp_priv = b"".join(((int.from_bytes(raw[2 * ql * i:2 * ql * (i + 1)], "little") % (q - 1)) + 1).to_bytes(ql, "big") for i in range(n))
        pr, pb, st = cv.key_pair_gen_raw(raw)
        assert pr == exp_priv and set(st) == {0}
        assert (pb, st) == cv.scalar_mult(exp_priv)
        o = Oracle(curve)
        assert o.scalar_mult(exp_priv[:ql * 40])[0] == pb[:2 * clen(curve) * 40]
    finally:
        gpu_ctx.set_secret_scalars(False)
        cv.free()


def test_secret_fixed_base_comb_p256(gpu_ctx):
    """secp256r1, secret-scalar mode, fixed base: the scanned 4-bit comb (k_p256_comb4m) gives the bytes of the public-mode comb and
    of the oracle on edge scalars (0, 1, small, q - 1, q, q + 1, 2^256 - 1, single nibbles, all-8 / all-7 / all-f nibbles), short
    scalars and random ones; ECAMD_NO_SECRET_COMB's path (the scanned window loop) is what test_gpu_multi covers"""
    rng = np.random.default_rng(77)
    q = O.CURVES["SECP256R1"]["q"]
    cv = gpu_ctx.curve("SECP256R1")
    try:
        vals = [0, 1, 2, 7, 8, 9, 15, 16, 17, q - 2, q - 1, q, q + 1, 2**256 - 1, 2**255, 2**255 - 1, int("8" * 64, 16), int("7" * 64, 16),
                int("f" * 64, 16), int("78" * 32, 16), int("87" * 32, 16)]
        vals += [m << (4 * j) for j in (0, 1, 31, 62, 63) for m in (1, 7, 8, 9, 15)]
        vals += [int.from_bytes(rand_bytes(rng, 32), "big") for _ in range(3000)]
        sc = b"".join(v.to_bytes(32, "big") for v in vals)
        pub = cv.scalar_mult(sc)
        gpu_ctx.set_secret_scalars(True)
        try:
            sec = cv.scalar_mult(sc)
            short = cv.scalar_mult(b"".join((v % 2**72).to_bytes(9, "big") for v in vals), slen=9)
        finally:
            gpu_ctx.set_secret_scalars(False)
        assert sec == pub
        assert short == cv.scalar_mult(b"".join((v % 2**72).to_bytes(9, "big") for v in vals), slen=9)
        o = Oracle("SECP256R1")
        assert o.scalar_mult(sc[:32 * 60]) == (sec[0][:64 * 60], sec[1][:60])
        assert 2 in sec[1]          # k = 0 and k = q: the point at infinity
    finally:
        cv.free()


@pytest.mark.parametrize("curve", ["SECP384R1", "SECP521R1", "BRAINPOOLP256R1", "SECP256K1", "WEI25519", "SECP224R1", "WEI448", "BRAINPOOLP512R1"])
def test_secret_fixed_base_comb_generic_units(gpu_ctx, curve):
    """the radix-2^29 units, secret-scalar mode, fixed base: the scanned 4-bit comb (k_comb_g<.., SCAN4>, built on the first such call
    of the handle) against the public-mode results and the oracle, on edge scalars and random ones of full and of short length"""
    rng = np.random.default_rng(78)
    c = O.CURVES[curve]
    q, ql = c["q"], qlen(curve)
    nb = 8 * ql
    cv = gpu_ctx.curve(curve)
    try:
        vals = [0, 1, 2, 7, 8, 9, 15, 16, q - 2, q - 1, q % 2**nb, (q + 1) % 2**nb, 2**nb - 1, 2**(nb - 1), int("8" * (2 * ql), 16), int("7" * (2 * ql), 16),
                int("f" * (2 * ql), 16)]
        vals += [(m << (4 * j)) % 2**nb for j in (0, 1, ql, 2 * ql - 2, 2 * ql - 1) for m in (1, 7, 8, 9, 15)]
        vals += [int.from_bytes(rand_bytes(rng, ql), "big") for _ in range(1200)]
        sc = b"".join(v.to_bytes(ql, "big") for v in vals)
        short = b"".join((v % 2**56).to_bytes(7, "big") for v in vals)
        pub, pub_short = cv.scalar_mult(sc), cv.scalar_mult(short, slen=7)
        gpu_ctx.set_secret_scalars(True)
        try:
            sec, sec_short = cv.scalar_mult(sc), cv.scalar_mult(short, slen=7)
        finally:
            gpu_ctx.set_secret_scalars(False)
        assert sec == pub and sec_short == pub_short
        o = Oracle(curve)
        assert o.scalar_mult(sc[:ql * 24]) == (sec[0][:2 * clen(curve) * 24], sec[1][:24])
    finally:
        cv.free()
