"""RFC 8032 section 7.1 (plain Ed25519) and RFC 7748 section 5.2 (iterated X25519 / X448, 1 000 iterations) -- the public
vectors the reference snapshot does not carry for this path (its tests/ed25519_test_vectors.h is absent).
tests/golden/rfc_vectors.json is written by tests/golden/make_rfc_vectors.py, which cross-checks every typed-in value with
two independent implementations.  CPU legs: the oracle; GPU legs: the C ABI."""
import hashlib
import json
import os

import pytest

import oracles as O
from oracles import Oracle

V = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rfc_vectors.json")))
ED_Q = 2**252 + 27742317777372353535851937790883648493


def ed_inputs():
    """(public keys, signatures, hram, r_hash, secret scalars) of the RFC 8032 vectors, as the entry points take them"""
    pubs = sigs = hram = rh = a = b""
    for v in V["ed25519"]:
        sk, pk, m, sg = (bytes.fromhex(v[k]) for k in ("secret_key", "public_key", "message", "signature"))
        hk = hashlib.sha512(sk).digest()
        s = (int.from_bytes(hk[:32], "little") & ((1 << 254) - 8)) | (1 << 254)
        pubs += pk
        sigs += sg
        hram += hashlib.sha512(sg[:32] + pk + m).digest()
        rh += hashlib.sha512(hk[32:] + m).digest()
        a += s.to_bytes(32, "little")
    return pubs, sigs, hram, rh, a


def test_ed25519_rfc8032_vectors_oracle():
    pubs, sigs, hram, rh, a = ed_inputs()
    n = len(V["ed25519"])
    o = Oracle("WEI25519")
    assert o.eddsa_verify(pubs, sigs, hram) == bytes(n)
    R, st = o.eddsa_sign_R(rh)
    assert set(st) == {0} and R == b"".join(sigs[64 * i:64 * i + 32] for i in range(n))
    assert o.eddsa_sign_S(rh, hram, a) == b"".join(sigs[64 * i + 32:64 * i + 64] for i in range(n))
    # public keys: [a]B encoded = the R step applied to the secret scalar
    A, st = o.eddsa_sign_R(b"".join(a[32 * i:32 * i + 32] + bytes(32) for i in range(n)))
    assert set(st) == {0} and A == pubs
    # one flipped bit anywhere rejects
    bad = bytearray(sigs)
    bad[3] ^= 1
    bad[64 + 40] ^= 0x80
    assert o.eddsa_verify(pubs, bytes(bad), hram)[:2] == b"\1\1"


@pytest.mark.parametrize("kind", ["x25519", "x448"])
def test_xdh_rfc7748_iterated_oracle(kind):
    """k, u = X(k, u), k -- 1 000 times (every u after the first is an output, hence on the curve)"""
    v = V["xdh_iterated"][kind]
    o = Oracle("WEI25519" if kind == "x25519" else "WEI448")
    k = u = bytes.fromhex(v["start"])
    for it in range(1, 1001):
        out, st = o.xdh(k, u)
        assert st == b"\0", it
        k, u = out, k
        if it == 1:
            assert k.hex() == v["after_1"]
    assert k.hex() == v["after_1000"]


@pytest.mark.gpu
def test_ed25519_rfc8032_vectors_gpu(gpu_ctx):
    pubs, sigs, hram, rh, a = ed_inputs()
    n = len(V["ed25519"])
    cv = gpu_ctx.curve("WEI25519")
    try:
        assert cv.eddsa_verify(pubs, sigs, hram) == bytes(n)
        assert cv.eddsa_verify_all(pubs, sigs, hram)[0] is True
        R, st = cv.eddsa_sign_R(rh)
        assert set(st) == {0} and R == b"".join(sigs[64 * i:64 * i + 32] for i in range(n))
        assert cv.eddsa_sign_S(rh, hram, a) == b"".join(sigs[64 * i + 32:64 * i + 64] for i in range(n))
        A, st = cv.eddsa_sign_R(b"".join(a[32 * i:32 * i + 32] + bytes(32) for i in range(n)))
        assert set(st) == {0} and A == pubs
        bad = bytearray(sigs)
        bad[3] ^= 1
        bad[64 + 40] ^= 0x80
        assert cv.eddsa_verify(pubs, bytes(bad), hram) == b"\1\1" + bytes(n - 2)
        # the same vectors tiled to a batch that takes the multi-scalar multiplication path of ec_eddsa_verify_all_batch
        reps = (1 << 17) // n + 1
        ok, first = cv.eddsa_verify_all(pubs * reps, sigs * reps, hram * reps)
        assert ok is True and first == n * reps
    finally:
        cv.free()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["x25519", "x448"])
def test_xdh_rfc7748_iterated_gpu(gpu_ctx, kind):
    """the 1 000-iteration chain of RFC 7748 section 5.2, as 64 identical chains in one batch per iteration (one wavefront)"""
    v = V["xdh_iterated"][kind]
    cv = gpu_ctx.curve("WEI25519" if kind == "x25519" else "WEI448")
    try:
        k = u = bytes.fromhex(v["start"])
        ln = len(k)
        for it in range(1, 1001):
            out, st = cv.xdh(k * 64, u * 64)
            assert set(st) == {0} and out == out[:ln] * 64, it
            k, u = out[:ln], k
            if it == 1:
                assert k.hex() == v["after_1"]
        assert k.hex() == v["after_1000"]
    finally:
        cv.free()
