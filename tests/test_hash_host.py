"""CPU test of libecc_amd/csrc/ecamd_hash.hip (SHA-224 / 256 / 384 / 512 of a batch of short messages in fixed-stride slots, the
hashing step of ec_ecdsa_verify_msg_batch_fmt / ec_eddsa_verify_msg_batch): the kernel source compiled for the host against a
stand-in for <hip/hip_runtime.h> (tests/hipstub) and run lane by lane, compared with hashlib on every length around the block and
padding boundaries and on random content."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
ALGS = {1: hashlib.sha224, 2: hashlib.sha256, 3: hashlib.sha384, 4: hashlib.sha512}


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "hash_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", os.path.join(ROOT, "tests", "hipstub"),
                           "-o", so, os.path.join(ROOT, "tests", "hash_host_shim.cpp")])
    return C.CDLL(so)


def slots_of(msgs, stride):
    buf = bytearray(stride * len(msgs))
    for i, m in enumerate(msgs):
        assert 4 + len(m) <= stride
        buf[stride * i:stride * i + 4] = len(m).to_bytes(4, "little")
        buf[stride * i + 4:stride * i + 4 + len(m)] = m
    return bytes(buf)


@pytest.mark.parametrize("hash_type", [1, 2, 3, 4])
def test_sha2_slots_match_hashlib(lib, hash_type):
    rng = np.random.default_rng(80 + hash_type)
    stride = 256
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in range(0, stride - 3)]
    msgs += [b"", b"abc", b"\x80" * 55, b"\xff" * 56, b"\x00" * 64, b"a" * 111, b"a" * 112, b"a" * 119, b"a" * 120, b"a" * 127, b"a" * 128]
    dl = ALGS[hash_type]().digest_size
    for st in (stride, 48 if max(len(m) for m in msgs[:45]) <= 44 else stride):
        ms = msgs if st == stride else msgs[:45]
        out = C.create_string_buffer(dl * len(ms))
        assert lib.sha2_slots_host(hash_type, slots_of(ms, st), st, len(ms), out, dl) == 0
        for i, m in enumerate(ms):
            assert out.raw[dl * i:dl * (i + 1)] == ALGS[hash_type](m).digest(), (hash_type, len(m))


def test_shake256_slots_match_hashlib(lib):
    """SHAKE256 (the Ed448 hash: 114 octets; the Ed448ph pre-hash: 64) of every length around the 136-octet rate boundaries"""
    rng = np.random.default_rng(85)
    stride = 300
    msgs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8)) for n in range(0, stride - 3)]
    msgs += [b"", b"abc", b"\x1f" * 135, b"\x80" * 136, b"\xff" * 137, b"a" * 271, b"a" * 272, b"a" * 273]
    for outlen in (114, 64, 136, 1):
        out = C.create_string_buffer(outlen * len(msgs))
        assert lib.shake256_slots_host(slots_of(msgs, stride), stride, len(msgs), out, outlen, outlen) == 0
        for i, m in enumerate(msgs):
            assert out.raw[outlen * i:outlen * (i + 1)] == hashlib.shake_256(m).digest(outlen), (outlen, len(m))


@pytest.mark.parametrize("cl,ql,rl_is_x", [(32, 32, True), (32, 32, False), (48, 48, False), (28, 29, True)])
def test_schnorr_prep_moves_the_bytes(lib, cl, ql, rl_is_x):
    """k_schnorr_prep (round 6), lane by lane on the CPU: the key's x into the blank of the hash input, the key's even-y representative
    (y <- p - y when y is odd, BIP0340's lift_x), s and the commitment split off the signature, the flag for a key that did not import"""
    rng = np.random.default_rng(90 + cl)
    n, rl = 37, (cl if rl_is_x else 2 * cl)
    p = (1 << (8 * cl)) - 189 if cl != 28 else (1 << 224) - 6803
    p |= 1
    keys = bytearray()
    for i in range(n):
        x = int.from_bytes(rng.integers(0, 256, size=cl, dtype=np.uint8).tobytes(), "big") % p
        y = int.from_bytes(rng.integers(0, 256, size=cl, dtype=np.uint8).tobytes(), "big") % p
        keys += x.to_bytes(cl, "big") + y.to_bytes(cl, "big")
    sigs = rng.integers(0, 256, size=n * (rl + ql), dtype=np.uint8).tobytes()
    stride, xo = 160, 70
    slots0 = rng.integers(0, 256, size=n * stride, dtype=np.uint8).tobytes()
    for even_y, use_xoff, kst in ((1, True, None), (0, False, bytes(n)), (1, True, bytes([0] * 5 + [2] + [0] * (n - 6)))):
        slots = C.create_string_buffer(slots0, n * stride)
        ko, so, ro = C.create_string_buffer(2 * cl * n), C.create_string_buffer(ql * n), C.create_string_buffer(rl * n)
        flag = C.c_uint32(0)
        assert lib.schnorr_prep_host(bytes(keys), kst, sigs, slots, ko, so, ro, C.byref(flag), n, cl, ql, rl, stride,
                                     xo if use_xoff else 0xffffffff, even_y, p.to_bytes(cl, "big")) == 0
        assert flag.value == (1 if kst and any(kst) else 0)
        for i in range(n):
            kx, ky = keys[2 * cl * i:2 * cl * i + cl], keys[2 * cl * i + cl:2 * cl * (i + 1)]
            want_y = (p - int.from_bytes(ky, "big")).to_bytes(cl, "big") if (even_y and ky[-1] & 1) else bytes(ky)
            assert ko.raw[2 * cl * i:2 * cl * (i + 1)] == bytes(kx) + want_y
            assert ro.raw[rl * i:rl * (i + 1)] == sigs[(rl + ql) * i:(rl + ql) * i + rl]
            assert so.raw[ql * i:ql * (i + 1)] == sigs[(rl + ql) * i + rl:(rl + ql) * (i + 1)]
            want = bytearray(slots0[stride * i:stride * (i + 1)])
            if use_xoff:
                want[4 + xo:4 + xo + cl] = kx
            assert slots.raw[stride * i:stride * (i + 1)] == bytes(want)
