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
