"""CPU-side pins of what the GPU tests of the EdDSA additions of round 2 compare against (tests/test_gpu_msm.py,
tests/test_gpu_formats.py::test_eddsa_encode_point_batch): the python ChaCha20, the python form of the batch equation, and the
python public-key encoding -- each against an RFC vector or the unmodified reference library."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import oracles as O  # noqa: E402
from oracles import Oracle, have_ref  # noqa: E402
from test_gpu_msm import chacha20_block, python_combination  # noqa: E402


def test_chacha20_block_rfc8439():
    """RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00 00 00 09 00 00 00 4a 00 00 00 00"""
    out = chacha20_block(bytes(range(32)), 1, [0x09000000, 0x4a000000, 0x00000000])
    assert out.hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
def test_python_batch_equation_vs_reference_ec_verify_batch():
    """[8]T = neutral for the python combination (any non-zero z_i) exactly when the reference's ec_verify_batch accepts --
    over the case families of test_oracle.ed25519_cases restricted to items both A and R of which decode (the others are
    rejected before the equation, by the reference and by the kernels alike)"""
    from test_oracle import ed25519_cases, eddsa_subset, ED_MSG_LEN
    rng = np.random.default_rng(72)
    pubs, sigs, msgs, hram = ed25519_cases(rng, 6)
    n = len(pubs) // 32
    one = Oracle("WEI25519").eddsa_verify(pubs, sigs, hram)
    decodable = [i for i in range(n) if O.ed_decode(pubs[32 * i:32 * i + 32]) and O.ed_decode(sigs[64 * i:64 * i + 32])]
    good = [i for i in decodable if one[i] == 0]
    bad = [i for i in decodable if one[i]]
    assert len(good) >= 6 and len(bad) >= 3

    def python_accepts(idx):
        P, S, M, H = eddsa_subset(idx, pubs, sigs, msgs, hram, 32, 64, ED_MSG_LEN, 64)
        zs = rng.integers(1, 256, size=16 * len(idx), dtype=np.uint8).tobytes()
        for i in range(len(idx)):   # the reference also wants S < q and a key that is not of small order
            if int.from_bytes(S[64 * i + 32:64 * i + 64], "little") >= O.ED_Q:
                return False
            if O.ed_mul(8, O.ed_decode(P[32 * i:32 * i + 32]))[0] % O.ED_P == 0:
                return False
        T = python_combination(P, S, H, zs)
        e8 = O.ed_mul(8, T)
        return e8[0] % O.ED_P == 0 and (e8[1] - e8[2]) % O.ED_P == 0

    def ref_accepts(idx):
        P, S, M, H = eddsa_subset(idx, pubs, sigs, msgs, hram, 32, 64, ED_MSG_LEN, 64)
        return O.ref_eddsa_verify_all(P, S, M, ED_MSG_LEN)
    for idx in ([g] for g in good):
        assert python_accepts(idx) and ref_accepts(idx), idx
    assert python_accepts(good) and ref_accepts(good)
    for b in bad:
        assert not python_accepts([b]) and not ref_accepts([b]), b
        mix = good[:2] + [b] + good[2:4]
        assert not python_accepts(mix) and not ref_accepts(mix), b


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/libecc_ref.so not built")
def test_python_pubkey_encoding_vs_reference_export():
    """ed_encode([a]B) is what libecc's eddsa_export_pub_key yields for the key held as the projective Weierstrass point
    [a]G (any representative); the point at infinity and a point off the curve behave as the GPU test expects"""
    rng = np.random.default_rng(31)
    p = O.ED_P
    o = Oracle("WEI25519")
    n = 24
    scal = [int.from_bytes(rng.integers(0, 256, size=32, dtype=np.uint8).tobytes(), "little") % O.ED_Q or 1 for _ in range(n)]
    aff, st = o.scalar_mult(b"".join(a.to_bytes(32, "big") for a in scal))
    assert set(st) == {0}
    prj = bytearray()
    for i in range(n):
        x, y = int.from_bytes(aff[64 * i:64 * i + 32], "big"), int.from_bytes(aff[64 * i + 32:64 * i + 64], "big")
        z = (int.from_bytes(rng.integers(0, 256, size=40, dtype=np.uint8).tobytes(), "big") % p or 1) if i % 2 else 1
        prj += (x * z % p).to_bytes(32, "big") + (y * z % p).to_bytes(32, "big") + z.to_bytes(32, "big")
    prj += (5).to_bytes(32, "big") + (7).to_bytes(32, "big") + (1).to_bytes(32, "big")   # not on the curve
    enc, ret = O.ref_eddsa_export_pub_key(bytes(prj))
    for i in range(n):
        assert ret[i] == 0 and enc[32 * i:32 * i + 32] == O.ed_encode(O.ed_mul(scal[i], O.ED_B)), i
    assert ret[n] == -1
