"""Wycheproof: the runner of tests/wycheproof.py (the reference harness' verdict rules over the official JSON files).

* official vectors present ($WYCHEPROOF_VECTORS, tests/wycheproof/, the reference tree): the restatement oracle (CPU) and the GPU
  must agree with every "valid" / "invalid" verdict;
* official vectors absent (this snapshot, this environment): those tests SKIP as UNPINNED -- they do not pass;
* the self-made file in the Wycheproof schema (reference verdicts on this repository's crafted families) always runs, so the
  runner itself is exercised on the CPU and on the GPU."""
import hashlib
import json
import os

import pytest

import oracles as O
import wycheproof as W
from oracles import GOLDEN, Oracle

SELF = json.load(open(os.path.join(GOLDEN, "wycheproof_style_selfmade.json")))
HASHES = {"SHA224": hashlib.sha224, "SHA256": hashlib.sha256, "SHA384": hashlib.sha384, "SHA512": hashlib.sha512,
          "SHA3_224": hashlib.sha3_224, "SHA3_256": hashlib.sha3_256, "SHA3_384": hashlib.sha3_384, "SHA3_512": hashlib.sha3_512}


def selfmade(prefix):
    return [(k, v) for k, v in SELF.items() if k.startswith(prefix)]


def ed448_key_as_libecc_hashes_it(b):
    """libecc stores [4^-1 mod q]A at import and hashes the key RE-ENCODED from it, [4][4^-1]A = [j q + 1]A (sig/eddsa.c:925-937,
    1975-1981): the reference harness goes through eddsa_import_pub_key + ec_verify, so it hashes these bytes"""
    Q = O.E4_Q
    j = next(j for j in (1, 2, 3) if (j * Q + 1) % 4 == 0)
    pt = O.e4_decode(b)
    return b if pt is None else O.e4_encode(O.e4_mul(j * Q + 1, pt))


def hram_ed(kind, pubs, sigs, msgs):
    kl = 32 if kind == "Ed25519" else 57
    out = b""
    cache = {}
    for i, m in enumerate(msgs):
        key = pubs[kl * i:kl * (i + 1)]
        if kind == "Ed448":
            if key not in cache:
                cache[key] = ed448_key_as_libecc_hashes_it(key)
            key = cache[key]
        ra = sigs[2 * kl * i:2 * kl * i + kl] + key
        out += hashlib.sha512(ra + m).digest() if kind == "Ed25519" else hashlib.shake_256(O.ed_dom4(0, b"") + ra + m).digest(114)
    return out


# ---- back ends: the same four call-backs on the CPU restatement and on the GPU ----
class CpuBackend:
    def __init__(self):
        self.o = {}

    def oracle(self, curve):
        if curve not in self.o:
            self.o[curve] = Oracle(curve)
        return self.o[curve]

    def ecdsa(self, curve, h, pubs, sigs, msgs):
        dg = b"".join(HASHES[h](m).digest() for m in msgs)
        return self.oracle(curve).ecdsa_verify(pubs, sigs, dg, len(dg) // len(msgs))

    def eddsa(self, kind, pubs, sigs, msgs):
        return self.oracle("WEI25519" if kind == "Ed25519" else "WEI448").eddsa_verify(pubs, sigs, hram_ed(kind, pubs, sigs, msgs))

    def xdh(self, kind, k, u):
        return self.oracle("WEI25519" if kind == "X25519" else "WEI448").xdh(k, u)

    def derive(self, curve, privs, peers):
        return self.oracle(curve).ecccdh(privs, peers)

    def decompress(self, curve, comp):
        o = self.oracle(curve)
        cl = o.clen
        n = len(comp) // (cl + 1)
        xs = b"".join(comp[(cl + 1) * i + 1:(cl + 1) * (i + 1)] for i in range(n))
        y1, y2, st = o.y_from_x(xs)
        out, stat = b"", bytearray(st)
        for i in range(n):
            want = comp[(cl + 1) * i] & 1
            a, b = y1[cl * i:cl * (i + 1)], y2[cl * i:cl * (i + 1)]
            y = a if (a[-1] & 1) == want else b
            if st[i] == 0 and (y[-1] & 1) != want:
                stat[i] = 1
            out += xs[cl * i:cl * (i + 1)] + y if stat[i] == 0 else bytes(2 * cl)
        return out, bytes(stat)


class GpuBackend(CpuBackend):
    def __init__(self, ctx):
        super().__init__()
        self.ctx, self.cv = ctx, {}

    def curve(self, name):
        if name not in self.cv:
            self.cv[name] = self.ctx.curve(name)
        return self.cv[name]

    def close(self):
        for c in self.cv.values():
            c.free()

    def ecdsa(self, curve, h, pubs, sigs, msgs):
        dg = b"".join(HASHES[h](m).digest() for m in msgs)
        return self.curve(curve).ecdsa_verify(pubs, sigs, dg, len(dg) // len(msgs))

    def eddsa(self, kind, pubs, sigs, msgs):
        return self.curve("WEI25519" if kind == "Ed25519" else "WEI448").eddsa_verify(pubs, sigs, hram_ed(kind, pubs, sigs, msgs))

    def xdh(self, kind, k, u):
        return self.curve("WEI25519" if kind == "X25519" else "WEI448").xdh(k, u)

    def derive(self, curve, privs, peers):
        return self.curve(curve).ecccdh(privs, peers)

    def decompress(self, curve, comp):
        return self.curve(curve).decompress(comp)


def run_all(be, ecdsa_files, eddsa_files, xdh_files, ecdh_files):
    t = W.Tally()
    W.run_ecdsa(ecdsa_files, be.ecdsa, t)
    W.run_eddsa(eddsa_files, be.eddsa, t)
    W.run_xdh(xdh_files, be.xdh, t)
    W.run_ecdh_ecpoint(ecdh_files, be.derive, be.decompress, t)
    return t


def official_files():
    dirs = W.find_vectors()
    return (W.load("ecdsa_*_test.json", dirs), W.load("eddsa_test.json", dirs) + W.load("ed448_test.json", dirs),
            W.load("x25519_test.json", dirs) + W.load("x448_test.json", dirs), W.load("ecdh_*_ecpoint_test.json", dirs))


def test_der_reader():
    q = 32
    assert W.der_to_raw(bytes.fromhex("3006020101020102"), q) == (1).to_bytes(q, "big") + (2).to_bytes(q, "big")
    assert W.der_to_raw(bytes.fromhex("300702020001020102"), q) is None        # non-minimal INTEGER
    assert W.der_to_raw(bytes.fromhex("30060201ff020102"), q) is None          # negative
    assert W.der_to_raw(bytes.fromhex("300602010102010200"), q) is None        # trailing byte
    assert W.der_to_raw(bytes.fromhex("3080020101020102"), q) is None          # indefinite length
    big = "00" + "ff" * 32
    assert W.der_to_raw(bytes.fromhex("3026" + "0221" + big + "020101"), q) == b"\xff" * 32 + (1).to_bytes(q, "big")
    assert W.der_to_raw(bytes.fromhex("3027" + "0222" + "01" + "00" * 33 + "020101"), q) is None   # r does not fit q


def test_runner_on_selfmade_file_cpu():
    """the runner over the self-made file in the Wycheproof schema (reference verdicts), on the restatement oracle"""
    t = run_all(CpuBackend(), selfmade("ecdsa_"), selfmade("eddsa_") + selfmade("ed448_"), selfmade("x25519") + selfmade("x448"), selfmade("ecdh_"))
    assert not t.errors, t.errors[:5]
    assert t.performed > 700 and t.skipped == 0
    # the ECDH families (round 4): compressed peers, invalid-curve points, coordinates out of range, wrong encodings, edge private keys
    ecdh = W.Tally()
    be = CpuBackend()
    W.run_ecdh_ecpoint(selfmade("ecdh_"), be.derive, be.decompress, ecdh)
    assert not ecdh.errors and ecdh.performed > 200


def test_official_wycheproof_vectors_cpu():
    if not W.find_vectors():
        pytest.skip("UNPINNED: no Wycheproof test-vector files here (set WYCHEPROOF_DIR / WYCHEPROOF_VECTORS to a Wycheproof checkout or its testvectors/, or fill tests/wycheproof/)")
    t = run_all(CpuBackend(), *official_files())
    assert not t.errors, (len(t.errors), t.errors[:10])
    assert t.performed > 0


@pytest.mark.gpu
def test_runner_on_selfmade_file_gpu(gpu_ctx):
    be = GpuBackend(gpu_ctx)
    try:
        t = run_all(be, selfmade("ecdsa_"), selfmade("eddsa_") + selfmade("ed448_"), selfmade("x25519") + selfmade("x448"), selfmade("ecdh_"))
        assert not t.errors, t.errors[:5]
        assert t.performed > 700
    finally:
        be.close()


@pytest.mark.gpu
def test_official_wycheproof_vectors_gpu(gpu_ctx):
    if not W.find_vectors():
        pytest.skip("UNPINNED: no Wycheproof test-vector files here (set WYCHEPROOF_DIR / WYCHEPROOF_VECTORS to a Wycheproof checkout or its testvectors/, or fill tests/wycheproof/)")
    be = GpuBackend(gpu_ctx)
    try:
        t = run_all(be, *official_files())
        assert not t.errors, (len(t.errors), t.errors[:10])
        assert t.performed > 0
    finally:
        be.close()


@pytest.mark.gpu
def test_ecdh_ecpoint_runner_on_made_up_group(gpu_ctx):
    """the ECDH leg of the runner (uncompressed and SEC 1 compressed peer keys -> decompression -> derivation) on a group
    built here from the oracle's answers: GPU and CPU back ends agree with it"""
    import numpy as np
    rng = np.random.default_rng(91)
    curve, wname = "SECP256R1", "secp256r1"
    o = Oracle(curve)
    cl, ql = o.clen, o.qlen
    tests = []
    for i in range(40):
        d = rng.integers(1, 255, size=ql, dtype=np.uint8).tobytes()
        e = rng.integers(1, 255, size=ql, dtype=np.uint8).tobytes()
        peer, st = o.scalar_mult(e)
        sec, st2 = o.ecccdh(d, peer)
        assert st == b"\0" and st2 == b"\0"
        if i % 3 == 0:
            pub = bytes([2 + (peer[-1] & 1)]) + peer[:cl]
        else:
            pub = b"\x04" + peer
        res = "valid"
        if i % 10 == 7:
            pub = pub[:-1] + bytes([pub[-1] ^ 1]) if pub[0] == 4 else bytes([pub[0] ^ 1]) + pub[1:]
            res = "invalid" if pub[0] == 4 else "valid"
            if pub[0] != 4:                       # the other root: the negated point, same x coordinate of the result
                res = "valid"
        tests.append({"tcId": i + 1, "comment": "", "public": pub.hex(), "private": d.hex(), "shared": sec.hex(), "result": res, "flags": []})
    files = [("ecdh_%s_ecpoint_test.json" % wname, {"testGroups": [{"curve": wname, "encoding": "ecpoint", "type": "EcdhEcpointTest", "tests": tests}]})]
    for be in (CpuBackend(), GpuBackend(gpu_ctx)):
        t = W.Tally()
        W.run_ecdh_ecpoint(files, be.derive, be.decompress, t)
        assert not t.errors, t.errors[:5]
        assert t.performed == 40
        if isinstance(be, GpuBackend):
            be.close()
