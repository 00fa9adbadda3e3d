#!/usr/bin/env python3
"""Ad-hoc throughput of the protocol entry points (host-pointer forms, so PCIe copies are included):
keygen (fixed-base scalar mult), ECDSA sign, ECDSA verify, ECC-CDH.  Not the driver's bench."""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libecc_amd  # noqa: E402
from oracles import CURVES, Oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="SECP256R1")
    ap.add_argument("--log2", type=int, default=18)
    a = ap.parse_args()
    n = 1 << a.log2
    q = CURVES[a.curve]["q"]
    ctx = libecc_amd.Context(0)
    cv = ctx.curve(a.curve)
    rng = np.random.default_rng(1)
    ql = cv.qlen

    def scalars():
        raw = rng.integers(0, 256, size=(n, ql + 8), dtype=np.uint8)
        return b"".join(((int.from_bytes(raw[i].tobytes(), "big") % (q - 1)) + 1).to_bytes(ql, "big") for i in range(n))

    privs, nonces = scalars(), scalars()
    dg = rng.integers(0, 256, size=n * 32, dtype=np.uint8).tobytes()
    res = {}
    for rep in range(2):
        t = time.time(); pubs, st = cv.scalar_mult(privs); res["keygen"] = n / (time.time() - t)
        assert set(st) == {0}
        t = time.time(); sigs, st = cv.ecdsa_sign(privs, nonces, dg, 32); res["sign"] = n / (time.time() - t)
        assert set(st) == {0}
        t = time.time(); ok = cv.ecdsa_verify(pubs, sigs, dg, 32); res["verify"] = n / (time.time() - t)
        assert set(ok) == {0}
        t = time.time(); sec, st = cv.ecccdh(privs, pubs[2 * cv.clen:] + pubs[:2 * cv.clen]); res["ecccdh"] = n / (time.time() - t)
    # parity spot check against the CPU oracle
    o = Oracle(a.curve)
    m = 16
    assert o.ecdsa_sign(privs[:m * ql], nonces[:m * ql], dg[:m * 32], 32)[0] == sigs[:m * 2 * ql]
    bad = bytearray(sigs[:m * 2 * ql]); bad[3] ^= 1
    assert cv.ecdsa_verify(pubs[:m * 2 * cv.clen], bytes(bad), dg[:m * 32], 32) == o.ecdsa_verify(pubs[:m * 2 * cv.clen], bytes(bad), dg[:m * 32], 32)
    print({k: f"{v / 1e6:.2f} M/s" for k, v in res.items()}, "batch", n, a.curve)
    if a.curve in ("WEI25519", "WEI448"):
        ln = cv.clen
        kk = rng.integers(0, 256, size=n * ln, dtype=np.uint8).tobytes()
        base = (9 if ln == 32 else 5).to_bytes(ln, "little") * n
        for rep in range(2):
            t = time.time(); pub, st = cv.xdh(kk, base); r1 = n / (time.time() - t)
            t = time.time(); sh, st2 = cv.xdh(kk[::-1], pub); r2 = n / (time.time() - t)
        assert set(st) == {0} and set(st2) == {0}
        print({"xdh_pubkey": f"{r1 / 1e6:.2f} M/s", "xdh_shared": f"{r2 / 1e6:.2f} M/s"})
    if a.curve == "WEI25519":
        import oracles as O
        m = 128
        items = [O.ed25519_sign(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), bytes([i]) * 16) for i in range(m)]
        reps = n // m
        P = b"".join(i[0] for i in items) * reps
        S = b"".join(i[1] for i in items) * reps
        H = b"".join(i[2] for i in items) * reps
        for rep in range(2):
            t = time.time(); ok = cv.eddsa_verify(P, S, H); r = len(ok) / (time.time() - t)
        assert set(ok) == {0}
        print({"ed25519_verify": f"{r / 1e6:.2f} M/s"})


if __name__ == "__main__":
    main()
