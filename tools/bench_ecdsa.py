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
    # the C entry points on preallocated, already-touched buffers (what a C caller sees); best of 3
    import ctypes as C
    L, cl = cv.L, cv.clen
    b_pub, b_st = C.create_string_buffer(b"\1" * (2 * cl * n)), C.create_string_buffer(b"\1" * n)
    b_sig, b_sec, b_ok = C.create_string_buffer(b"\1" * (2 * ql * n)), C.create_string_buffer(b"\1" * (cl * n)), C.create_string_buffer(b"\1" * n)

    def timed(name, fn):
        best = 0.0
        for _ in range(3):
            t = time.perf_counter()
            rc = fn()
            best = max(best, n / (time.perf_counter() - t))
            assert rc == 0, name
        res[name] = best
    timed("keygen", lambda: L.ec_prj_pt_mul_batch(cv.ctx.h, cv.h, n, privs, ql, None, b_pub, b_st))
    assert set(b_st.raw[:n]) == {0}
    pubs = b_pub.raw[:2 * cl * n]
    timed("sign", lambda: L.ec_ecdsa_sign_batch(cv.ctx.h, cv.h, n, privs, nonces, dg, 32, b_sig, b_st))
    assert set(b_st.raw[:n]) == {0}
    sigs = b_sig.raw[:2 * ql * n]
    timed("verify", lambda: L.ec_ecdsa_verify_batch(cv.ctx.h, cv.h, n, pubs, sigs, dg, 32, b_ok))
    assert set(b_ok.raw[:n]) == {0}
    peers = pubs[2 * cl:] + pubs[:2 * cl]
    timed("ecccdh", lambda: L.ec_ecccdh_derive_batch(cv.ctx.h, cv.h, n, privs, peers, b_sec, b_st))
    assert set(b_st.raw[:n]) == {0}
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
        b_out, b_out2 = C.create_string_buffer(b"\1" * (ln * n)), C.create_string_buffer(b"\1" * (ln * n))
        res.clear()
        timed("xdh_pubkey", lambda: L.ec_xdh_batch(cv.ctx.h, cv.h, n, kk, base, b_out, b_st))
        assert set(b_st.raw[:n]) == {0}
        pub, kr = b_out.raw[:ln * n], kk[::-1]
        timed("xdh_shared", lambda: L.ec_xdh_batch(cv.ctx.h, cv.h, n, kr, pub, b_out2, b_st))
        assert set(b_st.raw[:n]) == {0}
        print({k: f"{v / 1e6:.2f} M/s" for k, v in res.items()})
    if a.curve == "WEI25519":
        import oracles as O
        m = 128
        items = [O.ed25519_sign(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), bytes([i]) * 16) for i in range(m)]
        reps = n // m
        P = b"".join(i[0] for i in items) * reps
        S = b"".join(i[1] for i in items) * reps
        H = b"".join(i[2] for i in items) * reps
        res.clear()
        timed("ed25519_verify", lambda: L.ec_eddsa_verify_batch(cv.ctx.h, cv.h, n, P, S, H, 64, b_ok))
        assert set(b_ok.raw[:n]) == {0}
        # signing steps on random 64-byte hashes / 32-byte scalars (the hashes themselves are the caller's)
        rh, hh, aa = (rng.integers(0, 256, size=n * w, dtype=np.uint8).tobytes() for w in (64, 64, 32))
        b_R, b_S = C.create_string_buffer(b"\1" * (32 * n)), C.create_string_buffer(b"\1" * (32 * n))
        timed("ed25519_sign_R", lambda: L.ec_eddsa_sign_R_batch(cv.ctx.h, cv.h, n, rh, b_R, b_st))
        assert set(b_st.raw[:n]) == {0}
        timed("ed25519_sign_S", lambda: L.ec_eddsa_sign_S_batch(cv.ctx.h, cv.h, n, rh, hh, aa, b_S))
        r0 = int.from_bytes(rh[:64], "little") % O.ED_Q
        assert b_R.raw[:32] == O.ed_encode(O.ed_mul(r0, O.ED_B))
        assert b_S.raw[:32] == ((r0 + int.from_bytes(hh[:64], "little") * int.from_bytes(aa[:32], "little")) % O.ED_Q).to_bytes(32, "little")
        print({k: f"{v / 1e6:.2f} M/s" for k, v in res.items()})


if __name__ == "__main__":
    main()
