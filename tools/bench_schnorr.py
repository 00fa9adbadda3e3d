#!/usr/bin/env python3
"""Schnorr-type whole-batch verification: the multi-scalar form (ec_schnorr_verify_all_batch) against the item form the typed layer
runs for BIP0340 / ECFSDSA ([s]G from the comb, [q - e]Y by the window pipeline, the addition; libecc_amd_compat.c:schnorr_group), both
through the host-pointer C ABI on one MI355X, wall clock (copies included on both sides).  Items are made on the GPU (keys and nonce
points as fixed-base multiplications) and checked valid by both forms before timing.

usage: python tools/bench_schnorr.py [--curves SECP256K1,SECP256R1] [--log2 17,18,19,20] [--k 0] > profiles/<name>.md"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libecc_amd  # noqa: E402
from oracles import CURVES  # noqa: E402


def make(cv, curve, n, rng, even):
    c = CURVES[curve]
    q, p = c["q"], c["p"]
    cl, ql = cv.clen, cv.qlen
    raw = rng.integers(0, 256, size=(3, n, ql + 8), dtype=np.uint8)
    ints = [[(int.from_bytes(bytes(raw[t, i]), "big") % (q - 1)) + 1 for i in range(n)] for t in range(3)]
    x, k, e = ints
    be = lambda v: v.to_bytes(ql, "big")
    Y, st = cv.scalar_mult(b"".join(be(v) for v in x))
    R, st2 = cv.scalar_mult(b"".join(be(v) for v in k))
    assert set(st) == {0} and set(st2) == {0}
    if even:
        R = bytearray(R)
        for i in range(n):
            if R[2 * cl * (i + 1) - 1] & 1:
                y = int.from_bytes(R[2 * cl * i + cl:2 * cl * (i + 1)], "big")
                R[2 * cl * i + cl:2 * cl * (i + 1)] = (p - y).to_bytes(cl, "big")
                k[i] = q - k[i]
        R = bytes(R)
    s = b"".join(be((k[i] + e[i] * x[i]) % q) for i in range(n))
    ne = b"".join(be((q - v) % q) for v in e)
    rx = b"".join(R[2 * cl * i:2 * cl * i + cl] for i in range(n))
    return s, ne, Y, R, rx


def best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curves", default="SECP256K1,SECP256R1,SECP384R1")
    ap.add_argument("--log2", default="16,17,18,19,20")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--k", default="0", help="comma list of items per lane to try (0 = the library's choice)")
    a = ap.parse_args()
    ctx = libecc_amd.Context(0)
    rng = np.random.default_rng(11)
    print("# Schnorr-type whole-batch verification: multi-scalar form against the item form (tools/bench_schnorr.py; one MI355X, host-pointer C ABI, wall clock, best of %d)\n" % a.reps)
    print("| curve | items | K | item form ms | multi-scalar ms (points) | ratio | multi-scalar ms (abscissae, lift_x on the device) | ratio | M items/s (multi-scalar, points) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for curve in a.curves.split(","):
        cv = ctx.curve(curve)
        nmax = 1 << max(int(v) for v in a.log2.split(","))
        lift = cv.schnorr_msm_available(1)
        s, ne, Y, R, rx = make(cv, curve, nmax, rng, lift)
        cl, ql = cv.clen, cv.qlen
        for lg in (int(v) for v in a.log2.split(",")):
            n = 1 << lg
            S, NE, YY, RR, RX = s[:ql * n], ne[:ql * n], Y[:2 * cl * n], R[:2 * cl * n], rx[:cl * n]

            def item_form():
                A, stA = cv.scalar_mult(S)
                B, stB = cv.scalar_mult(NE, YY)
                W, stW = cv.pt_add(A, B)
                return W == RR and set(stW) == {0}
            t_item, ok = best(item_form, a.reps)
            assert ok
            for k in (int(v) for v in a.k.split(",")):
                if k:
                    os.environ["ECAMD_SCHNORR_MSM_K"] = str(k)
                else:
                    os.environ.pop("ECAMD_SCHNORR_MSM_K", None)
                cv.schnorr_verify_all(S, NE, YY, RR, 0)   # scratch growth outside the timed region
                t_pts, ok = best(lambda: cv.schnorr_verify_all(S, NE, YY, RR, 0), a.reps)
                assert ok
                if lift:
                    t_abs, ok = best(lambda: cv.schnorr_verify_all(S, NE, YY, RX, 1), a.reps)
                    assert ok
                print(f"| {curve} | 2^{lg} | {k or 'auto'} | {t_item * 1e3:.2f} | {t_pts * 1e3:.2f} | {t_item / t_pts:.2f} | "
                      + (f"{t_abs * 1e3:.2f} | {t_item / t_abs:.2f}" if lift else "- | -") + f" | {n / t_pts / 1e6:.1f} |", flush=True)
        cv.free()
    ctx.close()


if __name__ == "__main__":
    main()
