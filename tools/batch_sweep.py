#!/usr/bin/env python3
"""Throughput of the variable-base scalar multiplication against the batch size, 2^12 ... 2^22 items on ONE GPU (VERDICT round 3,
item 3): what a rank of a STRONG-scaling job sees when BASELINE.json's 2^20 items are split over 2 / 4 / 8 GPUs (2^19 / 2^18 / 2^17
items per GPU), and where the launch-bound regime starts.  Inputs resident in HBM, HIP-event timing on the launch stream, the
first result of every size checked against the oracle on 64 random items.

    python tools/batch_sweep.py [--curves SECP256R1,SECP384R1] [--lo 12 --hi 22] > gpurun_out/sweep.json
prints one JSON object {curve: [{log2n, items_per_s, ms_per_step, steps, rel_to_2^20}, ...]} and a markdown table on stderr."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libecc_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curves", default="SECP256R1,SECP384R1")
    ap.add_argument("--lo", type=int, default=12)
    ap.add_argument("--hi", type=int, default=22)
    a = ap.parse_args()
    import oracles as O
    O.build_oracle()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = libecc_amd.Context(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    out = {}
    for curve in a.curves.split(","):
        cp = O.CURVES[curve]
        slen, clen = (cp["q"].bit_length() + 7) // 8, (cp["p"].bit_length() + 7) // 8
        plen = 2 * clen
        cv = ctx.curve(curve)
        nmax = 1 << a.hi
        rng = np.random.default_rng(7)
        raw = rng.integers(0, 256, size=(2, nmax * slen), dtype=np.uint8)
        raw[:, ::slen] &= 0x7f if cp["q"].bit_length() % 8 == 0 else 0     # keep every scalar below q (top byte cleared on odd sizes)
        d_s, d_t = torch.from_numpy(raw[0]).to(dev), torch.from_numpy(raw[1]).to(dev)
        d_p = torch.empty(nmax * plen, dtype=torch.uint8, device=dev)
        d_o = torch.empty(nmax * plen, dtype=torch.uint8, device=dev)
        d_st = torch.empty(nmax, dtype=torch.uint8, device=dev)
        cv.scalar_mult_dev(nmax, d_t.data_ptr(), slen, None, d_p.data_ptr(), d_st.data_ptr(), stream.cuda_stream)   # P_i = [t_i]G
        torch.cuda.synchronize()
        rows = []
        for lg in range(a.lo, a.hi + 1):
            n = 1 << lg
            steps = int(max(4, min(64, (1 << 23) // n)))
            for _ in range(3):
                cv.scalar_mult_dev(n, d_s.data_ptr(), slen, d_p.data_ptr(), d_o.data_ptr(), d_st.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            # parity: 64 random items of this size against the oracle
            idx = np.random.default_rng(lg).choice(n, size=64, replace=False)
            sc, pt, got = (b"".join(bytes(t[w * i:w * i + w].cpu().numpy()) for i in idx) for t, w in ((d_s, slen), (d_p, plen), (d_o, plen)))
            exp, st = O.Oracle(curve).scalar_mult(sc, pt, slen)
            if got != exp or set(st) != {0}:
                raise SystemExit(f"PARITY FAILURE at {curve} n = 2^{lg}")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                cv.scalar_mult_dev(n, d_s.data_ptr(), slen, d_p.data_ptr(), d_o.data_ptr(), d_st.data_ptr(), stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            rows.append({"log2n": lg, "items_per_s": n / (ms * 1e-3), "ms_per_step": ms, "steps": steps})
        base = next((r["items_per_s"] for r in rows if r["log2n"] == 20), rows[-1]["items_per_s"])
        for r in rows:
            r["rel_to_2^20"] = r["items_per_s"] / base
        out[curve] = rows
        cv.free()
        del d_s, d_t, d_p, d_o, d_st
        torch.cuda.empty_cache()
    ctx.close()
    print(json.dumps(out))
    for curve, rows in out.items():
        print(f"\n### {curve}\n\n| batch | M items/s | ms/step | of the 2^20 rate |\n|---|---|---|---|", file=sys.stderr)
        for r in rows:
            print(f"| 2^{r['log2n']} | {r['items_per_s'] / 1e6:.2f} | {r['ms_per_step']:.3f} | {r['rel_to_2^20']:.3f} |", file=sys.stderr)


if __name__ == "__main__":
    main()
