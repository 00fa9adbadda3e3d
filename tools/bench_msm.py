#!/usr/bin/env python3
"""Ed25519 whole-batch verification: the multi-scalar multiplication (ec_eddsa_verify_all_batch_dev) against the item-by-item
verification (ec_eddsa_verify_batch_dev), inputs resident in HBM, HIP events on the launch stream: one JSON object."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libecc_amd  # noqa: E402
import oracles as O  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    stream = torch.cuda.Stream(device=dev)
    base = 251
    pubs, sigs, hram = bytearray(), bytearray(), bytearray()
    for _ in range(base):
        A, sg, h = O.ed25519_sign(rng.integers(0, 256, size=32, dtype=np.uint8).tobytes(), rng.integers(0, 256, size=32, dtype=np.uint8).tobytes())
        pubs += A
        sigs += sg
        hram += h

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    logs = [int(x) for x in os.environ.get("MSM_LOG2", "16,17,18,19,20").split(",")]
    ks = [int(x) for x in os.environ.get("MSM_K", "0,1,2,4,8,16").split(",")]
    ctx = libecc_amd.Context(0)
    cv = ctx.curve("WEI25519")
    for lg in logs:
        n = 1 << lg
        reps = (n + base - 1) // base
        tp = torch.frombuffer(bytearray((bytes(pubs) * reps)[:32 * n]), dtype=torch.uint8).to(dev)
        ts = torch.frombuffer(bytearray((bytes(sigs) * reps)[:64 * n]), dtype=torch.uint8).to(dev)
        th = torch.frombuffer(bytearray((bytes(hram) * reps)[:64 * n]), dtype=torch.uint8).to(dev)
        res = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        verdict = torch.full((1,), 7, dtype=torch.uint8, device=dev)
        row = {}
        ms = timed(lambda: cv.eddsa_verify_dev(n, tp.data_ptr(), ts.data_ptr(), th.data_ptr(), res.data_ptr(), stream.cuda_stream))
        assert int(res.max().item()) == 0
        row["per_item_ms"] = ms
        row["per_item_per_s"] = n / (ms * 1e-3)
        for k in ks:
            ctx.set_eddsa_msm(2, 0, k)
            ms = timed(lambda: cv.eddsa_verify_all_dev(n, tp.data_ptr(), ts.data_ptr(), th.data_ptr(), verdict.data_ptr(), stream.cuda_stream))
            assert int(verdict.item()) == 0, (lg, k)
            row[f"msm_k{k}_ms"] = ms
            row[f"msm_k{k}_speedup"] = row["per_item_ms"] / ms
        # a batch with one bad item is rejected
        th[64 * (n // 2) + 3] ^= 1
        ctx.set_eddsa_msm(2, 0, 0)
        cv.eddsa_verify_all_dev(n, tp.data_ptr(), ts.data_ptr(), th.data_ptr(), verdict.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        assert int(verdict.item()) == 1
        out[f"2^{lg}"] = row
    cv.free()
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
