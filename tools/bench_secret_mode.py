#!/usr/bin/env python3
"""Cost of the secret-scalar mode (ecamd_ctx_set_secret_scalars: complete formulas + masked full-scan look-ups) and of
blinded scalars against the default kernels, inputs resident in HBM, timed with HIP events: one JSON object."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libecc_amd  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    rng = np.random.default_rng(5)
    stream = torch.cuda.Stream(device=dev)

    def t(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

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

    for curve, n in (("SECP256R1", 1 << 18), ("SECP384R1", 1 << 16), ("WEI25519", 1 << 16)):
        row = {"batch": n}
        for mode in ("default", "secret"):
            ctx = libecc_amd.Context(0)
            if mode == "secret":
                ctx.set_secret_scalars(True)
            cv = ctx.curve(curve)
            ql, cl = cv.qlen, cv.clen
            sc = rng.integers(0, 256, size=ql * n, dtype=np.uint8).tobytes()
            base, st = cv.scalar_mult(sc)
            d_sc, d_base = t(sc), t(base)
            d_out, d_st = torch.empty(2 * cl * n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ms = timed(lambda: cv.scalar_mult_dev(n, d_sc.data_ptr(), ql, None, d_out.data_ptr(), d_st.data_ptr(), stream.cuda_stream))
            row[f"{mode}_fixed_base_per_s"] = n / (ms * 1e-3)
            ms = timed(lambda: cv.scalar_mult_dev(n, d_sc.data_ptr(), ql, d_base.data_ptr(), d_out.data_ptr(), d_st.data_ptr(), stream.cuda_stream))
            row[f"{mode}_variable_base_per_s"] = n / (ms * 1e-3)
            if mode == "default":
                # a blinded scalar m + b #E as the device sees it: 2 |q| + 8 bits
                long_sc = rng.integers(0, 256, size=(2 * ql + 1) * n, dtype=np.uint8).tobytes()
                d_long = t(long_sc)
                torch.cuda.synchronize()
                ms = timed(lambda: cv.scalar_mult_dev(n, d_long.data_ptr(), 2 * ql + 1, d_base.data_ptr(), d_out.data_ptr(), d_st.data_ptr(),
                                                      stream.cuda_stream))
                row["blinded_size_scalar_variable_base_per_s"] = n / (ms * 1e-3)
            cv.free()
            ctx.close()
        out[curve] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
