#!/usr/bin/env python3
"""X448 over 2^20 seeded scalars on the GPU: rate and the SHA-256 of all outputs.  The inputs are those of round 2's pass r
(tools/gpu_r2r.sh), so the digest must equal the one in profiles/r2r_x448_ladder.txt whatever the field arithmetic underneath
(a456b4cb2b6eef54... -- a bit-exactness check of the whole batch that needs no reference run)."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libecc_amd  # noqa: E402

EXPECTED = "a456b4cb2b6eef541ddee959586cda9ec1696675149ef4ae6159810327b37fb4"


def main():
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(3)
    n = 1 << 20
    ctx = libecc_amd.Context(0)
    cv = ctx.curve("WEI448")
    k = torch.frombuffer(bytearray(rng.integers(0, 256, size=56 * n, dtype=np.uint8).tobytes()), dtype=torch.uint8).to(dev)
    u = torch.frombuffer(bytearray((5).to_bytes(56, "little") * n), dtype=torch.uint8).to(dev)
    o = torch.empty(56 * n, dtype=torch.uint8, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)

    def f():
        cv.xdh_dev(n, k.data_ptr(), u.data_ptr(), o.data_ptr(), st.data_ptr(), stream.cuda_stream)
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(3):
        f()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    digest = hashlib.sha256(bytes(o.cpu().numpy())).hexdigest()
    print(json.dumps({"ms_per_2^20": ms, "x448_per_s": n / (ms * 1e-3), "rejected": int(st.sum().item()), "sha256_of_outputs": digest,
                      "equals_round2_digest": digest == EXPECTED}))
    return 0 if digest == EXPECTED else 1


if __name__ == "__main__":
    sys.exit(main())
