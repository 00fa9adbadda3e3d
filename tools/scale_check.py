#!/usr/bin/env python3
"""One-command correctness check of the multi-GPU layers for the first lease of a node with several GPUs.
No such node was available while this was written; on one GPU it runs the 1-rank legs and says so.

  python tools/scale_check.py [--gpus 1,2,4,8] [--log2 18]

For every N in the list (<= visible devices):
  1. C layer (`ecamd_multi_*` over N DISTINCT devices): a 2^log2-item secp256r1 batch with edge items -- outputs and status bytes must
     equal the single-device call's and the shards must be the contiguous ranges of `ecamd_multi_shard_range`; then each rank
     multiplies ITS shard with device-resident buffers on its context's stream (`ec_prj_pt_mul_batch_dev`), and
     `ecamd_multi_allgather` -- one RCCL all-gather, issued without any host synchronisation after the producers -- must leave the
     whole output on every device, equal to the single-device result.
  2. Process layer: `python bench.py --gpus N --steps 3 --warmup 1 --no-secondary --no-traffic` (one rank per GPU over RCCL); its JSON
     line must report world_size N and N distinct devices.
Prints one line per check and a JSON summary; exit status 0 iff everything held."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402

import libecc_amd  # noqa: E402
from oracles import CURVES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--log2", type=int, default=18)
    ap.add_argument("--no-bench", action="store_true")
    a = ap.parse_args()
    have = torch.cuda.device_count()
    want = [int(x) for x in a.gpus.split(",") if int(x) <= have]
    skipped = [int(x) for x in a.gpus.split(",") if int(x) > have]
    curve = "SECP256R1"
    q = CURVES[curve]["q"]
    n = 1 << a.log2
    rng = np.random.default_rng(5)
    sc = bytearray(rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes())
    for i, v in enumerate((0, 1, q - 1, q, q + 1, 2**256 - 1)):          # edge scalars at the shard borders
        for pos in (i, n // 2 - 3 + i, n - 6 + i):
            sc[32 * pos:32 * pos + 32] = v.to_bytes(32, "big")
    sc = bytes(sc)
    ctx0 = libecc_amd.Context(0)
    cv0 = ctx0.curve(curve)
    pts, st = cv0.scalar_mult(rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes())
    assert set(st) <= {0}
    pts = bytearray(pts)
    pts[64 * 7 + 63] ^= 1                                                 # one point off the curve
    pts = bytes(pts)
    ref = cv0.scalar_mult(sc, pts)
    cv0.free()
    ctx0.close()
    ok, report = True, []

    def check(name, cond, extra=""):
        nonlocal ok
        ok = ok and bool(cond)
        print(("ok      " if cond else "FAILED  ") + name + (" -- " + extra if extra else ""))
        report.append({"check": name, "ok": bool(cond), "info": extra})

    L = libecc_amd.load_library()
    for N in want:
        m = libecc_amd.Multi(list(range(N)))
        mc = m.curve(curve)
        got = mc.scalar_mult(sc, pts)
        check(f"N={N}: ecamd_multi_prj_pt_mul_batch equals the single-device call", got == ref)
        ranges = [m.shard_range(n, r) for r in range(N)]
        check(f"N={N}: shards are contiguous and cover the batch", ranges[0][0] == 0 and ranges[-1][1] == n and
              all(ranges[r][1] == ranges[r + 1][0] for r in range(N - 1)), str(ranges))
        # device-resident shards + the RCCL all-gather (equal shard sizes: n is a power of two, N divides it)
        per = n // N
        if per * N == n:
            sends, recvs, keep = [], [], []
            for r in range(N):
                dev = torch.device("cuda", r)
                lo = r * per
                d_s = torch.frombuffer(bytearray(sc[32 * lo:32 * (lo + per)]), dtype=torch.uint8).to(dev)
                d_p = torch.frombuffer(bytearray(pts[64 * lo:64 * (lo + per)]), dtype=torch.uint8).to(dev)
                d_o = torch.zeros(64 * per, dtype=torch.uint8, device=dev)
                d_t = torch.zeros(per, dtype=torch.uint8, device=dev)
                d_r = torch.zeros(64 * n, dtype=torch.uint8, device=dev)
                keep += [d_s, d_p, d_o, d_t, d_r]
                sends.append(d_o)
                recvs.append(d_r)
            torch.cuda.synchronize()
            for r in range(N):
                ctx = L.ecamd_multi_ctx(m.h, r)
                cvh = L.ecamd_multi_curve_handle(mc.h, r)
                d_s, d_p, d_o, d_t, _ = keep[5 * r:5 * r + 5]
                rc = L.ec_prj_pt_mul_batch_dev(C.c_void_p(ctx), C.c_void_p(cvh), per, C.c_void_p(d_s.data_ptr()), 32, C.c_void_p(d_p.data_ptr()),
                                               C.c_void_p(d_o.data_ptr()), C.c_void_p(d_t.data_ptr()), None)
                check(f"N={N}: rank {r} enqueued its shard on its own stream", rc == 0)
            sp = (C.c_void_p * N)(*[t.data_ptr() for t in sends])
            rp = (C.c_void_p * N)(*[t.data_ptr() for t in recvs])
            rc = L.ecamd_multi_allgather(m.h, sp, rp, C.c_size_t(64 * per))   # no host synchronisation in between
            check(f"N={N}: ecamd_multi_allgather returned 0", rc == 0, "" if rc == 0 else (L.ecamd_last_error() or b"").decode())
            for r in range(N):
                check(f"N={N}: device {r} holds the whole output after the gather", recvs[r].cpu().numpy().tobytes() == ref[0])
        mc.free()
        m.close()
        if not a.no_bench:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--steps", "3", "--warmup", "1", "--no-secondary",
                                "--no-traffic", "--no-cpu-baseline", "--parity-items", "2048"], capture_output=True, text=True, timeout=900)
            line = None
            for ln in p.stdout.splitlines():
                if ln.startswith("{"):
                    line = json.loads(ln)
            good = p.returncode == 0 and line is not None and line["n_gpus"] == N and line["config"]["world_size"] == N and \
                len(set(line["config"]["rank_devices"])) == N
            check(f"N={N}: bench.py ran one rank per GPU", good, f"{line['value'] / 1e6:.1f} M scalar mults/s" if line else p.stderr[-300:])
    for N in skipped:
        print(f"skipped N={N}: only {have} GPU(s) visible")
    print(json.dumps({"all_ok": ok, "visible_gpus": have, "checked": want, "skipped": skipped, "checks": report}))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
