"""world_size-2 gloo test (CPU) of the multi-GPU path: per-rank contiguous shards + all-gather of
the outputs reproduce the single-process result in batch order.  The compute stand-in is the CPU
oracle (allowed in tests); on the GPU box bench.py runs the same shard/gather code over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libecc_amd.shard import all_gather_shards, shard_range


def test_shard_ranges_cover_batch():
    for n in (0, 1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def _worker(rank, world, port, n, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracles import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle("SECP256R1")
    rng = np.random.default_rng(99)
    sc = rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes()
    lo, hi = shard_range(n, rank, world)
    out, st = o.scalar_mult(sc[32 * lo:32 * hi])
    loc = torch.frombuffer(bytearray(out), dtype=torch.uint8)
    full = all_gather_shards(loc, n, 64)
    stf = all_gather_shards(torch.frombuffer(bytearray(st), dtype=torch.uint8), n, 1)
    if rank == 0:
        q.put((full.numpy().tobytes(), stf.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    from oracles import Oracle
    n = 13  # ragged: 7 + 6
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q), daemon=True) for r in range(2)]
    try:
        for p in procs:
            p.start()
        got = q.get(timeout=180)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:  # never leave a worker behind: a live child would block interpreter exit
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    rng = np.random.default_rng(99)
    sc = rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes()
    assert got == Oracle("SECP256R1").scalar_mult(sc)
