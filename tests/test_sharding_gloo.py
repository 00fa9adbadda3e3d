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


def test_python_and_c_layers_partition_alike():
    """libecc_amd/shard.py (bench.py, torch.distributed) and ecamd_multi (C, one thread per device) cut a batch the same way"""
    import ctypes as C
    import libecc_amd
    L = libecc_amd.load_library()
    lo, hi = C.c_uint32(), C.c_uint32()
    for n in (0, 1, 7, 64, 1000, 1001, (1 << 20) + 3, (1 << 32) - 1):
        for world in (1, 2, 3, 8):
            for r in range(world):
                L.ecamd_multi_shard_range(n, r, world, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_range(n, r, world)


def _proto_worker(rank, world, port, n, q):
    """the protocol workloads of tools/bench_protocols.py sharded the same way: every rank verifies / derives its own
    contiguous shard, the per-item result bytes (1 byte per verification, clen bytes per secret) are gathered"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracles as O
    from oracles import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pubs, sigs, dg, privs, k, u, epubs, esigs, ehram = _proto_inputs(n)
    lo, hi = shard_range(n, rank, world)
    o = Oracle("SECP256R1")
    res = o.ecdsa_verify(pubs[64 * lo:64 * hi], sigs[64 * lo:64 * hi], dg[32 * lo:32 * hi], 32)
    sec, st = o.ecccdh(privs[32 * lo:32 * hi], pubs[64 * lo:64 * hi])
    o25 = Oracle("WEI25519")
    xo, xs = o25.xdh(k[32 * lo:32 * hi], u[32 * lo:32 * hi])
    er = o25.eddsa_verify(epubs[32 * lo:32 * hi], esigs[64 * lo:64 * hi], ehram[64 * lo:64 * hi])

    def t(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8)
    out = [all_gather_shards(t(res), n, 1), all_gather_shards(t(sec), n, 32), all_gather_shards(t(st), n, 1),
           all_gather_shards(t(xo), n, 32), all_gather_shards(t(xs), n, 1), all_gather_shards(t(er), n, 1)]
    if rank == 0:
        q.put([x.numpy().tobytes() for x in out])
    dist.barrier()
    dist.destroy_process_group()


def _proto_inputs(n):
    import hashlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracles as O
    from oracles import CURVES, Oracle
    rng = np.random.default_rng(123)
    o = Oracle("SECP256R1")
    q = CURVES["SECP256R1"]["q"]
    rs = lambda: ((int.from_bytes(rng.integers(0, 256, size=40, dtype=np.uint8).tobytes(), "big") % (q - 1)) + 1).to_bytes(32, "big")
    privs = b"".join(rs() for _ in range(n))
    ks = b"".join(rs() for _ in range(n))
    dg = rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes()
    sigs, st = o.ecdsa_sign(privs, ks, dg, 32)
    pubs, _ = o.scalar_mult(privs)
    sigs = bytearray(sigs)
    for i in range(0, n, 3):
        sigs[64 * i + 7] ^= 1
    k = rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes()
    u = bytearray(rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes())
    for i in range(n):
        u[32 * i + 31] &= 0x7f
    epubs, esigs, ehram = b"", b"", b""
    for i in range(n):
        a, sg, _ = O.ed25519_sign(bytes([i + 1]) * 32, b"msg%d" % i)
        if i % 4 == 1:
            sg = sg[:33] + bytes([sg[33] ^ 2]) + sg[34:]
        epubs += a
        esigs += sg
        ehram += hashlib.sha512(sg[:32] + a + b"msg%d" % i).digest()
    return pubs, bytes(sigs), dg, privs, k, bytes(u), epubs, esigs, ehram


def test_protocol_workloads_shard_like_the_headline_one():
    """ECDSA verification, ECC-CDH, X25519 and Ed25519 verification over two ranks: contiguous shards + gather of the per-item
    result bytes give the single-process answers in batch order (ragged split: 11 = 5 + 6)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracles import Oracle
    n = 11
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_proto_worker, args=(r, 2, port, n, q), daemon=True) for r in range(2)]
    try:
        for p in procs:
            p.start()
        got = q.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    pubs, sigs, dg, privs, k, u, epubs, esigs, ehram = _proto_inputs(n)
    o, o25 = Oracle("SECP256R1"), Oracle("WEI25519")
    sec, st = o.ecccdh(privs, pubs)
    xo, xs = o25.xdh(k, u)
    exp = [o.ecdsa_verify(pubs, sigs, dg, 32), sec, st, xo, xs, o25.eddsa_verify(epubs, esigs, ehram)]
    assert got == exp
    assert 0 < sum(exp[0]) < n and 0 < sum(exp[5]) < n


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


def _og_worker(rank, world, port, steps, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from libecc_amd.shard import OverlappedGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    og = OverlappedGather(world, n, torch.device("cpu"))
    seen = []
    for k in range(steps):
        buf = og.next_buffer()
        buf.fill_((17 * rank + k) % 251)          # stands for the kernels that fill the shard
        og.submit(buf)
        if k % 3 == 2:                            # a reader in the middle of the run must drain first
            og.drain()
            seen.append(og.gathered.clone())
    og.drain()
    seen.append(og.gathered.clone())
    if rank == 0:
        q.put([t.numpy().tobytes() for t in seen])
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_two_ranks():
    """bench.py's per-step all-gather (async, double-buffered shards) delivers every step's shards in rank order"""
    steps, world, n = 8, 2, 1000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_og_worker, args=(r, world, port, steps, q), daemon=True) for r in range(world)]
    try:
        for p in procs:
            p.start()
        got = q.get(timeout=180)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    checks = [k for k in range(steps) if k % 3 == 2] + [steps - 1]
    assert len(got) == len(checks)
    for blob, k in zip(got, checks):
        exp = b"".join(bytes([(17 * r + k) % 251]) * n for r in range(world))
        assert blob == exp, k


def test_overlapped_gather_single_rank_is_a_plain_buffer():
    from libecc_amd.shard import OverlappedGather
    og = OverlappedGather(1, 16, torch.device("cpu"))
    b0 = og.next_buffer()
    og.submit(b0)
    assert og.next_buffer() is b0 and og.gathered is None
    og.drain()
