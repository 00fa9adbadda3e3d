"""Multi-GPU sharding of a batch: items are independent, so rank r of N
owns the contiguous index range [lo, hi) and no data-path collective is needed to compute.
When every rank must end up with all output points (north_star: "RCCL gather over xGMI for the
output points") one all-gather of the per-rank output buffers follows; with equal shard sizes it
is a single `all_gather_into_tensor`, otherwise shards are padded to the largest one.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced: rank r of N owns [r*n//N, (r+1)*n//N) -- sizes differ by at most one and concatenate to
    range(n).  The same partition as the C layer's ecamd_multi_shard_range (libecc_amd/csrc/ecamd_multi.cpp)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def all_gather_shards(local, n, item_bytes, group=None):
    """local: uint8 tensor with this rank's (hi-lo)*item_bytes output bytes.
    Returns a uint8 tensor of n*item_bytes bytes holding every rank's outputs in batch order."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(n, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes) * item_bytes
    pad = torch.zeros(mx, dtype=torch.uint8, device=local.device)
    pad[:local.numel()] = local
    buf = torch.empty(world * mx, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = [buf[r * mx:r * mx + (hi - lo) * item_bytes] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts)


class OverlappedGather:
    """One all-gather of equal-size output shards per step, overlapped with the next step's kernels.

    The collective is issued with async_op=True (on RCCL's own stream for GPU tensors; it starts once the work
    already enqueued on the current stream has finished).  Shards AND gathered outputs are double-buffered, so
    a shard is never overwritten while it is still being gathered and two gathers in flight never write the
    same tensor (backends such as gloo may complete them out of order):

        buf = og.next_buffer()     # waits (stream-level, not a host block) for the gather that last read buf
        ... enqueue the kernels that fill buf ...
        og.submit(buf)             # async all_gather_into_tensor(<next gathered buffer>, buf)
        ...
        og.drain()                 # before reading og.gathered (the latest step's result) / closing a timed region

    With world == 1 there is one buffer and no collective."""

    def __init__(self, world, shard_numel, device, group=None):
        self.world = world
        self.group = group
        nb = 2 if world > 1 else 1
        self.bufs = [torch.empty(shard_numel, dtype=torch.uint8, device=device) for _ in range(nb)]
        self.gbufs = [torch.empty(world * shard_numel, dtype=torch.uint8, device=device) for _ in range(nb)] if world > 1 else []
        self.gathered = None       # result of the most recently submitted step, valid after drain()
        self.pending = []
        self.k = 0

    def next_buffer(self):
        buf = self.bufs[self.k % len(self.bufs)]
        self.k += 1
        if self.world > 1 and len(self.pending) >= len(self.bufs):
            self._wait(self.pending.pop(0))
        return buf

    @staticmethod
    def _wait(work):
        if work is not None:
            work.wait()

    def submit(self, buf):
        if self.world > 1:
            g = self.gbufs[(self.k - 1) % len(self.gbufs)]
            try:
                work = dist.all_gather_into_tensor(g, buf, group=self.group, async_op=True)
            except (RuntimeError, NotImplementedError):   # a backend without async collectives: gather in line
                dist.all_gather_into_tensor(g, buf, group=self.group)
                work = None
            self.pending.append(work)
            self.gathered = g

    def drain(self):
        while self.pending:
            self._wait(self.pending.pop(0))
