"""Multi-GPU sharding of a batch: items are independent, so rank r of N
owns the contiguous index range [lo, hi) and no data-path collective is needed to compute.
When every rank must end up with all output points (north_star: "RCCL gather over xGMI for the
output points") one all-gather of the per-rank output buffers follows; with equal shard sizes it
is a single `all_gather_into_tensor`, otherwise shards are padded to the largest one.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced: sizes differ by at most one and concatenate to range(n)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


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
