"""Multi-GPU: rays shard by contiguous row-major range, one process per GPU, no data-path collective
except ONE gather of the finished RGBA to rank 0.

The reference has no distribution at all; nothing couples two rays anywhere on the path
(renderer.py:72-73 already renders chunks of rays independently), so the sharded image is
bit-identical to the single-GPU one by construction.
"""

from __future__ import annotations

from typing import Optional, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, count) of rank's contiguous share of n rays; the first n % world ranks get one more."""
    base, rem = divmod(n, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_image(local_rgba, n_total: int, dst: int = 0, group=None):
    """Gather every rank's [n_local, 4] RGBA shard to `dst` (RCCL `ncclGather`-equivalent on the nccl
    backend: each peer sends straight to the root over its own xGMI link; gloo on CPU in tests).
    Returns the [n_total, 4] image on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local_rgba
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    width = max(counts)
    pad = local_rgba
    if local_rgba.shape[0] != width:            # uneven tail: pad to a common message size
        pad = torch.zeros((width,) + tuple(local_rgba.shape[1:]), dtype=local_rgba.dtype, device=local_rgba.device)
        pad[: local_rgba.shape[0]] = local_rgba
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)
