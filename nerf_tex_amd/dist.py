"""Multi-GPU: rays shard embarrassingly, one process per GPU, no data-path collective except ONE gather of the
finished RGBA to the root.

The reference has no distribution at all; nothing couples two rays anywhere on the path (renderer.py:72-73 already
renders chunks of rays independently), so the sharded image is bit-identical to the single-GPU one by construction.

`ShardMap` is the shard map of include/nerftex.h (`ntx_shard_count`): the row-major pixel sequence is cut into runs of
`run_length` pixels, run q belongs to rank q % world.  `run_length = ceil(n / world)` = contiguous bands;
`run_length = image width` deals rows round-robin, which balances the rays the proxy culls.  On the GPU the gather is
`ntx_gather_image` (RCCL `ncclGather` through the C ABI, communicator created from a `ncclUniqueId` that rank 0
broadcasts over the existing torch.distributed group); CPU tensors (the gloo tests) go through `torch.distributed.gather`
with the same map.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple


class ShardMap:
    """run q of `run_length` consecutive pixels -> rank q % world, local run q // world."""

    def __init__(self, n_pixels: int, world: int, run_length: Optional[int] = None):
        if n_pixels < 0 or world < 1:
            raise ValueError("bad shard map")
        self.n, self.world = int(n_pixels), int(world)
        self.run = int(run_length) if run_length else max(1, -(-self.n // self.world))
        if self.run < 1:
            raise ValueError("run_length must be >= 1")

    def count(self, rank: int) -> int:
        runs = -(-self.n // self.run)
        if runs <= rank:
            return 0
        mine = (runs - 1 - rank) // self.world + 1
        last = rank + (mine - 1) * self.world
        return mine * self.run - ((runs * self.run - self.n) if last == runs - 1 else 0)

    @property
    def capacity(self) -> int:
        return self.count(0)

    @property
    def contiguous(self) -> bool:
        return self.run * self.world >= self.n

    def pixel_set(self, rank: int) -> Tuple[int, int, int, int]:
        """(pixel0, n_pixels, run_length, run_stride) of `rank` for `ntx_generate_rays_strided`."""
        return rank * self.run, self.count(rank), self.run, self.world * self.run

    def local_pixels(self, rank: int):
        """global pixel index of every local ray of `rank` (numpy int64)."""
        import numpy as np
        k = np.arange(self.count(rank), dtype=np.int64)
        return rank * self.run + (k // self.run) * (self.world * self.run) + k % self.run


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, count) of rank's band under the contiguous map (`ShardMap(n, world)`)."""
    m = ShardMap(n, world)
    return min(rank * m.run, n), m.count(rank)


class Comm:
    """`ntx_comm` (RCCL communicator of the C ABI) of this process, bootstrapped over torch.distributed: rank 0 draws
    the ncclUniqueId (`ntx_comm_unique_id`) and broadcasts its 128 bytes."""

    def __init__(self, device_index: int, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        if self.rank == 0:
            _lib.check(_lib.lib.ntx_comm_unique_id(ident))
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(box[0])
        handle = C.c_void_p()
        _lib.check(_lib.lib.ntx_comm_create(ident, self.world, self.rank, device_index, C.byref(handle)))
        self.handle = handle.value
        self.device = torch.device("cuda", device_index)
        self._staging = None

    def gather_image(self, local_rgba, shard: ShardMap, dst: int = 0):
        """[count(rank), 4] float32 on this GPU -> [n, 4] on `dst` (None elsewhere), on the current stream."""
        import torch
        from . import _lib
        local_rgba = local_rgba.contiguous()
        if local_rgba.shape[0] != shard.count(self.rank) or local_rgba.dtype != torch.float32:
            raise ValueError(f"rank {self.rank} must pass [{shard.count(self.rank)}, 4] float32, got {tuple(local_rgba.shape)}")
        image = staging = None
        if self.rank == dst:
            image = torch.empty((shard.n, 4), device=self.device, dtype=torch.float32)
            equal = all(shard.count(r) == shard.capacity for r in range(shard.world))
            if not (equal and shard.contiguous):
                need = shard.world * shard.capacity * 4
                if self._staging is None or self._staging.numel() < need:
                    self._staging = torch.empty(need, device=self.device, dtype=torch.float32)
                staging = self._staging
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.ntx_gather_image(self.handle, local_rgba.data_ptr(), shard.n, shard.run,
                                                 image.data_ptr() if image is not None else None,
                                                 staging.data_ptr() if staging is not None else None, dst,
                                                 torch.cuda.current_stream(self.device).cuda_stream))
        return image

    def close(self) -> None:
        if self.handle:
            from . import _lib
            _lib.lib.ntx_comm_destroy(self.handle)
            self.handle = None


def gather_image(local_rgba, shard, dst: int = 0, group=None, comm: Optional[Comm] = None):
    """Gather every rank's RGBA shard to `dst` in pixel order.  `shard`: a ShardMap (or an int n_total = contiguous
    bands).  With `comm` (GPU shards): ntx_gather_image / RCCL behind the C ABI.  Without: torch.distributed.gather (gloo in
    the CPU plumbing tests).  Returns the [n_total, 4] image on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not isinstance(shard, ShardMap):
        shard = ShardMap(int(shard), world)
    if local_rgba.is_cuda and comm is not None:
        return comm.gather_image(local_rgba, shard, dst)
    # no C-ABI communicator: the same exchange through torch.distributed (gloo for the CPU plumbing tests; for GPU shards
    # only as the caller's explicit choice, e.g. bench.py when ntx_comm_create fails on a box -- it says so in its line)
    if world == 1:
        return local_rgba
    cap = shard.capacity
    pad = torch.zeros((cap,) + tuple(local_rgba.shape[1:]), dtype=local_rgba.dtype, device=local_rgba.device)
    pad[: local_rgba.shape[0]] = local_rgba
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    image = torch.empty((shard.n,) + tuple(local_rgba.shape[1:]), dtype=local_rgba.dtype, device=local_rgba.device)
    for r in range(world):
        image[torch.as_tensor(shard.local_pixels(r), device=local_rgba.device)] = bufs[r][: shard.count(r)]
    return image
