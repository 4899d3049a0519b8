"""Multi-GPU: rays shard embarrassingly, one process per GPU, no data-path collective except ONE gather of the
finished RGBA to the root.

The reference has no distribution at all; nothing couples two rays anywhere on the path (renderer.py:72-73 already
renders chunks of rays independently), so the sharded image is bit-identical to the single-GPU one by construction.

`ShardMap` is the shard map of include/nerftex.h (`ntx_shard_count`): the row-major pixel sequence is cut into runs of
`run_length` pixels, run q belongs to rank q % world.  `run_length = ceil(n / world)` = contiguous bands;
`run_length = image width` deals rows round-robin, which balances the rays the proxy culls.  On the GPU the gather is
`ntx_gather_image` (RCCL `ncclGather` through the C ABI, communicator created from a `ncclUniqueId` that rank 0
broadcasts over the existing torch.distributed group).  Without a communicator (CPU tensors in the gloo tests; GPU shards
only as a caller's explicit fallback) the SAME plan -- `ntx_gather_plan`: per-rank counts and block offsets, one gather when
the counts are equal, else exact-count send/recv; `ntx_unshard_map`: the staging slot of every pixel -- is executed through
`torch.distributed`, so the library's index arithmetic runs in both cases.
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

    def ray_index(self, rank: int) -> Tuple[int, int, int]:
        """(index0, run_length, run_stride): the global index of `rank`'s local rays, for `Renderer(..., ray_index=...)` /
        `ntx_render_opts` -- the generators behind jitter and noise are keyed by the PIXEL, not by its place in a shard."""
        return rank * self.run, self.run, self.world * self.run

    def plan(self):
        """`ntx_gather_plan` of this map: (counts [world], block offsets [world] in pixel slots, equal, direct)."""
        from . import _lib
        counts = (C.c_int64 * self.world)(); offs = (C.c_int64 * self.world)()
        eq, direct = C.c_int(), C.c_int()
        _lib.check(_lib.lib.ntx_gather_plan(self.n, self.run, self.world, counts, offs, C.byref(eq), C.byref(direct)))
        return list(counts), list(offs), bool(eq.value), bool(direct.value)

    def unshard_map(self):
        """`ntx_unshard_map`: for every pixel, the pixel slot of the gather destination that holds it (numpy int64 [n])."""
        import numpy as np
        from . import _lib
        src = np.empty(self.n, dtype=np.int64)
        _lib.check(_lib.lib.ntx_unshard_map(self.n, self.run, self.world, src.ctypes.data_as(C.POINTER(C.c_int64))))
        return src

    def local_pixels(self, rank: int):
        """global pixel index of every local ray of `rank` (numpy int64)."""
        import numpy as np
        k = np.arange(self.count(rank), dtype=np.int64)
        return rank * self.run + (k // self.run) * (self.world * self.run) + k % self.run


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[first, count) of rank's band under the contiguous map (`ShardMap(n, world)`)."""
    m = ShardMap(n, world)
    return min(rank * m.run, n), m.count(rank)


class CommUnavailable(RuntimeError):
    """Raised by `Comm(...)` on EVERY rank alike when any rank cannot create its communicator."""


class Comm:
    """`ntx_comm` (RCCL communicator of the C ABI) of this process, bootstrapped over torch.distributed: rank 0 draws
    the ncclUniqueId (`ntx_comm_unique_id`) and broadcasts its 128 bytes.

    The ranks AGREE before any of them enters `ncclCommInitRank` (which blocks until all peers have joined): every rank runs
    `ntx_comm_preflight` (librccl loadable, device valid), rank 0 also draws the id, and one all_gather of (ok, message)
    decides for everybody -- a rank whose librccl cannot be loaded, or a rank 0 that cannot draw the id, makes ALL ranks raise
    `CommUnavailable` instead of leaving its peers in mismatched collectives.  What is left is a failure inside
    ncclCommInitRank itself; callers bound that with a deadline (bench.py: watchdog per rank + launcher timeout)."""

    def __init__(self, device_index: int, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.handle = None
        ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        err = None
        if _lib.lib.ntx_comm_preflight(device_index) != _lib.NTX_OK:
            err = _lib.lib.ntx_last_error().decode("utf-8", "replace")
        elif self.rank == 0 and _lib.lib.ntx_comm_unique_id(ident) != _lib.NTX_OK:
            err = _lib.lib.ntx_last_error().decode("utf-8", "replace")
        if self.world > 1:
            votes = [None] * self.world
            dist.all_gather_object(votes, (err, bytes(ident) if self.rank == 0 else None), group=group)
            bad = {r: v[0] for r, v in enumerate(votes) if v[0] is not None}
            if bad:
                raise CommUnavailable("; ".join(f"rank {r}: {m}" for r, m in sorted(bad.items())))
            ident = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(votes[0][1])
        elif err is not None:
            raise CommUnavailable(f"rank 0: {err}")
        handle = C.c_void_p()
        import time
        t0 = time.perf_counter()
        _lib.check(_lib.lib.ntx_comm_create(ident, self.world, self.rank, device_index, C.byref(handle)))
        self.init_seconds = time.perf_counter() - t0          # ncclCommInitRank (blocks until every peer has joined)
        self.version = int(_lib.lib.ntx_comm_version())       # ncclGetVersion: 22606 = 2.26.6
        self.handle = handle.value
        self.device = torch.device("cuda", device_index)
        self.library = _lib.lib.ntx_comm_library().decode("utf-8", "replace")
        self._staging = None

    def gather_image(self, local_rgba, shard: ShardMap, dst: int = 0, force_exchange: bool = False):
        """[count(rank), 4] float32 on this GPU -> [n, 4] on `dst` (None elsewhere), on the current stream.  `force_exchange`
        (NTX_GATHER_FORCE_EXCHANGE): the grouped send / recv + un-shard branch whatever the counts (tests on one rank)."""
        import torch
        from . import _lib
        local_rgba = local_rgba.contiguous()
        if local_rgba.shape[0] != shard.count(self.rank) or local_rgba.dtype != torch.float32:
            raise ValueError(f"rank {self.rank} must pass [{shard.count(self.rank)}, 4] float32, got {tuple(local_rgba.shape)}")
        image = staging = None
        if self.rank == dst:
            image = torch.empty((shard.n, 4), device=self.device, dtype=torch.float32)
            equal = all(shard.count(r) == shard.capacity for r in range(shard.world))
            if force_exchange or not (equal and shard.contiguous):
                need = shard.world * shard.capacity * 4
                if self._staging is None or self._staging.numel() < need:
                    self._staging = torch.empty(need, device=self.device, dtype=torch.float32)
                staging = self._staging
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.ntx_gather_image_ex(self.handle, local_rgba.data_ptr(), shard.n, shard.run,
                                                    image.data_ptr() if image is not None else None,
                                                    staging.data_ptr() if staging is not None else None, dst,
                                                    _lib.GATHER_FORCE_EXCHANGE if force_exchange else 0,
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
    # no C-ABI communicator: the plan of ntx_gather_image (counts, block offsets, gather vs exact-count send/recv, un-shard map
    # -- all from the library, csrc/ntx_shard.h) through torch.distributed (gloo for the CPU plumbing tests; for GPU shards only as
    # the caller's explicit choice, e.g. bench.py when ntx_comm_create fails on a box -- it says so in its line)
    if world == 1:
        return local_rgba
    counts, offs, equal, direct = shard.plan()
    if local_rgba.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} must pass {counts[rank]} rows, got {tuple(local_rgba.shape)}")
    tail = tuple(local_rgba.shape[1:])
    local_rgba = local_rgba.contiguous()
    cap = counts[0]
    staging = None
    if rank == dst:
        staging = torch.zeros((world * cap,) + tail, dtype=local_rgba.dtype, device=local_rgba.device)
    if equal:                                                # one gather: block r lands at offs[r] = r * cap
        bufs = [staging[offs[r]: offs[r] + cap] for r in range(world)] if rank == dst else None
        dist.gather(local_rgba, bufs, dst=dst, group=group)
    elif rank == dst:                                        # exact counts: recv block r at offs[r]  (ncclRecv in ntx_gather_image)
        staging[offs[dst]: offs[dst] + counts[dst]] = local_rgba
        reqs = [dist.irecv(staging[offs[r]: offs[r] + counts[r]], src=r, group=group) for r in range(world) if r != dst and counts[r] > 0]
        for q in reqs:
            q.wait()
    elif counts[rank] > 0:
        dist.send(local_rgba, dst=dst, group=group)
    if rank != dst:
        return None
    if direct:                                               # the blocks in rank order are the image
        return staging[: shard.n]
    return staging[torch.as_tensor(shard.unshard_map(), device=local_rgba.device)]
