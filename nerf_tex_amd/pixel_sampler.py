"""Pixel sampling (reference: network/pixel_sampler.py).  Only `Full` is on the render path."""

from __future__ import annotations


class Full:
    """network.pixel_sampler.Full (pixel_sampler.py:8-15): every pixel, row-major.  Returns the
    pixel RANGE (first, count) rather than a materialised [HW,2] index tensor: ray generation
    derives (row, col) = (k // W, k % W) on the device.  With `shard=(ShardMap, rank)` it returns this rank's pixel
    set (first, count, run_length, run_stride) of the image (multi-GPU, nerf_tex_amd/dist.py)."""

    def __init__(self, height: int, width: int, shard=None, **kwargs) -> None:
        self.height = height
        self.width = width
        self.shard = shard

    def __call__(self, **kwargs):
        if self.shard is not None:
            shard_map, rank = self.shard
            return shard_map.pixel_set(rank)
        return (0, self.height * self.width)
