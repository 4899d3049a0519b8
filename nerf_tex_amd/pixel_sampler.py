"""Pixel sampling (reference: network/pixel_sampler.py).  Only `Full` is on the render path."""

from __future__ import annotations


class Full:
    """network.pixel_sampler.Full (pixel_sampler.py:8-15): every pixel, row-major.  Returns the
    pixel RANGE (first, count) rather than a materialised [HW,2] index tensor: ray generation
    derives (row, col) = (k // W, k % W) on the device."""

    def __init__(self, height: int, width: int, **kwargs) -> None:
        self.height = height
        self.width = width

    def __call__(self, **kwargs):
        return (0, self.height * self.width)
