"""Proxy geometry (reference: network/proxy.py)."""

from __future__ import annotations


class AABB:
    """network.proxy.AABB (proxy.py:6-35).  Holds the box; the slab test itself runs inside
    `ntx_generate_rays` (mode 0), fused with ray generation."""

    def __init__(self, b_0: list, b_1: list):
        self.b_0 = [float(v) for v in b_0]
        self.b_1 = [float(v) for v in b_1]
