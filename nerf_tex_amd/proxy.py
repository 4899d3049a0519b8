"""Proxy geometry (reference: network/proxy.py)."""

from __future__ import annotations


class AABB:
    """network.proxy.AABB (proxy.py:6-35): the box, and the slab test on the caller's rays.  On the render path the test runs fused
    with ray generation (`ntx_generate_rays*`, mode 0: the same device function); `__call__` is the reference's own interface
    (`ray_sampler.Proxy`, `pixel_sampler.Proxy` call it on rays they made themselves) through `ntx_aabb_intersect`."""

    def __init__(self, b_0: list, b_1: list):
        self.b_0 = [float(v) for v in b_0]
        self.b_1 = [float(v) for v in b_1]

    def __call__(self, rays_o, rays_d):
        """rays_o, rays_d [n,3] float32 on the GPU -> t [n,2]; [inf, inf] where the ray misses the box (proxy.py:13-35; assumes, as
        the reference does, that the origin lies outside the box)."""
        import torch
        from . import _lib
        rays_o = rays_o.contiguous().float(); rays_d = rays_d.contiguous().float()
        if rays_o.shape != rays_d.shape or rays_o.dim() != 2 or rays_o.shape[1] != 3:
            raise ValueError(f"rays_o / rays_d must be [n,3], got {tuple(rays_o.shape)} / {tuple(rays_d.shape)}")
        if not rays_o.is_cuda:
            raise ValueError("AABB.__call__ runs on the GPU: pass CUDA(ROCm) tensors")
        t = torch.empty((rays_o.shape[0], 2), device=rays_o.device, dtype=torch.float32)
        with torch.cuda.device(rays_o.device):
            _lib.check(_lib.lib.ntx_aabb_intersect(rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], _lib.f3(self.b_0), _lib.f3(self.b_1),
                                                   t.data_ptr(), torch.cuda.current_stream(rays_o.device).cuda_stream))
        return t
