"""Ray generation (reference: network/ray_sampler.py)."""

from __future__ import annotations

import ctypes as C
from typing import Any, Tuple

import numpy as np


def _generate(pixel_range, height: int, width: int, focal: float, c2w, mode: int, b0, b1,
              near: float, far: float, device=None):
    """`pixel_range`: a pixel set of `pixel_sampler.Full` -- (first, count[, run_length, run_stride]) of the row-major grid -- or, as in
    the reference (ray_sampler.py:17, 25), any [n,2] tensor of (row, col) image-plane locations (`Independent`, `Proxy`, `Full.as_tensor`)."""
    import torch
    from . import _lib
    loc = None
    if hasattr(pixel_range, "shape"):                                # image_plane_loc tensor: tf.cast(image_plane_loc, tf.float32)
        loc = torch.as_tensor(pixel_range)
        if loc.dim() != 2 or loc.shape[1] != 2:
            raise ValueError(f"image_plane_loc must be [n,2], got {tuple(loc.shape)}")
        if device is None and loc.is_cuda:
            device = loc.device
        first, count, run, stride = 0, loc.shape[0], 1, 1
    else:
        first, count, run, stride = pixel_range if len(pixel_range) == 4 else (pixel_range[0], pixel_range[1], max(1, pixel_range[1]), max(1, pixel_range[1]))
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    rays_o = torch.empty((count, 3), device=dev, dtype=torch.float32)
    rays_d = torch.empty((count, 3), device=dev, dtype=torch.float32)
    t = torch.empty((count, 2), device=dev, dtype=torch.float32)
    cone = torch.empty((count, 1), device=dev, dtype=torch.float32)
    c2w_h = np.ascontiguousarray(np.asarray(c2w.detach().cpu() if hasattr(c2w, "detach") else c2w, dtype=np.float32))
    if c2w_h.shape != (4, 4):
        raise ValueError(f"c2w must be 4x4, got {c2w_h.shape}")
    if loc is not None:
        loc = loc.to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_generate_rays_at(c2w_h.ctypes.data_as(C.POINTER(C.c_float)), height, width, float(np.float32(focal)),
                                                     loc.data_ptr(), count, mode, _lib.f3(b0) if b0 is not None else None,
                                                     _lib.f3(b1) if b1 is not None else None, float(near), float(far),
                                                     rays_o.data_ptr(), rays_d.data_ptr(), t.data_ptr(), cone.data_ptr(),
                                                     torch.cuda.current_stream(dev).cuda_stream))
        return rays_o, rays_d, t, cone
    with torch.cuda.device(dev):
        _lib.check(_lib.lib.ntx_generate_rays_strided(c2w_h.ctypes.data_as(C.POINTER(C.c_float)), height, width,
                                                      float(np.float32(focal)), first, count, run, stride, mode,
                                                      _lib.f3(b0) if b0 is not None else None,
                                                      _lib.f3(b1) if b1 is not None else None, float(near), float(far),
                                                      rays_o.data_ptr(), rays_d.data_ptr(), t.data_ptr(), cone.data_ptr(),
                                                      torch.cuda.current_stream(dev).cuda_stream))
    return rays_o, rays_d, t, cone


class Frustum:
    """network.ray_sampler.Frustum (ray_sampler.py:6-21)."""

    def __init__(self, height: int, width: int, focal: float, near: float, far: float, **kwargs) -> None:
        self.height, self.width, self.focal, self.near, self.far = height, width, focal, near, far

    def __call__(self, image_plane_loc, c2w, device=None):
        return _generate(image_plane_loc, self.height, self.width, self.focal, c2w, 1, None, None, self.near,
                         self.far, device)


class Proxy:
    """network.ray_sampler.Proxy (ray_sampler.py:23-37): normalised rays_d and the proxy's t-range."""

    def __init__(self, height: int, width: int, focal: float, proxy: Any, **kwargs) -> None:
        self.height, self.width, self.focal, self.proxy = height, width, focal, proxy

    def __call__(self, image_plane_loc, c2w, device=None):
        """`image_plane_loc`: the (first_pixel, n_pixels[, run_length, run_stride]) pixel set produced by `pixel_sampler.Full`, or an
        [n,2] tensor of (row, col) as the reference's pixel samplers return it."""
        return _generate(image_plane_loc, self.height, self.width, self.focal, c2w, 0, self.proxy.b_0,
                         self.proxy.b_1, 0.0, 0.0, device)
