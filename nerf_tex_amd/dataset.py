"""Camera/parameter generation feeding the render path (reference: network/dataset.py, the parts
the render harness needs: `GenerateData` 198-229, `look_at` 231-238, `Dataset` 10-75).

The reference builds a tf.data pipeline; here a dataset is a plain Python iterable of batch dicts
with the same keys (`parameters`, `rays_o`, `rays_d`, `t`, `cone_scale`), rays generated on the GPU.
"""

from __future__ import annotations

from math import tan
from typing import Union

import numpy as np

from . import util
from .util import EasyDict


def look_at(pos, to=(0., 0., 0.), offset=(0., 0., 0.), eps=1e-6) -> np.ndarray:
    """dataset.look_at (dataset.py:231-238), float32 like the reference's tf ops."""
    f32 = np.float32
    pos = np.asarray(pos, dtype=f32); to = np.asarray(to, dtype=f32); offset = np.asarray(offset, dtype=f32)
    e = f32(eps)

    def normalize(v):
        return v / np.sqrt(np.sum(v * v))

    v_forward = normalize(pos - to + e)
    v_right = normalize(np.cross(np.asarray([0, 0, 1.], dtype=f32), v_forward) + e)
    v_up = normalize(np.cross(v_forward, v_right) + e)
    top = np.stack([v_right, v_up, v_forward, pos + offset], axis=1)
    return np.concatenate([top, np.asarray([[0, 0, 0, 1.]], dtype=f32)], axis=0).astype(f32)


def GenerateData(height: int = 256, width: int = 256, angle: float = .7,
                 pose_dist_config: EasyDict = None, radius: Union[float, dict] = 5., offset: list = [0., 0., 0.],
                 parameter_dist_config: EasyDict = None, dataset_size: int = -1, composite_bkgd: bool = False,
                 bkgd_color=[1, 1, 1.]):
    """dataset.GenerateData (dataset.py:198-229).  `pose_dist_config` / `parameter_dist_config` are
    instantiated through the plugin mechanism, so the reference's own TF-free `data.distribution.*`
    classes can be used as they are when its tree is importable."""
    if pose_dist_config is None or parameter_dist_config is None:
        raise ValueError("pose_dist_config and parameter_dist_config are required")
    pose_dist = util.instantiate(pose_dist_config)
    param_dist = util.instantiate(parameter_dist_config)
    rad = util.instantiate(radius) if isinstance(radius, dict) else (lambda: radius)
    n = max([dataset_size, pose_dist.sampler.n, param_dist.sampler.n])                  # dataset.py:212
    if n <= 0 or n > 256:
        raise NotImplementedError("generator-backed datasets (dataset.py:222-227) are not needed by the render configs")
    data = []
    for _ in range(n):                                                                  # dataset.py:217-219
        pose = look_at(np.asarray(pose_dist()) * rad(), offset=offset)
        data.append({"pose": pose, "parameters": np.asarray(param_dist(), dtype=np.float32)})
    return data, height, width, width / tan(angle / 2) / 2, composite_bkgd, bkgd_color


def FromViews(views: list, height: int = 256, width: int = 256, angle: float = .7, composite_bkgd: bool = False,
              bkgd_color=[1, 1, 1.]):
    """Data loader with explicit views (`[{'pose': c2w 4x4, 'parameters': [...]}, ...]`): same return tuple
    as GenerateData (dataset.py:229) without the pose/parameter distributions, which stay the reference's."""
    data = [{"pose": np.asarray(v["pose"], dtype=np.float32), "parameters": np.asarray(v["parameters"], dtype=np.float32)}
            for v in views]
    return data, height, width, width / tan(angle / 2) / 2, composite_bkgd, bkgd_color


class Dataset:
    """dataset.Dataset (dataset.py:10-75) for ray-only datasets: iterating yields one batch dict per
    `batchsize` views.  Attributes `height/width/focal/composite_bkgd/bkgd_color/n_samples/n_parameters`
    are set like the reference does by reflection (dataset.py:64-73)."""

    def __init__(self, data_loader_config, pixel_sampler_config, ray_sampler_config=None, proxy_config=None,
                 n_epochs: int = None, batchsize: int = 1, shuffle_buffer_size: int = 1, step=None, device=None):
        self.views, self.height, self.width, self.focal, self.composite_bkgd, self.bkgd_color = \
            util.instantiate(data_loader_config)
        proxy = util.instantiate(proxy_config)
        common = {"height": self.height, "width": self.width, "focal": self.focal, "proxy": proxy, "step": step}
        pixel_sampler_config.update(common)
        self.pixel_sampler = util.instantiate(pixel_sampler_config)
        if ray_sampler_config is None:       # dataset.py:24,35 allow it (image-only datasets); the render path needs rays
            raise NotImplementedError("a Dataset without ray_sampler_config yields no rays; the render path needs one")
        ray_sampler_config.update(common)
        self.ray_sampler = util.instantiate(ray_sampler_config)
        self.batchsize = batchsize
        self.device = device
        self.n_samples = self.height * self.width
        self.n_parameters = int(np.asarray(self.views[0]["parameters"]).shape[-1])

    def __len__(self):
        return (len(self.views) + self.batchsize - 1) // self.batchsize

    def __iter__(self):
        import torch
        for b in range(0, len(self.views), self.batchsize):
            chunk = self.views[b:b + self.batchsize]
            rays = [self.ray_sampler(image_plane_loc=self.pixel_sampler(c2w=v["pose"]), c2w=v["pose"], device=self.device)
                    for v in chunk]
            dev = rays[0][0].device
            yield {"parameters": torch.as_tensor(np.stack([v["parameters"] for v in chunk]), dtype=torch.float32, device=dev),
                   "rays_o": torch.stack([r[0] for r in rays]), "rays_d": torch.stack([r[1] for r in rays]),
                   "t": torch.stack([r[2] for r in rays]), "cone_scale": torch.stack([r[3] for r in rays])}
