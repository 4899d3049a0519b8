"""Pixel sampling (reference: network/pixel_sampler.py).  `Full` is the render path's sampler; `Independent` and `Proxy` are the
training configs' (they feed the same `ray_sampler` interface, which takes their [n,2] pixel tensors through
`ntx_generate_rays_at`)."""

from __future__ import annotations

from typing import Any


class Full:
    """network.pixel_sampler.Full (pixel_sampler.py:8-15): every pixel, row-major.  Returns the
    pixel RANGE (first, count) rather than a materialised [HW,2] index tensor: ray generation
    derives (row, col) = (k // W, k % W) on the device.  With `shard=(ShardMap, rank)` it returns this rank's pixel
    set (first, count, run_length, run_stride) of the image (multi-GPU, nerf_tex_amd/dist.py).  `as_tensor()` gives the reference's
    own [HW,2] tensor."""

    def __init__(self, height: int, width: int, shard=None, **kwargs) -> None:
        self.height = height
        self.width = width
        self.shard = shard

    def __call__(self, **kwargs):
        if self.shard is not None:
            shard_map, rank = self.shard
            return shard_map.pixel_set(rank)
        return (0, self.height * self.width)

    def as_tensor(self, device=None):
        """tf.stack([range(HW) // W, range(HW) % W], -1) (pixel_sampler.py:15) as an int32 tensor."""
        import torch
        k = torch.arange(self.height * self.width, device=device, dtype=torch.int32)
        return torch.stack([k // self.width, k % self.width], -1)


class Independent:
    """network.pixel_sampler.Independent (pixel_sampler.py:17-29): n_samples pixels iid uniform over the image, as an int32 [n,2]
    tensor of (row, col).  Draws from torch's generator (TensorFlow's stream cannot be reproduced; the distribution is the same)."""

    def __init__(self, height: int, width: int, n_samples: int, **kwargs) -> None:
        self.height, self.width, self.n_samples = height, width, n_samples

    def __call__(self, device=None, generator=None, **kwargs):
        import torch
        i = torch.randint(0, self.height, (self.n_samples,), device=device, generator=generator, dtype=torch.int32)
        j = torch.randint(0, self.width, (self.n_samples,), device=device, generator=generator, dtype=torch.int32)
        return torch.stack([i, j], -1)


class Proxy:
    """network.pixel_sampler.Proxy (pixel_sampler.py:31-69): n_samples pixels among those whose rays hit the proxy, decided on a
    grid `downsample_factor` times coarser and upsampled by nearest neighbour (tf.image.resize(method='nearest') of TF 2.x:
    half-pixel centres, source index = min(floor((dst + 0.5) * src_size / dst_size), src_size - 1) in float32), then a random
    permutation of the hit pixels' indices and the first n_samples of it."""

    def __init__(self, height: int, width: int, n_samples: int, proxy: Any, focal: float, downsample_factor: int = 8, **kwargs) -> None:
        self.height, self.width, self.n_samples, self.proxy = height, width, n_samples, proxy
        self.downsample_factor = downsample_factor
        self.focal = focal // downsample_factor                       # (sic: floor division, pixel_sampler.py:40)
        self.height_down = height // downsample_factor
        self.width_down = width // downsample_factor

    def hit_mask(self, c2w, device=None):
        """the upsampled hit mask [H, W] (bool) of pixel_sampler.py:44-59"""
        import torch
        from .ray_sampler import Frustum
        hd, wd = self.height_down, self.width_down
        # rays of the coarse grid (ray_sampler.rays_from_camera: un-normalised directions), then the proxy on them (:47, :50)
        rays_o, rays_d, _, _ = Frustum(hd, wd, self.focal, 0.0, 0.0)((0, hd * wd), c2w, device=device)
        t = self.proxy(rays_o, rays_d)
        hit = (t[:, 0] != float("inf")).reshape(hd, wd)               # tf.where(t[:,0] == np.inf, 0, 1)
        if self.downsample_factor > 1:
            # resize_nearest_neighbor with half_pixel_centers=True (what tf.image.resize v2 passes), scales as float32 like the kernel
            def src(n_dst, n_src):
                scale = torch.tensor(n_src, dtype=torch.float32) / torch.tensor(n_dst, dtype=torch.float32)
                k = torch.floor((torch.arange(n_dst, dtype=torch.float32) + 0.5) * scale).to(torch.int64)
                return k.clamp_(max=n_src - 1).to(hit.device)
            hit = hit[src(self.height, hd)][:, src(self.width, wd)]
        return hit

    def hit_pixels(self, c2w, device=None):
        """`tf.where(hit_up == 1)` (pixel_sampler.py:62): the [m, 2] (row, col) of the pixels that count as hits, row-major.  It depends on
        the camera alone: `Dataset` keeps it per view, so a batch costs no mask, no `nonzero` and no wait for the step before it."""
        import torch
        return torch.nonzero(self.hit_mask(c2w, device))

    def __call__(self, c2w, device=None, generator=None, hits=None, rng=None):
        """`hits`: this camera's `hit_pixels` if the caller kept them.  `rng` (a numpy Generator): draw the n_samples positions on the host
        -- the head of a uniform random permutation is a uniform sample without replacement, which takes 256 draws instead of a device
        sort of the whole list (0.3 M hits of an 800 x 800 view: 0.2 ms of GPU time a view, serial with the training step) -- and send
        them through pinned memory without waiting for the device."""
        import torch
        idxs = self.hit_pixels(c2w, device) if hits is None else hits
        if rng is not None:
            m = int(idxs.shape[0])
            pick = torch.from_numpy(rng.choice(m, size=min(self.n_samples, m), replace=False))
            if idxs.is_cuda:
                pick = pick.pin_memory().to(idxs.device, non_blocking=True)
            return idxs[pick].to(torch.int32)
        perm = torch.randperm(idxs.shape[0], device=idxs.device, generator=generator)   # tf.random.shuffle
        return idxs[perm[: self.n_samples]].to(torch.int32)           # tf.gather_nd(idxs, range(n_samples)) of the shuffled list
