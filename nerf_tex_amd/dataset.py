"""The data side of the render and training paths (reference: network/dataset.py): `Dataset` 10-75, the loaders `TFRecord` 77-129,
`FileFolder` 131-196, `GenerateData` 198-229, `look_at` 231-238.

The reference builds a tf.data pipeline; here a dataset is a plain Python iterable of batch dicts with the same keys (`parameters`,
`rays_o`, `rays_d`, `t`, `cone_scale`, and for image datasets `color`, `alpha`), rays generated on the GPU, the images resident in HBM as
the uint8 RGBA they were stored as (2.56 MB per 800 x 800 view: a whole training set fits many times over) and turned into premultiplied
float32 colours only at the pixels a batch samples.
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
    """dataset.GenerateData (dataset.py:198-229): `n = max(dataset_size, pose_dist.sampler.n, parameter_dist.sampler.n)` views, each a
    pose drawn from `pose_dist_config` times a radius (a number, or a distribution block) looking at the origin, and a parameter vector
    from `parameter_dist_config` -- drawn in that order, so random blocks consume numpy's global stream as the reference does.  The
    distribution blocks (`data.distribution.*`, `data.sampler.*`) resolve to `nerf_tex_amd.distributions` (pinned against the reference's
    own modules: tests/golden/cameras_*.json, distributions.json).  Two quirks kept: above 256 views the reference switches to a generator
    that forgets `offset` (:222-225); n <= 0 (every sampler endless, no `dataset_size`) is an empty dataset (`range(-1)`, :216)."""
    if pose_dist_config is None or parameter_dist_config is None:
        raise ValueError("pose_dist_config and parameter_dist_config are required (the reference's default names a module, data.dist, it does not have)")
    pose_dist = util.instantiate(pose_dist_config)
    param_dist = util.instantiate(parameter_dist_config)
    rad = util.instantiate(radius) if isinstance(radius, dict) else (lambda: radius)
    n = max([dataset_size, pose_dist.sampler.n, param_dist.sampler.n])                  # dataset.py:212
    data = []
    for _ in range(n):                                                                  # dataset.py:217-219 / :222-225
        pose = look_at(np.asarray(pose_dist()) * rad(), offset=offset if n <= 256 else (0., 0., 0.))
        data.append({"pose": pose, "parameters": np.asarray(param_dist(), dtype=np.float32)})
    return data, height, width, width / tan(angle / 2) / 2, composite_bkgd, bkgd_color


def FromViews(views: list, height: int = 256, width: int = 256, angle: float = .7, composite_bkgd: bool = False,
              bkgd_color=[1, 1, 1.]):
    """Data loader with explicit views (`[{'pose': c2w 4x4, 'parameters': [...][, 'rgba': uint8 or float32 [H, W, 4]]}, ...]`): same return
    tuple as the other loaders without the pose / parameter distributions or files; views with an `rgba` image make an image dataset
    (un-premultiplied, as a PNG holds it), whose height and width are the images'."""
    data = []
    for v in views:
        d = {"pose": np.asarray(v["pose"], dtype=np.float32), "parameters": np.asarray(v["parameters"], dtype=np.float32)}
        if v.get("rgba") is not None:
            d.update(rgba=v["rgba"], premultiplied=bool(v.get("premultiplied", False)))
            height, width = int(d["rgba"].shape[0]), int(d["rgba"].shape[1])
        data.append(d)
    return data, height, width, width / tan(angle / 2) / 2, composite_bkgd, bkgd_color


def _image_view(pose, parameters, rgba, premultiplied: bool) -> dict:
    return {"pose": np.asarray(pose, dtype=np.float32), "parameters": np.asarray(parameters, dtype=np.float32).reshape(-1), "rgba": rgba,
            "premultiplied": premultiplied}


def TFRecord(tfr_path: str, composite_bkgd: bool = False, bkgd_color=[1, 1, 1.], read_exr: bool = False, compression_type: str = None,
             verify: bool = True):
    """dataset.TFRecord (dataset.py:77-129): the views of one TFRecord file, or of every file of a directory (`os.listdir` order, as there),
    of `tf.train.Example`s with `image` (an encoded PNG; a serialized float32 [H, W, 4] tensor when `read_exr`), `pose` / `parameters`
    (serialized float32 tensors) and `angle`.  Height, width and angle are the first record's (:121-123); `read_exr` switches the background
    compositing off (:125-126).  Same return tuple as the other loaders.  The images stay as stored (uint8 RGBA; float32 for EXR tensors):
    `Dataset` does `convert_image_dtype`, the premultiplication and the compositing (:106-110) on the pixels it samples."""
    import os
    from . import png, tfrecord
    paths = [os.path.join(tfr_path, n) for n in os.listdir(tfr_path)] if os.path.isdir(tfr_path) else [tfr_path]
    views, angle = [], None
    for path in paths:
        for record in tfrecord.read_records(path, compression_type, verify):
            f = tfrecord.parse_example(record)
            for k in ("image", "pose", "angle", "parameters"):
                if k not in f or len(f[k]) != 1:
                    raise ValueError(f"{path}: feature {k!r} is missing or not a single value")
            if read_exr:
                img = tfrecord.parse_tensor(f["image"][0], np.float32)                       # tf.io.parse_tensor(features['image'], tf.float32)
                if img.ndim != 3 or img.shape[2] < 4:
                    raise ValueError(f"{path}: an EXR tensor [H, W, 4] is expected, got {img.shape}")
                img = np.ascontiguousarray(img[..., :4])
            else:
                img = png.with_channels(png.decode_png(f["image"][0], path), 4)             # tf.image.decode_image(..., channels=4)
            if angle is None:
                angle = float(f["angle"][0])
            views.append(_image_view(tfrecord.parse_tensor(f["pose"][0], np.float32), tfrecord.parse_tensor(f["parameters"][0], np.float32), img, read_exr))
    if not views:
        raise ValueError(f"{tfr_path}: no records")
    height, width = views[0]["rgba"].shape[:2]
    if read_exr:
        composite_bkgd = False
    return views, height, width, width / tan(angle / 2) / 2, composite_bkgd, bkgd_color


def load_poses(pose_path: str, idxs: list):
    """dataset.load_poses (dataset.py:174-196): `transform_matrix` and the VALUES of `driver_parameters` in the file's own order for the
    frames whose position is in `idxs`, and `camera_angle_x` -- NeRF's Blender `transforms_*.json`."""
    import json
    with open(pose_path) as f:
        d = json.load(f)
    frames = [p for i, p in enumerate(d["frames"]) if i in idxs]
    poses = [np.asarray(p["transform_matrix"], np.float32) for p in frames]
    parameters = [np.asarray(list(p["driver_parameters"].values()) if "driver_parameters" in p else [], np.float32) for p in frames]
    return poses, parameters, d["camera_angle_x"]


def load_imgs(imgs_path: str, idxs: list):
    """dataset.load_imgs (dataset.py:148-172): the `.png` files of a folder in sorted order whose position is in `idxs`, as uint8 RGBA.
    (`.jpg` files are listed there too: there is no JPEG decoder here, one is refused by name.  The reference's composite_bkgd branch
    multiplies the result of `list.append` by the colour (:165) and cannot run; what it means -- TFRecord's :107-108 -- is done in `Dataset`.)"""
    import os
    from . import png
    names = sorted(n for n in os.listdir(imgs_path) if n[-4:] in (".png", ".jpg"))
    out = []
    for name in [n for i, n in enumerate(names) if i in idxs]:
        if name.endswith(".jpg"):
            raise NotImplementedError(f"{name}: JPEG images are not decoded here (PNG only)")
        out.append(png.with_channels(png.read_png(os.path.join(imgs_path, name)), 4))
    return out


def FileFolder(imgs_path: str = None, poses_path: str = None, idxs: list = [], height: int = 256, width: int = 256, angle: float = .7,
               composite_bkgd: bool = False, bkgd_color=[1, 1, 1.]):
    """dataset.FileFolder (dataset.py:131-146): a folder of images and a pose file after NeRF's Blender dataset layout; either may be
    missing (poses only: a ray dataset of the file's cameras)."""
    poses = parameters = imgs = None
    if poses_path is not None:
        poses, parameters, angle = load_poses(poses_path, idxs)
    if imgs_path is not None:
        imgs = load_imgs(imgs_path, idxs)
        height, width = imgs[0].shape[:2]
    n = len(poses) if poses is not None else len(imgs) if imgs is not None else 0
    if poses is not None and imgs is not None and len(imgs) != n:
        raise ValueError(f"{len(imgs)} images for {n} poses")              # tf.data.Dataset.from_tensor_slices refuses it too
    views = []
    for k in range(n):
        v = {}
        if poses is not None:
            v.update(pose=poses[k], parameters=parameters[k])
        if imgs is not None:
            v.update(rgba=imgs[k], premultiplied=False)
        views.append(v)
    return views, height, width, width / tan(angle / 2) / 2, composite_bkgd, bkgd_color


class Dataset:
    """dataset.Dataset (dataset.py:10-75): loader, pixel sampler and ray sampler combined; iterating yields batch dicts of `batchsize` views
    -- `parameters` [B, P], `rays_o` / `rays_d` [B, R, 3], `t` [B, R, 2], `cone_scale` [B, R, 1] and, when the views carry images, `color`
    [B, R, 3] and `alpha` [B, R] gathered at the sampled pixels (`tf.gather_nd`, :49, :57) -- in the order of
    `map(...).shuffle(shuffle_buffer_size, reshuffle_each_iteration=True).repeat(n_epochs).batch(batchsize)` (:62): every epoch through a
    shuffle buffer of its own, the epochs end to end, batches cut across their seams, the last one short; `n_epochs=None` repeats for ever
    and `take(n)` is how the loop bounds it (train.py:60).  The pixels are sampled anew each time a view comes up.  Randomness (shuffle,
    pixel samplers) comes from a torch generator seeded with `seed` (TensorFlow's streams cannot be reproduced; `None`: torch's global one).
    Attributes `height / width / focal / composite_bkgd / bkgd_color / n_samples / n_parameters` as the reference sets them (:64-73)."""

    def __init__(self, data_loader_config, pixel_sampler_config, ray_sampler_config=None, proxy_config=None,
                 n_epochs: int = None, batchsize: int = 1, shuffle_buffer_size: int = 1, step=None, device=None, seed: int = None,
                 fused_batches: bool = True):
        self.views, self.height, self.width, self.focal, self.composite_bkgd, self.bkgd_color = \
            util.instantiate(data_loader_config)
        proxy = util.instantiate(proxy_config) if proxy_config is not None else None
        common = {"height": self.height, "width": self.width, "focal": self.focal, "proxy": proxy, "step": step}
        pixel_sampler_config.update(common)
        self.pixel_sampler = util.instantiate(pixel_sampler_config)
        self.has_images = len(self.views) > 0 and "rgba" in self.views[0]
        if ray_sampler_config is None and not self.has_images:
            raise ValueError("a Dataset without ray_sampler_config and without images yields nothing")
        self.ray_sampler = None
        if ray_sampler_config is not None:
            ray_sampler_config.update(common)
            self.ray_sampler = util.instantiate(ray_sampler_config)
        self.n_epochs = n_epochs
        self.batchsize = int(batchsize)
        self.shuffle_buffer_size = max(1, int(shuffle_buffer_size))
        self.device = device
        self.seed = seed
        self._generator = None
        self._host_rng = np.random.default_rng(seed if seed is None else [int(seed), 0])       # the order of the views
        self._pixel_rng = np.random.default_rng(seed if seed is None else [int(seed), 1])      # the pixels of a view
        self._resident, self._hits, self._params, self._bkgd = {}, {}, {}, None
        self._hits_host, self._stacked = {}, None
        self.fused_batches = bool(fused_batches)
        from .pixel_sampler import Full
        self._sampled = not isinstance(self.pixel_sampler, Full)
        self.n_samples = int(self.pixel_sampler.n_samples) if self._sampled else self.height * self.width
        self.n_parameters = int(np.asarray(self.views[0]["parameters"]).shape[-1]) if self.views and "parameters" in self.views[0] else 0

    def __len__(self):
        """batches of ONE epoch"""
        return (len(self.views) + self.batchsize - 1) // self.batchsize

    # ---- order: shuffle(buffer).repeat(n_epochs).batch(batchsize) ----
    def _order(self):
        epoch = 0
        while self.n_epochs is None or epoch < self.n_epochs:
            epoch += 1
            if self.shuffle_buffer_size <= 1:
                yield from range(len(self.views))
                continue
            buf, src = [], iter(range(len(self.views)))
            for k in src:                                         # a buffer of shuffle_buffer_size elements, one drawn at random as the next arrives
                if len(buf) < self.shuffle_buffer_size:
                    buf.append(k)
                    continue
                j = int(self._host_rng.integers(len(buf)))
                yield buf[j]
                buf[j] = k
            while buf:
                yield buf.pop(int(self._host_rng.integers(len(buf))))

    def _device(self):
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if self.device is None else torch.device(self.device)

    def _gen(self, dev):
        import torch
        if self.seed is None:
            return None
        if self._generator is None:
            self._generator = torch.Generator(device=dev)
            self._generator.manual_seed(int(self.seed))
        return self._generator

    def _image(self, k: int, dev):
        import torch
        if k not in self._resident:
            self._resident[k] = torch.as_tensor(self.views[k]["rgba"]).to(dev)
        return self._resident[k]

    def colors_at(self, k: int, loc, dev=None):
        """(color [n, 3], alpha [n]) of view k at the pixels `loc` ([n, 2] rows / columns, or a `Full` range): `tf.gather_nd(image, loc)` of the
        loaders' maps -- `convert_image_dtype` to float32 (x * (1 / 255) in float32), colour times alpha, and over `bkgd_color` if
        `composite_bkgd` (dataset.py:104-112) -- evaluated at the gathered pixels only."""
        import torch
        dev = self._device() if dev is None else dev
        img = self._image(k, dev)
        if hasattr(loc, "shape"):
            if loc.dtype.is_floating_point:
                raise NotImplementedError("float image-plane locations need util.interpolate_img, which the reference does not have (dataset.py:44, 52)")
            px = img[loc[:, 0].long(), loc[:, 1].long()]
        else:
            first, count = loc[0], loc[1]
            if len(loc) == 4 and loc[2] != loc[3]:
                raise NotImplementedError("strided pixel sets carry no images")
            px = img.reshape(-1, img.shape[-1])[first:first + count]
        if px.dtype == torch.uint8:
            px = px.to(torch.float32) * torch.tensor(1.0 / 255, dtype=torch.float32)       # tf.image.convert_image_dtype(uint8 -> float32)
        alpha = px[:, 3]
        if self.views[k]["premultiplied"]:                        # read_exr: img[..., :3] as it is (:100-101)
            return px[:, :3].contiguous(), alpha.contiguous()
        color = px[:, :3] * px[:, 3:]
        if self.composite_bkgd:
            if self._bkgd is None or self._bkgd.device != px.device:
                self._bkgd = torch.as_tensor(self.bkgd_color, dtype=torch.float32, device=px.device)
            color = color + (1 - px[:, 3:]) * self._bkgd
        return color, alpha.contiguous()

    def _element(self, k: int, dev):
        import torch
        v = self.views[k]
        if self._sampled:
            kw = {}
            if hasattr(self.pixel_sampler, "hit_pixels"):             # the proxy's hit pixels of this camera: found once
                if k not in self._hits:
                    self._hits[k] = self.pixel_sampler.hit_pixels(v["pose"], dev)
                kw.update(hits=self._hits[k], rng=self._pixel_rng)
            loc = self.pixel_sampler(c2w=v["pose"], device=dev, generator=self._gen(dev), **kw)
            if loc.shape[0] != self.n_samples:
                raise ValueError(f"view {k}: the pixel sampler found {loc.shape[0]} of {self.n_samples} pixels")     # tf.gather_nd out of range (pixel_sampler.py:69)
        else:
            loc = self.pixel_sampler(c2w=v["pose"])
        out = {}
        if "parameters" in v:
            if k not in self._params:
                self._params[k] = torch.as_tensor(v["parameters"], dtype=torch.float32, device=dev)
            out["parameters"] = self._params[k]
        if self.ray_sampler is not None:
            out["rays_o"], out["rays_d"], out["t"], out["cone_scale"] = self.ray_sampler(image_plane_loc=loc, c2w=v["pose"], device=dev)
        if "rgba" in v:
            out["color"], out["alpha"] = self.colors_at(k, loc, dev)
        return out

    # ---- a training batch in one piece --------------------------------------------------------------------------------------------
    def _can_fuse(self, dev) -> bool:
        """Proxy-sampled rays of uint8 images of one size: the pixel positions of ALL the batch's views are drawn on the host, go to the
        device in one pinned copy, and the colours come out of one gather over the stacked images -- a dozen kernels a batch instead of
        fifty (they run on the training step's stream, serial with it).  The batches are the per-view path's, bit for bit."""
        import torch
        if not (self.fused_batches and self._sampled and hasattr(self.pixel_sampler, "hit_pixels") and self.has_images and self.ray_sampler is not None):
            return False
        v0 = self.views[0]
        return dev.type == "cuda" and all("parameters" in v and "rgba" in v and not v["premultiplied"] and tuple(v["rgba"].shape) == tuple(v0["rgba"].shape)
                                          and str(v["rgba"].dtype).endswith("uint8") for v in self.views)

    def _fused_batch(self, ks, dev) -> dict:
        import torch
        H, W, R = self.height, self.width, self.n_samples
        if self._stacked is None:
            self._stacked = torch.stack([self._image(k, dev) for k in range(len(self.views))]).reshape(-1, 4)
            self._resident = {k: self._stacked.reshape(len(self.views), H, W, 4)[k] for k in range(len(self.views))}
        loc = np.empty((len(ks), R, 2), np.int64)
        for e, k in enumerate(ks):
            if k not in self._hits_host:
                if k not in self._hits:
                    self._hits[k] = self.pixel_sampler.hit_pixels(self.views[k]["pose"], dev)
                self._hits_host[k] = self._hits[k].cpu().numpy()
            hits = self._hits_host[k]
            if hits.shape[0] < R:
                raise ValueError(f"view {k}: the pixel sampler found {hits.shape[0]} of {R} pixels")
            loc[e] = hits[self._pixel_rng.choice(hits.shape[0], size=R, replace=False)]
        flat = ((np.asarray(ks, np.int64)[:, None] * H + loc[..., 0]) * W + loc[..., 1]).reshape(-1)
        locf = torch.from_numpy(loc.astype(np.float32)).pin_memory().to(dev, non_blocking=True)
        flat = torch.from_numpy(flat).pin_memory().to(dev, non_blocking=True)
        rays = [self.ray_sampler(image_plane_loc=locf[e], c2w=self.views[k]["pose"], device=dev) for e, k in enumerate(ks)]
        for k in ks:
            if k not in self._params:
                self._params[k] = torch.as_tensor(self.views[k]["parameters"], dtype=torch.float32, device=dev)
        px = self._stacked[flat].to(torch.float32) * torch.tensor(1.0 / 255, dtype=torch.float32)
        color = px[:, :3] * px[:, 3:]
        if self.composite_bkgd:
            if self._bkgd is None or self._bkgd.device != px.device:
                self._bkgd = torch.as_tensor(self.bkgd_color, dtype=torch.float32, device=px.device)
            color = color + (1 - px[:, 3:]) * self._bkgd
        n = len(ks)
        return {"parameters": torch.stack([self._params[k] for k in ks]), "rays_o": torch.stack([r[0] for r in rays]), "rays_d": torch.stack([r[1] for r in rays]),
                "t": torch.stack([r[2] for r in rays]), "cone_scale": torch.stack([r[3] for r in rays]), "color": color.reshape(n, R, 3),
                "alpha": px[:, 3].reshape(n, R).contiguous()}

    def __iter__(self):
        import torch
        dev = self._device()
        fuse = self._can_fuse(dev)
        batch = []
        for k in self._order():
            batch.append(k if fuse else self._element(k, dev))
            if len(batch) == self.batchsize:
                yield self._fused_batch(batch, dev) if fuse else {key: torch.stack([e[key] for e in batch]) for key in batch[0]}
                batch = []
        if batch:
            yield self._fused_batch(batch, dev) if fuse else {key: torch.stack([e[key] for e in batch]) for key in batch[0]}

    def take(self, n: int):
        """`tf.data.Dataset.take`: at most n batches (a fresh iteration)."""
        import itertools
        return itertools.islice(iter(self), max(0, int(n)))
