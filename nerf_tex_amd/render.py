"""Render harness (reference: network/render.py `Render` + the render half of network/logger.py).

`Render` builds dataset -> model -> renderer exactly as render.py:16-25 does and then plays the
part of `Logger.render_images` (logger.py:88-137): loop over views, call the renderer, pack RGBA.
With a `target_path` every view is written under `media/test/` as the reference's Logger does (logger.py:88-144): the PNG of the
filtered, un-premultiplied uint8 image -- or, with `logger_config.write_exr`, the premultiplied float32 image as an OpenEXR file
(`write_image`; nerf_tex_amd/exr.py) --, and beside it the raw premultiplied float32 RGBA as .npy.
"""

from __future__ import annotations

import os
from typing import List, Optional

from . import util


def render_image(renderer, dataset, data: dict):
    """logger.Logger.render_image up to the packing (logger.py:121-126): premultiplied RGBA [B, H, W, 4]."""
    import torch
    pred = renderer(**data, composite_bkgd=dataset.composite_bkgd, bkgd_color=dataset.bkgd_color, training=False)
    img = torch.cat([pred["color_pred"].reshape(-1, 3), pred["alpha_pred"].reshape(-1, 1)], -1)
    return img.reshape(-1, dataset.height, dataset.width, 4)


def image_epilogue(rgba, downsampling_factor: int = 1, write_exr: bool = False, uint8: bool = False):
    """The rest of logger.Logger.render_image / write_image (logger.py:128-144) on one [H,W,4] premultiplied
    image: filtered downsample (util/interpolate.py:78-82), un-premultiply unless EXR, optional uint8
    (`ntx_image_epilogue`).  Returns the float32 image, or (float32, uint8) when `uint8`."""
    import torch
    from . import _lib
    rgba = rgba.contiguous().float()
    h, w = rgba.shape[0], rgba.shape[1]
    f = int(downsampling_factor)
    oh, ow = -(-h // f), -(-w // f)
    out = torch.empty((oh, ow, 4), device=rgba.device, dtype=torch.float32)
    u8 = torch.empty((oh, ow, 4), device=rgba.device, dtype=torch.uint8) if uint8 else None
    with torch.cuda.device(rgba.device):
        _lib.check(_lib.lib.ntx_image_epilogue(rgba.data_ptr(), h, w, f, 0 if write_exr else 1, out.data_ptr(),
                                               u8.data_ptr() if uint8 else None,
                                               torch.cuda.current_stream(rgba.device).cuda_stream))
    return (out, u8) if uint8 else out


def Render(target_path: Optional[str], test_dataset_config, model_config, renderer_config, logger_config=None,
           source_path: str = None, override: bool = True, weights=None, weights_order: str = None, return_imgs: bool = True,
           **kwargs) -> List:
    """network.render.Render (render.py:6-29).  `weights`: `keras_model.get_weights()` as it is (a list), or the same flattened
    -- and then `weights_order="keras_get_weights"` must say so: the flat blob of ABI v1 had the alpha head after the trunk, v2+ has
    it last (Keras' order), both have the same size, and a blob in the other order renders garbage without any error."""
    test_dataset = util.instantiate(test_dataset_config)
    model_config.setdefault("n_parameters", test_dataset.n_parameters)                  # render.py:20
    model = util.instantiate(model_config)
    if weights is not None:                      # explicit weights win: `keras_model.get_weights()` as it is (a list: every
        m0 = next(iter(model.values()))          # layer's shape is checked), or the same flattened into one blob
        if isinstance(weights, (list, tuple)):
            m0.set_weights(weights)
        elif weights_order != "keras_get_weights":
            raise ValueError("Render(weights=<flat array>) needs weights_order='keras_get_weights': the blob is np.concatenate([w.ravel() for w in "
                             "keras_model.get_weights()]) -- trunk 0-7, feature, colour layers, color, alpha head LAST (ABI v1 had the alpha head after "
                             "the trunk; same size, so the order cannot be checked).  Pass the get_weights() list itself to have every layer's shape checked.")
        else:
            m0.set_blob(weights)
    else:                                        # logger.py:30-39: restore the newest ckpt-* under <source>/checkpoints
        ckpt_dir = os.path.join(source_path if source_path is not None else (target_path or ""), "checkpoints")
        if os.path.isdir(ckpt_dir) and any(f.endswith(".index") for f in os.listdir(ckpt_dir)):
            from .checkpoint import load_checkpoint
            for name, m in model.items():
                print("Restored model from {}.".format(load_checkpoint(m, ckpt_dir, root=name)))
    renderer_config.update(model)                                                      # render.py:24
    renderer = util.instantiate(renderer_config)
    imgs = []
    # One process per GPU (torch.distributed initialised, world > 1): rays shard freely (renderer.py:72-73 treats chunks of rays independently), so
    # every rank renders ITS pixels of each view -- a contiguous band of the row-major order, generated on its own device -- and ONE gather brings
    # the RGBA to rank 0 (`ncclGather` through the C ABI over the `nccl` backend, torch.distributed through host memory over `gloo`), which
    # writes the files and returns the images; the other ranks return [].  Jitter and noise are keyed by the pixel, so the image is the one-GPU image.
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    comm = shard = None
    if world > 1:
        from .dist import Comm, ShardMap
        from .pixel_sampler import Full
        if not isinstance(test_dataset.pixel_sampler, Full):
            raise NotImplementedError("a sharded Render needs pixel_sampler.Full")
        shard = ShardMap(test_dataset.height * test_dataset.width, world)
        test_dataset.pixel_sampler.shard = (shard, rank)
        if dist.get_backend() == "nccl":
            import torch
            comm = Comm(torch.cuda.current_device())
    for i, data in enumerate(test_dataset):
        if world > 1:
            img = _render_image_sharded(renderer, test_dataset, dict(data, seed=data.get("seed", 7919 * (i + 1))), shard, rank, comm)   # one jitter / noise stream a view, whatever the rank
            renderer.raise_if_nonfinite()
            if img is None:
                continue
        else:
            img = render_image(renderer, test_dataset, data)
            renderer.raise_if_nonfinite()
        if target_path is not None:
            import numpy as np
            os.makedirs(os.path.join(target_path, "media", "test"), exist_ok=True)
            np.save(os.path.join(target_path, "media", "test", util_format(i, len(test_dataset))), img.cpu().numpy())
            write_images(os.path.join(target_path, "media", "test"), img, i, len(test_dataset), int((logger_config or {}).get("downsampling_factor", 1)),
                         bool((logger_config or {}).get("write_exr", False)))
        if return_imgs:
            imgs.append(img)
    if comm is not None:
        comm.close()
    return imgs


def _render_image_sharded(renderer, dataset, data: dict, shard, rank: int, comm):
    """This rank's pixels of every view of the batch, gathered to rank 0: [B, H, W, 4] there, None elsewhere."""
    import torch
    from .dist import gather_image
    extra = {}
    if hasattr(renderer, "instancer") and data["t"].shape[0] == 1 and shard.contiguous:
        # InstanceRenderer keys its instancer's draws (the marching offset of a ray) by the ray's place among the image's proxy hits
        # (renderer.py:58-73): this rank's first hit comes after the hits of the bands before it -- one small all_gather a view
        import torch.distributed as dist
        on_device = dist.get_backend() == "nccl"
        mine = (data["t"][0, :, 0] != float("inf")).sum().reshape(1).to(torch.int64)
        mine = mine if on_device else mine.cpu()
        counts = [torch.empty_like(mine) for _ in range(shard.world)]
        dist.all_gather(counts, mine)
        extra["hit_base"] = int(sum(int(c.item()) for c in counts[:rank]))
    pred = renderer(**data, composite_bkgd=dataset.composite_bkgd, bkgd_color=dataset.bkgd_color, training=False, ray_index=shard.ray_index(rank), **extra)
    out = []
    for b in range(pred["color_pred"].shape[0]):
        local = torch.cat([pred["color_pred"][b].reshape(-1, 3), pred["alpha_pred"][b].reshape(-1, 1)], -1)
        if comm is not None:
            full = gather_image(local, shard, comm=comm)
        else:                                                        # gloo: the same plan through host memory
            full = gather_image(local.cpu(), shard)
            full = full.to(local.device) if full is not None else None
        out.append(full)
    return None if out[0] is None else torch.stack(out).reshape(-1, dataset.height, dataset.width, 4)


def write_images(directory: str, rgba, first_idx: int, max_idx: int, downsampling_factor: int = 1, write_exr: bool = False) -> List[str]:
    """Logger.render_image's post-processing and write_image (logger.py:128-144) for a batch of premultiplied RGBA images [B, H, W, 4]:
    filtered downsample, then either un-premultiply, uint8 (`ntx_image_epilogue`) and one PNG each, or with `write_exr` the float32 image
    as it is (premultiplied) in one OpenEXR file each (`nerf_tex_amd/exr.py`); named like `util.format_name('', idx, max_idx, '.png' / '.exr')`."""
    from . import exr, png
    os.makedirs(directory, exist_ok=True)
    out = []
    for k in range(rgba.shape[0]):
        stem = os.path.join(directory, util_format(first_idx + k, max_idx)[:-4])
        if write_exr:
            exr.write_exr(stem + ".exr", image_epilogue(rgba[k], downsampling_factor, write_exr=True).cpu().numpy())
            out.append(stem + ".exr")
        else:
            _, u8 = image_epilogue(rgba[k], downsampling_factor, write_exr=False, uint8=True)
            png.write_png(stem + ".png", u8.cpu().numpy())
            out.append(stem + ".png")
    return out


def util_format(idx: int, max_idx: int) -> str:
    import math
    n_chars = max(1, math.ceil(math.log10(max_idx + 1)))                               # util.py:56-62
    return ("{:0" + str(n_chars) + "d}").format(idx) + ".npy"
