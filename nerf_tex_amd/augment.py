"""The blur augmentation that makes a filter-conditioned training set (reference: data/blur.py -- the tool behind `config_grass_filtered_train.py`'s
data, whose first geometry parameter is the blur a view was given: `blur_idx: 0`): every image of a NeRF (Blender layout) folder, possibly
several times over, blurred by a gaussian of a random width, the width written into the pose file as the FIRST driver parameter `Blur`.

    python -m nerf_tex_amd.augment <path_in> <path_out> [--subsets train] [--max_sigma 10] [--dataset_size_increase 2]

As there: sigmas = max_sigma * F^-1(u), u = `np.random.rand` under seed 0, F the exponential distribution on [0, 1] with rate p = 3 (blur.py:52-55
sets p = 3 whatever `--p` says; kept); PNGs are linearised (^2.2), premultiplied, blurred with zero padding (`skimage.filters.gaussian(...,
mode='constant', multichannel=True)` = `scipy.ndimage.gaussian_filter` with sigma (s, s, 0), truncate 4), un-premultiplied (+1e-5), brought back
(^1/2.2), clipped and rounded to uint8; EXR images go through `util.interpolate.filtered_downsample(img, 1, sigma)` (a normalised
int(6 sigma)-tap gaussian, SAME padding).  Unpinned (no skimage / TensorFlow here); scipy is the one dependency beyond numpy."""

from __future__ import annotations

import json
import math
import os
from typing import List, Sequence

import numpy as np


def sample_sigmas(n: int, max_sigma: float, p: float = 3.0) -> np.ndarray:
    """blur.py:45-56"""
    np.random.seed(0)
    x = np.random.rand(n)
    samples = x if -1e-4 < p < 1e-4 else -np.log(1 - x * (1 - np.exp(-p))) / p
    return samples * max_sigma


def blur_png(rgba_u8: np.ndarray, sigma: float) -> np.ndarray:
    """blur.py:66-78 on a uint8 RGBA image"""
    from scipy import ndimage
    img = np.asarray(rgba_u8, np.uint8).astype(np.float64) / 255                     # skimage.util.img_as_float
    img[:, :, :3] = img[:, :, :3] ** 2.2 * img[:, :, 3:]
    img = ndimage.gaussian_filter(img, sigma=(sigma, sigma, 0), mode="constant", cval=0.0, truncate=4.0)
    img[:, :, :3] = (img[:, :, :3] / (img[:, :, 3:] + 1e-5)) ** (1 / 2.2)
    return np.rint(np.clip(img, 0, 1) * 255).astype(np.uint8)                         # img_as_ubyte


def blur_exr(img: np.ndarray, sigma: float) -> np.ndarray:
    """interpolate.filtered_downsample(img, 1, sigma) (util/interpolate.py:68-82) in float32: depthwise, SAME padding (for an even number
    of taps TensorFlow pads one more behind than in front)."""
    a = np.asarray(img, np.float32)
    size = int(1 * sigma * 6)
    if size < 1:
        return a.copy()
    x = np.linspace(-(size - 1) / 2, (size - 1) / 2, size).astype(np.float32) + np.float32(.5 if size % 2 == 0 else 0)
    k1 = np.exp(np.float32(-.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    k2 = np.tensordot(k1, k1, axes=0)
    k2 = (k2 / k2.sum(dtype=np.float32)).astype(np.float32)
    lo, hi = (size - 1) // 2, size - 1 - (size - 1) // 2
    padded = np.pad(a, ((lo, hi), (lo, hi), (0, 0)))
    out = np.zeros_like(a)
    for i in range(size):
        for j in range(size):
            out += k2[i, j] * padded[i:i + a.shape[0], j:j + a.shape[1]]
    return out


def blur_folder(path_in: str, path_out: str, subsets: Sequence[str] = ("train",), max_sigma: float = 0.0, dataset_size_increase: int = 1) -> List[str]:
    """blur.py:23-120.  Returns the pose files written."""
    from . import exr, png
    os.makedirs(path_out)
    written = []
    for subset in subsets:
        imgs_path = os.path.join(path_in, subset)
        names = sorted(os.listdir(imgs_path))
        n, n_out = len(names), len(names) * dataset_size_increase
        with open(os.path.join(path_in, "transforms_" + subset + ".json")) as f:
            poses = json.load(f)
        os.makedirs(os.path.join(path_out, subset))
        sigma = sample_sigmas(n_out, max_sigma).tolist()
        fmt = "{:0" + str(math.ceil(np.log10(n_out))) + "d}"
        for idx, name, s in zip(range(n_out), names * dataset_size_increase, sigma):
            ext = os.path.splitext(name)[-1]
            out_name = name.split("_")[0] + "_" + fmt.format(idx) + ext
            if ext == ".png":
                png.write_png(os.path.join(path_out, subset, out_name), blur_png(png.with_channels(png.read_png(os.path.join(imgs_path, name)), 4), s))
            elif ext == ".exr":
                exr.write_exr(os.path.join(path_out, subset, out_name), blur_exr(exr.read_exr(os.path.join(imgs_path, name)), s))
            else:
                raise ValueError("Unknown filetype.")
        frames = []
        for i in range(n_out):
            fr = dict(poses["frames"][i % n])
            fr["file_path"] = fr["file_path"].split("_")[0] + "_" + fmt.format(i)
            fr["driver_parameters"] = {"Blur": sigma[i], **fr["driver_parameters"]}
            frames.append(fr)
        out_pose = os.path.join(path_out, "transforms_" + subset + ".json")
        with open(out_pose, "w+") as f:
            json.dump({"camera_angle_x": poses["camera_angle_x"], "frames": frames}, f, sort_keys=False, indent=4)
        written.append(out_pose)
    return written


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="Applies blur with random sigma to images and saves the amount to pose file.")
    ap.add_argument("path_in"); ap.add_argument("path_out")
    ap.add_argument("--subsets", nargs="+", default=["train"])
    ap.add_argument("--max_sigma", type=float, default=0)
    ap.add_argument("--dataset_size_increase", type=int, default=1)
    ap.add_argument("--p", type=float, default=3, help="(ignored, as in the reference: the exponent is 3)")
    a = ap.parse_args()
    print("wrote", ", ".join(blur_folder(a.path_in, a.path_out, a.subsets, a.max_sigma, a.dataset_size_increase)))
