"""A small training set for configs/example_carpet_train.py, made with nothing but this package: a "teacher" ParamNerf with seeded weights
(dense media) is rendered from cameras on the upper hemisphere; the views are written as a NeRF (Blender layout) folder -- uint8 RGBA PNGs as
the Logger writes them, `transforms_train.json` with the seven carpet parameters as `driver_parameters` -- and converted to TFRecord shards
(`nerf_tex_amd.tfrecord.convert_folder` = the reference's data/nerf2tfr.py).
    python tools/make_example_dataset.py [--out datasets/example_carpet] [--views 24] [--size 128]
    python -m nerf_tex_amd.main configs/example_carpet_train.py"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

BOX = {"module": "network.proxy.AABB", "b_0": [-1.5, -1.3, -.2], "b_1": [1.3, 1.3, 1.9]}       # config_carpet_train.py:28-31
PARAMS = [1.0, 1.0, 1.0, 0.1, 0.0, -0.707, 0.707]


def make(out: str, n_views: int = 24, size: int = 128, n_samples: int = 256, seed: int = 2, device: int = 0):
    from nerf_tex_amd import dataset as D, png, synthetic, tfrecord
    from nerf_tex_amd.layer import FourierFeatures
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.render import image_epilogue, render_image
    from nerf_tex_amd.renderer import Renderer
    dev = torch.device("cuda", device)
    teacher = ParamNerf(FourierFeatures(10), FourierFeatures(4), FourierFeatures(4), [1, 6])["model"]
    teacher.set_blob(synthetic.synthetic_weights(teacher.layer_table(), seed=seed, dense_media=True))
    rng = np.random.default_rng(0)
    cams = [np.asarray([np.cos(a) * np.sqrt(1 - z * z), np.sin(a) * np.sqrt(1 - z * z), z]) * 5 for a, z in zip(rng.uniform(0, 2 * np.pi, n_views), rng.uniform(0.3, 0.9, n_views))]
    views = [{"pose": D.look_at(c), "parameters": PARAMS} for c in cams]
    ds = D.Dataset({"module": "nerf_tex_amd.dataset.FromViews", "views": views, "height": size, "width": size, "angle": 0.63}, {"module": "network.pixel_sampler.Full"},
                   {"module": "network.ray_sampler.Proxy"}, dict(BOX), n_epochs=1, device=dev)
    renderer = Renderer(model=teacher, n_samples=n_samples, perturb=False)
    os.makedirs(os.path.join(out, "nerf", "train"))
    frames = []
    for k, data in enumerate(ds):
        u8 = image_epilogue(render_image(renderer, ds, data)[0], uint8=True)[1].cpu().numpy()
        png.write_png(os.path.join(out, "nerf", "train", f"r_{k:03d}.png"), u8)
        frames.append({"file_path": f"./train/r_{k:03d}", "transform_matrix": views[k]["pose"].tolist(), "driver_parameters": {f"p{i}": v for i, v in enumerate(PARAMS)}})
    with open(os.path.join(out, "nerf", "transforms_train.json"), "w") as f:
        json.dump({"camera_angle_x": 0.63, "frames": frames}, f)
    return tfrecord.convert_folder(os.path.join(out, "nerf"), os.path.join(out, "tfr"), imgs_per_shard=8), teacher


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="datasets/example_carpet")
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--size", type=int, default=128)
    a = ap.parse_args()
    files, _ = make(a.out, a.views, a.size)
    print("wrote", ", ".join(files))
