"""Entry point (reference: main.py): `python -m nerf_tex_amd.main <config.py>` loads a config file written for the reference,
points its hot-path modules at this package (`util.remap_reference_config`) and instantiates the top-level module.

What main.py does for a render config is kept: the config is a python file with a `config` dict (main.py:17-23), the random
seed is set from it (main.py:30-32; numpy only -- it drives the pose / parameter distributions, the initial weights and the
stratified jitter's seeds), the target folder is created and the config copied into it as `config_render.py` / `config_train.py`
(main.py:35-42).  Training configs (`network.train.Train`) run through `nerf_tex_amd.train.Train` with their own dataset, model, loss,
renderer and logger blocks.  The pose / parameter generators a config names (`data.distribution.*`, `data.sampler.*`) are this package's
(`nerf_tex_amd/distributions.py`); `--reference-root` puts a reference tree on the path for anything else a config may name.

    python -m nerf_tex_amd.main configs/example_carpet_render.py
    python -m nerf_tex_amd.main /path/to/nerf-tex/configs/config_carpet_render.py --reference-root /path/to/nerf-tex --volumetric
"""

from __future__ import annotations

import argparse
import importlib.util
import os
import shutil
import sys

import numpy as np

from . import util


def load_config(path: str) -> util.EasyDict:
    """main.py:17-23: the `config` dict of a python file (given as a path, with or without `.py`)."""
    path = path if path.endswith(".py") else path + ".py"
    spec = importlib.util.spec_from_file_location("_ntx_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return util.EasyDict(mod.config)


def prepare(config: dict, volumetric: bool = False) -> util.EasyDict:
    """Remap a reference config for this package.  `volumetric`: the shipped render configs name `InstanceRenderer`, which
    needs an instancer (the reference's on Embree, or nerf_tex_amd.instancer.Instancer with an exported transformation list); render
    the bare volume inside the proxy with `Renderer` instead."""
    cfg = util.remap_reference_config(config)
    if volumetric:
        rc = cfg.renderer_config
        if rc.module.endswith("InstanceRenderer"):
            rc.module = rc.module.replace("MipInstanceRenderer", "MipRenderer").replace("InstanceRenderer", "Renderer")
            for k in ("instancer_config", "step_size", "density_scale", "density_reweighting", "false_color"):
                rc.pop(k, None)
            rc.n_samples = min(int(rc.get("n_samples", 64)), 128)          # marching steps of the instancer -> samples in the proxy
    cfg.pop("seed", None); cfg.pop("override", None)
    return cfg


def main(argv=None) -> list:
    ap = argparse.ArgumentParser(description="Render as specified in a (reference) config file.")
    ap.add_argument("config", help="Path to config file.")
    ap.add_argument("--reference-root", default=None, help="directory of the reference tree (for its data.* / util.* modules)")
    ap.add_argument("--volumetric", action="store_true", help="replace InstanceRenderer by Renderer (no Embree instancer)")
    args = ap.parse_args(argv)
    if args.reference_root:
        sys.path.insert(0, os.path.abspath(args.reference_root))
    raw = load_config(args.config)
    if raw.get("seed") is not None:                                        # main.py:30-32
        np.random.seed(raw["seed"])
    # One process per GPU: started as `python -m torch.distributed.run --nproc-per-node N -m nerf_tex_amd.main <config>` this joins the
    # process group (RCCL: backend nccl), takes the GPU of its LOCAL_RANK, and `Render` / `Train` shard the work; rank 0
    # owns the target folder.  (NTX_MAIN_SHARE_GPU=1, development: every rank on GPU 0 over gloo.)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        share = os.environ.get("NTX_MAIN_SHARE_GPU") == "1"
        local = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group("gloo") if share else dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        if "train" in str(raw.get("module", "")):
            raw["device"] = local
    target = raw.get("target_path")
    if target and rank != 0:
        os.makedirs(target, exist_ok=True)
    elif target:                                                           # main.py:35-42
        os.makedirs(target, exist_ok=bool(raw.get("override", True)))
        dst = os.path.join(target, "config_train.py" if "train" in str(raw.get("module", "")) else "config_render.py")   # main.py:36-37
        src = args.config if args.config.endswith(".py") else args.config + ".py"
        if os.path.abspath(src) != os.path.abspath(dst):
            shutil.copy(src, dst)
        try:                                                               # main.py:44-47: the commit the run was made with, behind the copy
            import subprocess
            label = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).strip().decode("utf-8")
            with open(dst, "a") as f:
                f.write("\n# GIT COMMIT HASH: " + label)
        except (OSError, subprocess.CalledProcessError):
            pass                                                           # (not a git checkout: the reference stops here, this goes on)
    return util.instantiate(prepare(raw, args.volumetric))                 # main.py:50


if __name__ == "__main__":
    out = main()
    print(f"trained to step {out['step']}" if isinstance(out, dict) else f"rendered {len(out)} image(s)")
