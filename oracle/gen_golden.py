#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only
(`python oracle/gen_golden.py`): part (1) imports the reference's TF-free modules from /root/reference.

  (1) cameras_<family>.json   -- camera positions, radii and material parameters exactly as the
      reference's own `configs/*.py` + `data/distribution.py` + `data/sampler.py` produce them
      (np.random.seed(config['seed']) like main.py:30-31).  This is the only part of the path whose
      reference implementation can run here (everything else imports TensorFlow), so it is the only
      part pinned on outputs of the reference itself.  The c2w matrices are added by the oracle's
      restatement of `look_at` (dataset.py:231-238 imports TF) and flagged as such.
  (2) golden_<family>.npz     -- seeded inputs + float64 oracle outputs of every stage for
      64 camera rays x 32 samples (weights are regenerated from the seed; their sha256 is stored).
  (3) golden_edge.npz         -- composite edge cases.
  (4) golden_plumbing.npz     -- BASELINE configs[0]: carpet 200x200x32 image, float64 oracle, stored
      as float32 RGBA, plus the float32 oracle's image and its rel-Linf distance from the float64 one, plus the image of
      the float64 network on the float32 sample points (`rgba_net64`; nerftex_oracle.render_rays: points_dtype).
  (5) renderer_configs.json  -- the `renderer_config` blocks (with their `instancer_config`) of the reference's four shipped render
      configs, as the reference's own `configs/*.py` evaluate (data: keywords, numbers, file names), so that tests can hand them to
      this package verbatim.
Parts (2)-(4) are outputs of THIS repo's oracle (parity unpinned, see nerftex_oracle.py): they pin the
oracle against drift and give the GPU tests fixed vectors, they do not pin it to TensorFlow."""

from __future__ import annotations

import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import nerftex_oracle as orc          # noqa: E402
from nerf_tex_amd import synthetic                # noqa: E402  (seeded weights: shared by tests and bench)


def renderer_configs():
    """Part (5): the shipped render configs' renderer blocks, from the reference's own config modules."""
    sys.path.insert(0, REF)
    out = {}
    for fam in ("carpet", "grass", "grass_filtered", "plush"):
        cfg = importlib.import_module(f"configs.config_{fam}_render").config
        out[fam] = {"source": f"configs/config_{fam}_render.py", "renderer_config": cfg["renderer_config"],
                    "model_n_parameters": cfg["model_config"]["n_parameters"]}
    with open(os.path.join(OUT, "renderer_configs.json"), "w") as f:
        json.dump(out, f, indent=1)


def train_configs():
    """Part (6): what the shipped TRAINING configs hand the step of train.py:49-67 -- model, loss, learning-rate schedule, renderer block and
    the batch shape (images per batch x rays per image), and the data and logger blocks `Train` is called with (train.py:7-17) -- from the
    reference's own config modules."""
    sys.path.insert(0, REF)
    out = {}
    for fam in ("carpet", "fur", "grass", "grass_filtered", "plush"):
        cfg = importlib.import_module(f"configs.config_{fam}_train").config
        out[fam] = {"source": f"configs/config_{fam}_train.py", "model_config": cfg["model_config"], "loss_config": cfg["loss_config"],
                    "lrate": cfg["lrate"], "lrate_decay": cfg["lrate_decay"], "renderer_config": cfg["renderer_config"],
                    "batchsize": cfg["train_dataset_config"]["batchsize"], "rays_per_image": cfg["train_dataset_config"]["pixel_sampler_config"]["n_samples"],
                    "train_dataset_config": cfg["train_dataset_config"], "val_dataset_config": cfg["val_dataset_config"], "n_iters": cfg["n_iters"],
                    "logger_config": cfg["logger_config"]}
        # what the reference's own data/distribution.py + data/sampler.py make of the validation block (dataset.py:201-219), run here
        from util import util as ref_util
        np.random.seed(cfg["seed"])
        dl = ref_util.EasyDict(cfg["val_dataset_config"]["data_loader_config"])
        pose_dist, param_dist = ref_util.instantiate(dl.pose_dist_config), ref_util.instantiate(dl.parameter_dist_config)
        radius = dl.get("radius", 5.)                                                      # dataset.py:198 default
        rad = ref_util.instantiate(radius) if isinstance(radius, dict) else (lambda r=radius: r)
        n = max([dl.get("dataset_size", -1), pose_dist.sampler.n, param_dist.sampler.n])
        out[fam]["seed"] = cfg["seed"]
        out[fam]["val_views_reference"] = {"n": n, "views": [{"pose_dist_sample": np.asarray(pose_dist(), np.float64).tolist(), "radius": float(np.asarray(rad()).reshape(-1)[0]),
                                                                "parameters": np.asarray(param_dist(), np.float64).tolist()} for _ in range(min(n, 8))]}
    with open(os.path.join(OUT, "train_configs.json"), "w") as f:
        json.dump(out, f, indent=1)


def distributions():
    """Part (7): the reference's data/distribution.py + data/sampler.py (TensorFlow-free, run here) on blocks that reach what the shipped
    configs do not: random points, boxes, ranges, hemispheres, grids in several dimensions.  tests/test_data.py asks
    nerf_tex_amd/distributions.py for the same sequences."""
    sys.path.insert(0, REF)
    from util import util as ref_util
    S, D = "data.sampler.", "data.distribution."
    cases = [
        {"module": D + "Sphere"},
        {"module": D + "Sphere", "u_range": [0.1, 0.4], "v_range": [0.25, 0.5]},
        {"module": D + "Hemisphere", "axis": 0, "sampler_config": {"module": S + "Grid", "d": 2, "n": 9, "sample_center": True}},
        {"module": D + "Hemisphere", "axis": 1, "sampler_config": {"module": S + "Grid", "d": 2, "n": 7}},
        {"module": D + "Hemisphere", "axis": 2},
        {"module": D + "AABB"},
        {"module": D + "AABB", "b_0": [-1.5, 0.25, 2.0], "b_1": [1.5, 0.75, 3.0]},
        {"module": D + "AABB", "sampler_config": {"module": S + "Grid", "d": 3, "n": 20}, "b_0": [-2.0, -2.0, 0.5], "b_1": [1.0, 2.0, 4.0]},
        {"module": D + "Range", "n": 5, "b_0": [0.0, 1.0], "b_1": [1.0, 3.0]},
        {"module": D + "Range", "n": 3},
        {"module": D + "Constant", "constants": [[1, 2], [3, 4], [5, 6]]},
        {"module": D + "Concat", "distribution_config_0": {"module": D + "Constant", "constants": [[0.5], [0.75]]},
         "distribution_config_1": {"module": D + "Range", "n": 4, "b_0": [0.0, 10.0], "b_1": [1.0, 20.0]}},
        {"module": D + "Concat", "distribution_config_0": {"module": D + "Constant", "constants": [[7]]}, "distribution_config_1": {"module": D + "AABB"}},
        {"module": D + "Sphere", "sampler_config": {"module": S + "Concat", "sampler_config_0": {"module": S + "Constant", "c": [0.3]},
                                                   "sampler_config_1": {"module": S + "Grid", "sample_center": True}, "n": 6}},
        {"module": D + "AABB", "sampler_config": {"module": S + "Concat", "sampler_config_0": {"module": S + "Constant", "d": 2, "c": 0.25},
                                                 "sampler_config_1": {"module": S + "Independent", "d": 1}, "n": 4, "idx": 2}},
    ]
    out = []
    for k, block in enumerate(cases):
        np.random.seed(100 + k)
        dist = ref_util.instantiate(ref_util.EasyDict(json.loads(json.dumps(block))))
        samples = [np.asarray(dist(), np.float64).tolist() for _ in range(7)]
        out.append({"config": block, "seed": 100 + k, "n": int(dist.sampler.n), "idx_after": int(dist.sampler.idx), "done_after": bool(dist.sampler.done()),
                    "samples": samples})
    names = [[idx, mx, ref_util.format_name("", idx, mx, ".png")] for idx, mx in ((0, 0), (0, 1), (3, 9), (3, 10), (7, 99), (7, 100), (12, 256), (250000, 500000), (5, 1000000))]
    with open(os.path.join(OUT, "distributions.json"), "w") as f:
        json.dump({"source": "/root/reference/data/distribution.py + data/sampler.py + util/util.py format_name (reference code, run here by oracle/gen_golden.py)", "cases": out,
                   "format_name": names}, f, indent=1)
    print("distributions", len(out), "cases")


def cameras():
    """Part (1): run the reference's TF-free pose / parameter generators."""
    sys.path.insert(0, REF)
    from util import util as ref_util                  # /root/reference/util/util.py
    configs = {"carpet": "configs.config_carpet_render", "grass": "configs.config_grass_render",
               "grass_filtered": "configs.config_grass_filtered_render", "plush": "configs.config_plush_render"}
    for fam, modname in configs.items():
        cfg = importlib.import_module(modname).config
        np.random.seed(cfg["seed"])                                                      # main.py:30
        dl = ref_util.EasyDict(cfg["test_dataset_config"]["data_loader_config"])
        pose_dist = ref_util.instantiate(dl.pose_dist_config)                            # dataset.py:201
        param_dist = ref_util.instantiate(dl.parameter_dist_config)                      # dataset.py:203
        radius = dl.radius
        rad = ref_util.instantiate(radius) if isinstance(radius, dict) else (lambda: radius)
        n = max([dl.get("dataset_size", -1), pose_dist.sampler.n, param_dist.sampler.n])  # dataset.py:212
        n = min(n, 8)
        views = []
        for _ in range(n):                                                               # dataset.py:217-219
            p = np.asarray(pose_dist(), dtype=np.float64)
            r = float(np.asarray(rad()).reshape(-1)[0])
            prm = np.asarray(param_dist(), dtype=np.float64)
            views.append({"pose_dist_sample": p.tolist(), "radius": r, "parameters": prm.tolist(),
                          "c2w_oracle_look_at_f32": orc.look_at(p * r, offset=dl.get("offset", [0., 0., 0.]), dtype=np.float32).tolist()})
        proxy = cfg["test_dataset_config"]["proxy_config"]
        doc = {"source": f"/root/reference/{modname.replace('.', '/')}.py via data/distribution.py + data/sampler.py (reference code, run here)",
               "seed": cfg["seed"], "height": dl.height, "width": dl.width, "angle": dl.angle,
               "focal": orc.focal_from_angle(dl.width, dl.angle), "b_0": proxy["b_0"], "b_1": proxy["b_1"],
               "n_parameters": cfg["model_config"]["n_parameters"], "views": views,
               "data_loader_config": cfg["test_dataset_config"]["data_loader_config"],      # the block itself: tests/test_data.py runs the package's own generators on it
               "note": "c2w_oracle_look_at_f32 comes from the oracle's restatement of dataset.look_at, not from the reference"}
        with open(os.path.join(OUT, f"cameras_{fam}.json"), "w") as f:
            json.dump(doc, f, indent=1)
        print("cameras", fam, n, "views; first pose", views[0]["pose_dist_sample"])


def small(family):
    """Part (2)."""
    fam = synthetic.FAMILIES[family]
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
    blob = synthetic.synthetic_weights(orc.layer_table(spec), seed=0, dense_media=True)
    w = orc.split_blob(spec, blob)
    H, W, S = 8, 8, 32
    c2w = orc.look_at(fam["cam"], dtype=np.float32)
    focal = orc.focal_from_angle(W, fam["angle"] * 2.5)       # widened so the 8x8 grid has misses
    ro, rd, t, cone = orc.proxy_rays(orc.full_pixels(H, W), H, W, focal, c2w, fam["b_0"], fam["b_1"], np.float32)
    params = np.asarray([fam["params"]], np.float32)
    hit = np.isfinite(t[:, 0])
    aux = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, hit.sum(), 0), cone[hit], S, False,
                          (1., 1., 1.), fam["blur_idx"], dtype=np.float64, return_aux=True)
    full = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.),
                             fam["blur_idx"], dtype=np.float64)
    full_bk = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, True, (.25, .5, 1.),
                                fam["blur_idx"], dtype=np.float64)
    # model-only vectors: 96 random samples
    rng = np.random.default_rng(11)
    pos = rng.uniform(-1.5, 1.5, size=(96, 3)).astype(np.float32)
    dirs = rng.normal(size=(96, 3)); dirs = (dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(np.float32)
    prm = rng.uniform(0, 1, size=(96, spec.n_params)).astype(np.float32)
    col, alp, inter = orc.model_forward(w, spec, pos, dirs, prm, np.float64, return_intermediates=True)
    np.savez_compressed(
        os.path.join(OUT, f"golden_{family}.npz"),
        n_parameters=np.asarray(fam["n_parameters"]), weights_seed=0, weights_dense_media=True,
        weights_sha256=hashlib.sha256(blob.tobytes()).hexdigest(), height=H, width=W, n_samples=S,
        blur_idx=-1 if fam["blur_idx"] is None else fam["blur_idx"], c2w=c2w, focal=focal,
        b_0=np.asarray(fam["b_0"]), b_1=np.asarray(fam["b_1"]),
        rays_o=ro, rays_d=rd, t=t, cone_scale=cone, parameters=params, hit=hit,
        z_vals=aux["z_vals"], pts=aux["pts"], raw_color=aux["raw_color"], raw_alpha=aux["raw_alpha"],
        weights=aux["weights"], color_pred=full["color_pred"], alpha_pred=full["alpha_pred"],
        color_pred_bkgd=full_bk["color_pred"], alpha_pred_bkgd=full_bk["alpha_pred"], bkgd=np.asarray([.25, .5, 1.]),
        m_pos=pos, m_dirs=dirs, m_params=prm, m_color=col, m_alpha=alp, m_pos_map=inter["pos_map"][:8],
        m_dir_map=inter["dir_map"][:8], m_trunk0=inter["trunk0"][:8], m_trunk7=inter["trunk7"][:8],
        m_feature=inter["feature"][:8], m_color_half=inter["color_half"][:8])
    print("golden", family, "hits", int(hit.sum()), "/", hit.size, "max alpha", float(full["alpha_pred"].max()))


def edge():
    """Part (3): composite edge cases (renderer.py:170-213)."""
    rng = np.random.default_rng(7)
    n, S = 12, 16
    color = (rng.normal(size=(n, S, 3)) * 3).astype(np.float32)
    sigma = (rng.normal(size=(n, S)) * 10).astype(np.float32)
    z = np.sort(rng.uniform(1, 5, size=(n, S)), -1).astype(np.float32)
    rays_d = rng.normal(size=(n, 3)).astype(np.float32)
    sigma[0] = 1e9                      # every sample opaque: transmittance floors at 1e-10 per step (:198)
    sigma[1] = -3.0                     # relu -> nothing
    sigma[2] = 0.0
    sigma[3, :] = 0; sigma[3, -1] = 50  # only the last sample: its dist is the copy of the previous (:177)
    rays_d[4] *= 100.0                  # |d| >> 1 (:180)
    rays_d[5] *= 1e-3
    z[6] = 3.0                          # zero-length steps
    color[7] = 80.0; color[8] = -80.0   # sigmoid saturation / elu tail
    out = {}
    for exr in (False, True):
        for bk in (False, True):
            c, a, w, _ = orc.map_model_output(color, sigma, z, rays_d, bk, (.1, .6, .9), exr, None, np.float64)
            out[f"color_exr{int(exr)}_bk{int(bk)}"] = c
            out[f"alpha_exr{int(exr)}_bk{int(bk)}"] = a
            out[f"weights_exr{int(exr)}_bk{int(bk)}"] = w
    np.savez_compressed(os.path.join(OUT, "golden_edge.npz"), color=color, sigma=sigma, z=z, rays_d=rays_d,
                        bkgd=np.asarray([.1, .6, .9]), **out)
    print("edge ok")


def plumbing():
    """Part (4): BASELINE configs[0]."""
    fam = synthetic.FAMILIES["carpet"]
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=(1, 6))
    blob = synthetic.synthetic_weights(orc.layer_table(spec), seed=0, dense_media=True)
    w = orc.split_blob(spec, blob)
    H = W = 200; S = 32
    with open(os.path.join(OUT, "cameras_carpet.json")) as f:
        cam = json.load(f)
    v = cam["views"][0]
    c2w = np.asarray(v["c2w_oracle_look_at_f32"], np.float32)
    focal = orc.focal_from_angle(W, cam["angle"])
    ro, rd, t, cone = orc.proxy_rays(orc.full_pixels(H, W), H, W, focal, c2w, cam["b_0"], cam["b_1"], np.float32)
    params = np.asarray([v["parameters"]], np.float32)
    pred = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, dtype=np.float64)
    rgba = orc.render_image_rgba(pred, H, W)
    # the same image from the float32 restatement = what a float32 TF-CPU run would produce up to summation
    # order; its distance from the float64 truth is the float32 noise floor of this workload
    pred32 = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, dtype=np.float32)
    rgba32 = orc.render_image_rgba(pred32, H, W)
    floor = orc.rel_linf(rgba32, rgba)
    print("plumbing float32-vs-float64 floor: rel-Linf", floor)
    # the exact (float64) network and composite on the sample points a float32 run evaluates (render_rays: points_dtype): what
    # separates the float32 rounding of the INPUTS, which the floor above consists of, from the arithmetic of the network
    pred_net = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, dtype=np.float64, points_dtype=np.float32)
    rgba_net = orc.render_image_rgba(pred_net, H, W)
    print("plumbing: float32 restatement vs exact network on float32 points", orc.rel_linf(rgba32, rgba_net),
          "; input-rounding floor", orc.rel_linf(rgba_net, rgba))
    np.savez_compressed(os.path.join(OUT, "golden_plumbing.npz"), rgba=rgba.astype(np.float32), rgba_f32=rgba32.astype(np.float32),
                        rgba_net64=rgba_net.astype(np.float32), input_floor_rel_linf=orc.rel_linf(rgba_net, rgba),
                        f32_floor_rel_linf=floor, c2w=c2w, focal=focal,
                        parameters=params, b_0=np.asarray(cam["b_0"]), b_1=np.asarray(cam["b_1"]), height=H, width=W,
                        n_samples=S, weights_sha256=hashlib.sha256(blob.tobytes()).hexdigest(),
                        rgba_f64_sum=float(rgba.sum()), rgba_f64_max=float(rgba.max()))
    print("plumbing ok; hits", int(np.isfinite(t[:, 0]).sum()), "max", float(rgba.max()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["plumbing"]:                   # regenerate part (4) alone
        plumbing()
        sys.exit(0)
    if sys.argv[1:] == ["renderer_configs"]:           # part (5) alone
        renderer_configs()
        sys.exit(0)
    if sys.argv[1:] == ["distributions"]:              # part (7) alone
        distributions()
        sys.exit(0)
    if sys.argv[1:] == ["train_configs"]:              # part (6) alone
        train_configs()
        sys.exit(0)
    cameras()
    for fam in ("carpet", "grass", "fur", "grass_filtered"):
        small(fam)
    edge()
    plumbing()
    renderer_configs()
    train_configs()
    distributions()
