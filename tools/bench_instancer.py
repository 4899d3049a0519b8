#!/usr/bin/env python3
"""The patch instancer on a scene shaped like the shipped render configs (configs/config_carpet_render.py: patch box, patch_scale
0.09, step 0.002, 1024 samples, chunks of 16 384 rays, 'nearest'): achieved HBM rate of `ntx_instancer_model_input` with HIP events,
and the chunk end to end -- rays -> instancer -> ntx_render_instanced -- without leaving the GPU.  GPU box only:

    python tools/bench_instancer.py [--grid 48] [--rays 16384] [--samples 1024] [--steps 20] [--no-render]

The reference's meshes are LFS pointers, so the scene is synthetic: grid x grid patches on a gently waving sheet (the instancer
mesh, 2 (grid-1)^2 triangles), seen from the carpet config's first camera.  One JSON line per figure."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_tex_amd import _lib, synthetic                      # noqa: E402
from nerf_tex_amd.instancer import Instancer                  # noqa: E402
from nerf_tex_amd.model import ParamNerf                      # noqa: E402
from nerf_tex_amd.proxy import AABB                           # noqa: E402
from nerf_tex_amd.ray_sampler import Proxy                    # noqa: E402
from nerf_tex_amd.renderer import InstanceRenderer            # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md
B0, B1 = synthetic.PATCH_BOX                                  # config_carpet_render.py:83-84


sheet = synthetic.patch_sheet


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=48)
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--step-size", type=float, default=0.002)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--method", default="nearest")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--no-mesh", action="store_true")
    ap.add_argument("--sparse", action="store_true", help="NTX_OPT_INSTANCER_SPARSE: no defaults behind a ray's last step (what InstanceRenderer asks for)")
    ap.add_argument("--dense-render", action="store_true", help="the chunk end to end with the dense ten buffers (A/B of the sparse hand-off)")
    ap.add_argument("--scale-with-grid", action="store_true", help="patch_scale and step_size * 48 / grid: the same sheet covered by smaller "
                    "patches, the same number of patches and steps along a ray whatever the grid (a scene of 10^5 patches at --grid 316)")
    ap.add_argument("--textures", type=int, default=0, metavar="N", help="a one-channel image on parameter 0, looked up on the sheet: n_texture_samples = N, "
                    "min 8 (config_carpet_render.py:86-93: 256)")
    ap.add_argument("--shadows", type=int, default=0, metavar="N", help="cast_shadow_rays with n_shadow_samples = N, min 8 (config_grass_render.py:92-98: 128)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    scale = 0.09 * (48.0 / a.grid if a.scale_with_grid else 1.0)
    if a.scale_with_grid:
        a.step_size *= 48.0 / a.grid
    tr, v, f = sheet(a.grid, scale=scale)
    textures = ['', '', '', '', 'light']                                         # config_carpet_render.py:86 without the image texture
    tkw = {}
    if a.textures:                                                               # ... and with it: a smooth checkerboard of 256 x 256 texels
        yy, xx = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
        textures[0] = (127.5 + 127.5 * np.sin(xx * 0.2) * np.sin(yy * 0.2)).astype(np.uint8)
        uv = ((v[:, :2] + 1.5) / 3.0).astype(np.float32)
        tkw = dict(instancer_mesh=(v, f, uv), patch_scale=scale, min_texture_samples=8, n_texture_samples=a.textures)
    inst = Instancer(B0, B1, textures=textures, transformations=tr, instance_sampling_method=a.method,
                     mesh=None if a.no_mesh else (v, f), cast_shadow_rays=a.shadows > 0, min_shadow_samples=8, n_shadow_samples=max(a.shadows, 1), **tkw)
    fam = synthetic.FAMILIES["carpet"]
    side = int(np.sqrt(a.rays))
    assert side * side == a.rays, "--rays must be a square number (a side x side window of the 800 x 800 camera)"
    # the camera of config_carpet_render.py:23-37 (pose 0), a centred side x side window of its 800 x 800 grid: every ray meets the sheet
    from nerf_tex_amd.dataset import look_at
    c2w = look_at(np.asarray(fam["cam"], np.float32))
    focal = 800 / np.tan(fam["angle"] / 2) / 2
    r0 = (800 - side) // 2
    rows, cols = np.meshgrid(np.arange(r0, r0 + side), np.arange(r0, r0 + side), indexing="ij")
    loc = torch.as_tensor(np.stack([rows.ravel(), cols.ravel()], -1).astype(np.float32), device=dev)
    ro, rd, t, cone = Proxy(800, 800, focal, AABB([-1.7, -1.7, -.3], [1.7, 1.7, .4]))(loc, c2w, device=dev)
    P = inst.n_parameters
    par = np.asarray([fam["params"]], np.float32)
    if a.shadows:
        par[0, 4:7] = (0.9, 0.2, 0.08)                                            # a sun 5 degrees over the horizon: the waves of the sheet (slopes up to 9 degrees) shade each other
    params = torch.as_tensor(par, device=dev).repeat(a.rays, 1)
    S = a.samples
    out = inst.get_model_input(ro, rd, params, S, a.step_size, seed=1, sparse=a.sparse, fill=0.0 if a.sparse else None)   # (sparse: the rows it leaves alone read as 0 for the counts below)
    torch.cuda.synchronize()
    dists, hit = out[3], inst.last_hit
    in_patch = int((dists > 0).sum().item())
    emitted = int((out[2] > 0).sum().item())
    ms = timed(lambda: inst.get_model_input(ro, rd, params, S, a.step_size, seed=1, sparse=a.sparse), a.steps)
    out_bytes = a.rays * S * (12 + 12 + 4 + 4 + 4 + 4 + 4 * P) + a.rays * (12 + 4 + 1)
    if a.sparse:                                                                  # dists in full, the other six buffers where dists > 0
        out_bytes = a.rays * S * 4 + in_patch * (12 + 12 + 4 + 4 + 4 + 4 * P) + a.rays * (12 + 4 + 1)
    in_bytes = a.rays * (24 + 4 * P)
    gbps = (out_bytes + in_bytes) / (ms * 1e-3) / 1e9
    line = {"what": "ntx_instancer_model_input (hits + mesh + march kernels, HIP events around the call incl. the output torch.empty)" + (", NTX_OPT_INSTANCER_SPARSE" if a.sparse else ""),
            "scene": f"{a.grid}x{a.grid} = {a.grid ** 2} patches + {0 if a.no_mesh else f.shape[0]} triangles, method {a.method}"
                     + (f", shadow rays ({a.shadows} per unit length, min 8)" if a.shadows else "")
                     + (f", a parameter texture ({a.textures} lookups per unit length, min 8)" if a.textures else ""),
            "patch_scale": round(scale, 5), "shadowed_samples": int((out[9][..., 4:7] == torch.tensor([0., 0., -1.], device=dev)).all(-1).logical_and(out[2] > 0).sum().item()),
            "rays": a.rays, "n_pts": S, "step_size": a.step_size, "hit_rays": int(hit.sum().item()), "in_patch_samples": in_patch,
            "emitted_samples": emitted, "status": inst.status(), "ms": round(ms, 4),
            "rays_per_s": round(a.rays / (ms * 1e-3)), "in_patch_samples_per_s": round(in_patch / (ms * 1e-3)),
            "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                         "algorithmic_bytes": out_bytes + in_bytes,
                         "what": ("dists in full and the other six [N,S,...] buffers where dists > 0" if a.sparse else
                                  "every element of the ten output buffers written once ((3+3+1+1+1+1+P)*4 B per (ray, step))") + " + the rays read"}}
    print(json.dumps(line), flush=True)
    if a.no_render:
        return
    emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
    model = ParamNerf(emb(10), emb(4), emb(4), [1, 6])["model"]
    model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0, dense_media=True))
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=scale, step_size=a.step_size, render_chunk=a.rays,
                         density_scale=1.0, check_numerics=False)
    b = lambda x: x[None]
    call = lambda: r(b(ro), b(rd), b(t), parameters=params[:1], cone_scale=b(cone), instancer_seed=1, instancer_sparse=not a.dense_render)
    res = call(); torch.cuda.synchronize()
    ms_all = timed(call, max(3, a.steps // 2))
    alpha = res["alpha_pred"]
    print(json.dumps({"what": "InstanceRenderer.__call__ on one render chunk: instancer + ntx_render_instanced, nothing through the host" + (", dense hand-off" if a.dense_render else ", sparse hand-off"),
                      "rays": a.rays, "in_patch_samples": in_patch, "ms": round(ms_all, 3), "ms_instancer": round(ms, 4),
                      "instancer_share": round(ms / ms_all, 4), "in_patch_samples_per_s": round(in_patch / (ms_all * 1e-3)),
                      "mfma_frac_of_f32_peak": round(in_patch * 1361664 / (ms_all * 1e-3) / 157.3e12, 4),
                      "alpha_mean": round(float(alpha.mean().item()), 4)}), flush=True)
    # the reference's way round: the ten buffers come from the host (instancer.pyx:41-54 -> tf.constant)
    host = [x.cpu() for x in out]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _ = [x.to(dev) for x in host]
    torch.cuda.synchronize()
    up_ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"what": "uploading the ten buffers of one chunk from pageable host memory, which the reference does after its CPU instancer "
                              "(instancer.pyx:54) and this path never does", "bytes": int(sum(x.numel() * x.element_size() for x in host)),
                      "ms": round(up_ms, 2)}), flush=True)


if __name__ == "__main__":
    main()
