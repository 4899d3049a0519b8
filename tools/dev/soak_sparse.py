#!/usr/bin/env python3
"""GPU box: NTX_OPT_INSTANCER_SPARSE against the dense ten buffers on RANDOM scenes -- patch count, sampling rule, mesh or none, shadow rays,
step counts and sizes drawn per case; the sparse call's untouched rows hold NaN from before.  Per case: dists equal everywhere, the other six
[n,S,...] buffers equal wherever dists > 0 and still NaN behind, the per-ray outputs equal, and the image of InstanceRenderer (sparse by
default, NaN behind) bit-identical to the image of the dense buffers with finite numerics.
    python tools/dev/soak_sparse.py [--cases 120] [--seed 0]"""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.common import make_model                                               # noqa: E402
from tests.test_oracle_instancer import random_rays, random_scene                 # noqa: E402
from tests.test_gpu_instancer import gpu_instancer                                # noqa: E402
from nerf_tex_amd.renderer import InstanceRenderer                                # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=120); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
F = np.float32
model, mspec, wts = make_model((1, 6), dense_media=True)
dv = torch.device("cuda", 0)
t0, fails, live_frac = time.time(), [], []
for case in range(a.cases):
    seed = int(rng.integers(1 << 20))
    method = ["random", "nearest", "nearest_blend"][rng.integers(3)]
    mesh, shadows = bool(rng.integers(2)), bool(rng.integers(2))
    k = int(rng.integers(1, 60)); n = int(rng.integers(1, 400)); S = int(rng.choice([8, 33, 64, 200, 256, 512])); step = float(rng.choice([0.004, 0.01, 0.03]))
    spec0 = random_scene(seed, k=k, method=method, mesh=mesh)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    inst = gpu_instancer(box, tr, textures=["", "", "", "", "light"], instance_sampling_method=method, mesh=(spec0.mesh_v, spec0.mesh_f) if mesh else None,
                         cast_shadow_rays=shadows)
    o, d = random_rays(seed, n)
    params = np.tile(F([[1, 1, 1, .1, 0.3, 0.2, 1]]), (n, 1))
    dense = inst.get_model_input(o, d, params, S, step, seed=seed)
    sparse = inst.get_model_input(o, d, params, S, step, seed=seed, sparse=True, fill=float("nan"))
    live = dense[3] > 0
    ok = torch.equal(dense[3], sparse[3])
    for kk in (0, 1, 2, 6, 7, 9):
        x, y = dense[kk], sparse[kk]
        m = live if x.dim() == 2 else live[..., None].expand_as(x)
        ok = ok and torch.equal(x[m], y[m]) and (x.dtype != torch.float32 or bool(torch.isnan(y[~m]).all()))
    for kk in (4, 5, 8):
        ok = ok and torch.equal(dense[kk], sparse[kk])
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=0.35, step_size=step, render_chunk=int(rng.choice([64, 4096])), density_scale=40.0)
    dd = lambda x: torch.as_tensor(x, device=dv)[None]
    tt = np.tile(F([[1.0, 2.0]]), (n, 1))
    call = lambda **kw: r(dd(o), dd(d), dd(tt), parameters=torch.as_tensor(params[:1], device=dv), cone_scale=dd(np.full((n, 1), 2e-3, F)), instancer_seed=seed, **kw)
    i0 = call(instancer_sparse=False); i1 = call(instancer_fill=float("nan"))
    r.raise_if_nonfinite()
    ok = ok and torch.equal(i0["color_pred"], i1["color_pred"]) and torch.equal(i0["alpha_pred"], i1["alpha_pred"])
    live_frac.append(float(live.float().mean().item()))
    if not ok:
        fails.append(dict(case=case, seed=seed, method=method, mesh=mesh, shadows=shadows, k=k, n=n, S=S, step=step))
        print("FAILED", fails[-1], flush=True)
print(json.dumps({"cases": a.cases, "seed": a.seed, "failed": len(fails), "mean_fraction_of_rows_with_dists_gt_0": float(np.mean(live_frac)), "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if fails else 0)
