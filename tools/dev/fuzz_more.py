"""dev helper (GPU box): more seeds of tests/test_gpu_fuzz.py, and a second fuzz with ray counts around the 8192-ray
blocks of the per-workgroup direction rows (dir_block): partial last blocks, culled rays, both precisions, perturb on/off.
usage: python tools/dev/fuzz_more.py [n_seeds]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nerftex_oracle as orc
from tests.common import TOL, make_model
from tests.test_gpu_fuzz import test_fuzz_render_rays
from nerf_tex_amd import synthetic
from nerf_tex_amd.renderer import Renderer

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(12, 12 + n_seeds):
    for prec in ("float32", "fp16x3"):
        try:
            test_fuzz_render_rays(seed, prec)
        except AssertionError as e:
            bad += 1; print("FAIL small", seed, prec, str(e)[:200], flush=True)
print("small fuzz done, failures:", bad, flush=True)

dv = torch.device("cuda", 0)
d = lambda a: torch.as_tensor(a, device=dv)
for seed in range(n_seeds // 2):
    rng = np.random.default_rng(5000 + seed)
    family = ["carpet", "grass", "grass_filtered", "fur"][seed % 4]
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], seed=seed, dense_media=bool(seed & 1))
    n = int(rng.choice([8191, 8192, 8193, 9000, 16383, 16385, 20000, 24577, 40001]))
    S = int(rng.choice([2, 17, 32, 33, 40]))
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"], seed=seed)
    t = t.copy(); t[rng.uniform(size=n) < [0.0, 0.4, 0.97][seed % 3]] = np.inf
    params = (rng.uniform(0, 1, size=(1, sum(fam["n_parameters"]))) * np.asarray(fam["params"], np.float32)).astype(np.float32)
    perturb = bool(seed % 2)
    blur = fam["blur_idx"] if seed % 8 < 6 else None          # grass_filtered: also without blur_idx (geometry hoisting path n/a: variant 2)
    hit = np.isfinite(t[:, 0])
    tz = np.where(np.isfinite(t), t, 0).astype(np.float32)
    z = orc.z_values_perturbed(tz, S, 77 + seed, np.float32) if perturb else None
    ref = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, hit.sum(), 0), cone[hit], S, False, (1, 1, 1.), blur,
                          z_override=None if z is None else z[hit], dtype=np.float64) if hit.any() else None
    for prec in ("float32", "fp16x3"):
        r = Renderer(model=model, n_samples=S, perturb=perturb, blur_idx=blur, precision=prec)
        out = r(d(ro[None]), d(rd[None]), d(t[None]), parameters=d(params), cone_scale=d(cone[None]), seed=77 + seed)
        r.raise_if_nonfinite()
        got = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
        ok = np.all(got[~hit] == 0)
        if ref is not None:
            want = np.concatenate([ref["color_pred"], ref["alpha_pred"][:, None]], -1)
            ref32 = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, hit.sum(), 0), cone[hit], S, False, (1, 1, 1.), blur,
                                    z_override=None if z is None else z[hit], dtype=np.float32)
            w32 = np.concatenate([ref32["color_pred"], ref32["alpha_pred"][:, None]], -1)
            scale = max(float(np.abs(want).max()), 1e-3)
            floor = float(np.abs(w32 - want).max()) / scale
            e32 = float(np.abs(got[hit] - w32).max()) / scale; e64 = float(np.abs(got[hit] - want).max()) / scale
            ok = ok and e32 <= TOL and e64 <= max(TOL, 1.25 * floor)
        if not ok:
            bad += 1; print("FAIL big", seed, family, n, S, prec, perturb, flush=True)
print("fuzz done, failures:", bad, flush=True)
