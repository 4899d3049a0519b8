"""Seeded fuzz of the fused render path against the float64 oracle: odd ray counts (fewer / more rays than the 1024
persistent waves, counts that are not multiples of the 4-wave lockstep groups), ragged sample counts, random culling,
per-image and per-ray parameters, both precisions.  `-m gpu`."""

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from tests.common import TOL, make_model

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ORACLE_RAYS = 800       # the GPU renders every ray of a case; the three restatement passes (float64, float32, float64 on float32 points) run on
                        # at most this many of the hit rays -- a seeded draw that always holds the first and the last hit ray (block edges)


def oracle_subset(hit, rng):
    idx = np.nonzero(hit)[0]
    if idx.size > ORACLE_RAYS:
        idx = np.unique(np.concatenate([idx[:8], idx[-8:], rng.choice(idx, size=ORACLE_RAYS - 16, replace=False)]))
    sel = np.zeros_like(hit); sel[idx] = True
    return sel

CASES = [(seed, prec) for seed in range(12) for prec in ("float32", "fp16x3")]


@pytest.mark.parametrize("seed,precision", CASES)
def test_fuzz_render_rays(seed, precision):
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    rng = np.random.default_rng(1000 + seed)
    family = ["carpet", "grass", "grass_filtered", "fur"][seed % 4]
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], seed=seed, dense_media=bool(seed & 1))
    n = int(rng.choice([1, 3, 5, 63, 257, 1023, 1025, 1029, 2050, 4099]))
    S = int(rng.choice([2, 3, 31, 32, 33, 64, 65, 96, 100, 127]))
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"], seed=seed)
    t = t.copy()
    t[rng.uniform(size=n) < [0.0, 0.3, 0.9][seed % 3]] = np.inf           # none / some / most rays culled
    per_ray = bool(seed % 5 == 0)                                          # parameters per ray (B = n, HW = 1) or per image
    P = sum(fam["n_parameters"])
    params = (rng.uniform(0, 1, size=(n if per_ray else 1, P)) * np.asarray(fam["params"], np.float32)).astype(np.float32)
    bk = bool(seed & 2)
    dv = torch.device("cuda", 0)
    d = lambda a: torch.as_tensor(a, device=dv)
    shape = (lambda a: a[:, None]) if per_ray else (lambda a: a[None])
    r = Renderer(model=model, n_samples=S, perturb=False, blur_idx=fam["blur_idx"], precision=precision, map_exr=bool(seed & 4))
    out = r(d(shape(ro)), d(shape(rd)), d(shape(t)), parameters=d(params), cone_scale=d(shape(cone)), composite_bkgd=bk,
            bkgd_color=[0.2, 0.5, 0.9])
    r.raise_if_nonfinite()
    got = np.concatenate([out["color_pred"].cpu().numpy().reshape(n, 3), out["alpha_pred"].cpu().numpy().reshape(n, 1)], -1)
    all_hit = np.isfinite(t[:, 0])
    want = np.zeros((n, 4))
    if bk:
        want[~all_hit, :3] = (0.2, 0.5, 0.9)
    assert np.array_equal(got[~all_hit], want[~all_hit].astype(np.float32))        # culled rays exact
    assert np.isfinite(got).all()
    hit = oracle_subset(all_hit, rng)
    if hit.any():
        pr = params if per_ray else np.repeat(params, n, 0)
        ref = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], pr[hit], cone[hit], S, bk, (0.2, 0.5, 0.9), fam["blur_idx"],
                              bool(seed & 4), dtype=np.float64)
        want[hit, :3] = ref["color_pred"]; want[hit, 3] = ref["alpha_pred"]
    if hit.any():
        # Two strict gates (tests/common.py: TOL = 1e-4): against the float32 restatement (the north star's comparison) and
        # against the float64 network on the float32 sample points (the arithmetic the kernel answers for; oracle render_rays:
        # points_dtype).  `want` (all-float64) adds the float32 rounding of the sample positions in front of sin(2^9 x), which is the
        # reference's own and up to a few 1e-4 with the dense-media weights: bounded by exactly that model, measured oracle to oracle.
        kw2 = dict(blur_idx=fam["blur_idx"], map_exr=bool(seed & 4))
        ref32 = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], pr[hit], cone[hit], S, bk, (0.2, 0.5, 0.9), dtype=np.float32, **kw2)
        refn = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], pr[hit], cone[hit], S, bk, (0.2, 0.5, 0.9), dtype=np.float64,
                               points_dtype=np.float32, **kw2)
        w32 = np.concatenate([ref32["color_pred"], ref32["alpha_pred"][:, None]], -1).astype(np.float64)
        wn = np.concatenate([refn["color_pred"], refn["alpha_pred"][:, None]], -1)
        scale = max(float(np.max(np.abs(want[hit]))), 1e-3)
        input_floor = float(np.max(np.abs(wn - want[hit]))) / scale
        err_net = float(np.max(np.abs(got[hit] - wn))) / scale
        assert float(np.max(np.abs(got[hit] - w32))) / scale <= TOL
        assert err_net <= TOL
        assert float(np.max(np.abs(got[hit] - want[hit]))) / scale <= err_net + input_floor * (1 + 1e-6) + 1e-9


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_around_the_direction_blocks(seed):
    """Ray counts around the 8192-ray blocks in which a launch hands out the hit list (32 rays per workgroup and block, their
    per-ray rows computed into LDS by dir_block): partial last blocks, most rays culled, in-kernel jitter on and off, both
    precisions (60 more seeds: tools/dev/fuzz_more.py)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    rng = np.random.default_rng(5000 + seed)
    family = ["carpet", "grass", "grass_filtered", "fur"][seed % 4]
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], seed=seed, dense_media=bool(seed & 1))
    n = [8191, 8193, 16385, 9000, 24577, 8192][seed]
    S = [17, 33, 32, 2, 40, 32][seed]
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"], seed=seed)
    t = t.copy(); t[rng.uniform(size=n) < [0.0, 0.4, 0.97][seed % 3]] = np.inf
    params = (rng.uniform(0, 1, size=(1, sum(fam["n_parameters"]))) * np.asarray(fam["params"], np.float32)).astype(np.float32)
    perturb = bool(seed % 2)
    all_hit = np.isfinite(t[:, 0])
    hit = oracle_subset(all_hit, rng)
    tz = np.where(np.isfinite(t), t, 0).astype(np.float32)
    z = orc.z_values_perturbed(tz, S, 77 + seed, np.float32) if perturb else None
    kw = dict(z_override=None if z is None else z[hit])
    pr = np.repeat(params, hit.sum(), 0)
    ref = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], pr, cone[hit], S, False, (1, 1, 1.), fam["blur_idx"], dtype=np.float64, **kw)
    ref32 = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], pr, cone[hit], S, False, (1, 1, 1.), fam["blur_idx"], dtype=np.float32, **kw)
    refn = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], pr, cone[hit], S, False, (1, 1, 1.), fam["blur_idx"], dtype=np.float64,
                           points_dtype=np.float32, **kw)
    want = np.concatenate([ref["color_pred"], ref["alpha_pred"][:, None]], -1)
    w32 = np.concatenate([ref32["color_pred"], ref32["alpha_pred"][:, None]], -1)
    wn = np.concatenate([refn["color_pred"], refn["alpha_pred"][:, None]], -1)
    scale = max(float(np.abs(want).max()), 1e-3)
    input_floor = float(np.abs(wn - want).max()) / scale              # float32 rounding of the inputs: oracle to oracle
    dv = torch.device("cuda", 0)
    d = lambda a: torch.as_tensor(a, device=dv)
    for prec in ("float32", "fp16x3"):
        r = Renderer(model=model, n_samples=S, perturb=perturb, blur_idx=fam["blur_idx"], precision=prec)
        out = r(d(ro[None]), d(rd[None]), d(t[None]), parameters=d(params), cone_scale=d(cone[None]), seed=77 + seed)
        r.raise_if_nonfinite()
        got = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
        assert np.all(got[~all_hit] == 0) and np.isfinite(got).all()
        err_net = float(np.abs(got[hit] - wn).max()) / scale
        assert float(np.abs(got[hit] - w32).max()) / scale <= TOL                       # vs the float32 restatement
        assert err_net <= TOL                                                            # vs the float64 network on the float32 points
        assert float(np.abs(got[hit] - want).max()) / scale <= err_net + input_floor * (1 + 1e-6) + 1e-9   # all-float64: + the input rounding, nothing more
