"""The opt-in fp16x3 precision of the fused render kernel (include/nerftex.h: ntx_precision) against the oracle.  `-m gpu`.

Same bar as the float32 kernel: rel-L-inf <= 1e-4 against the float64 restatement of the reference.  The float32
kernel stays the default; these tests also bound the distance between the two precisions.
"""

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from tests.common import TOL, camera_rays, make_model

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def to_dev(*arrs):
    return [torch.as_tensor(a, device=torch.device("cuda", 0)) for a in arrs]


def _render(model, precision, S, ro, rd, t, params, cone, bk=False, blur_idx=None, z=None, map_exr=False):
    from nerf_tex_amd.renderer import Renderer
    r = Renderer(model=model, n_samples=S, perturb=z is not None, blur_idx=blur_idx, precision=precision, map_exr=map_exr)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0],
            composite_bkgd=bk, bkgd_color=[1, 1, 1.], z_vals=None if z is None else to_dev(z)[0])
    r.raise_if_nonfinite()
    return np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)


@pytest.mark.parametrize("family,S", [("carpet", 64), ("carpet", 33), ("grass", 128), ("fur", 2), ("grass_filtered", 48)])
@pytest.mark.parametrize("dense", [True, False])
def test_fp16x3_render_rays_camera(family, S, dense):
    from nerf_tex_amd import synthetic
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], dense_media=dense)
    h, wd = 24, 20
    (ro, rd, t, cone), _, _ = camera_rays(family, h, wd)
    params = np.asarray([fam["params"]], np.float32)
    got = _render(model, "fp16x3", S, ro, rd, t, params, cone, bk=True, blur_idx=fam["blur_idx"])
    f32 = _render(model, "float32", S, ro, rd, t, params, cone, bk=True, blur_idx=fam["blur_idx"])
    ref = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, True, (1., 1., 1.),
                            fam["blur_idx"], False, dtype=np.float64)
    want = np.concatenate([ref["color_pred"][0], ref["alpha_pred"][0][:, None]], -1)
    miss = ~np.isfinite(t[:, 0])
    assert miss.any() and (~miss).any()
    assert np.all(got[miss, 3] == 0.0) and np.all(got[miss, :3] == 1.0)      # culled rays exact (renderer.py:81-86)
    err, err32 = orc.rel_linf(got, want), orc.rel_linf(got, f32)
    assert err <= TOL, (err, err32)
    assert err32 <= TOL, err32
    assert not np.array_equal(got, f32) or S == 2      # it really is the other kernel


def test_fp16x3_plain_nerf_and_exr():
    from nerf_tex_amd import synthetic
    fam = synthetic.FAMILIES["carpet"]
    model, spec, w = make_model((0, 0), kind="Nerf", dense_media=True)
    n, S = 300, 64
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    params = np.zeros((1, 0), np.float32)
    got = _render(model, "fp16x3", S, ro, rd, t, params, cone, map_exr=True)
    ref = orc.render_rays(w, spec, ro, rd, t, np.zeros((n, 0), np.float32), cone, S, False, (1, 1, 1.), map_exr=True,
                          dtype=np.float64)
    want = np.concatenate([ref["color_pred"], ref["alpha_pred"][:, None]], -1)
    assert orc.rel_linf(got, want) <= TOL


def test_fp16x3_zvals_and_weights():
    """caller-supplied depths and the per-sample compositing weights (hierarchical path) at fp16x3"""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    model, spec, w = make_model((1, 6), dense_media=True)
    n, S, NI = 200, 32, 32
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    params = np.asarray([fam["params"]], np.float32)
    r = Renderer(model=model, n_samples=S, n_importance=NI, perturb=True, precision="fp16x3")
    z0 = orc.z_values(t, S, np.float32)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0],
            z_vals=to_dev(z0)[0])
    r.raise_if_nonfinite()
    pr = np.repeat(params, n, 0)
    ref = orc.render_rays(w, spec, ro, rd, t, pr, cone, S, False, (1, 1, 1.), dtype=np.float64)
    got = np.concatenate([out["color_pred_coarse"][0].cpu().numpy(), out["alpha_pred_coarse"][0].cpu().numpy()[:, None]], -1)
    want = np.concatenate([ref["color_pred"], ref["alpha_pred"][:, None]], -1)
    assert orc.rel_linf(got, want) <= TOL
    # fine pass: the oracle on the HIP path's own merged depths (the sampling itself is covered at float32 in
    # test_gpu_parity.test_hierarchical_sampling)
    zg = r._last_z.cpu().numpy()
    assert np.all(np.diff(zg, axis=-1) >= 0)
    ref2 = orc.render_rays(w, spec, ro, rd, t, pr, cone, S + NI, False, (1, 1, 1.), z_override=zg, dtype=np.float64)
    gotf = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
    wantf = np.concatenate([ref2["color_pred"], ref2["alpha_pred"][:, None]], -1)
    assert orc.rel_linf(gotf, wantf) <= TOL


def test_fp16x3_nan_propagates_and_unsupported_blur():
    from nerf_tex_amd import synthetic, _lib
    from nerf_tex_amd.renderer import Renderer, MipRenderer
    fam = synthetic.FAMILIES["carpet"]
    model, spec, w = make_model((1, 6))
    n, S = 64, 32
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    ro = ro.copy(); ro[5, 1] = np.nan
    params = np.asarray([fam["params"]], np.float32)
    r = Renderer(model=model, n_samples=S, perturb=False, precision="fp16x3")
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0])
    c = out["color_pred"][0].cpu().numpy()
    assert np.isnan(c[5]).all() and np.isfinite(np.delete(c, 5, 0)).all()
    with pytest.raises(FloatingPointError):
        r.raise_if_nonfinite()
    with pytest.raises(ValueError):
        Renderer(model=model, precision="fp8")
    # a blur_idx on an appearance parameter is a per-sample scaling of a direction-side input: float32 only
    rb = Renderer(model=model, n_samples=S, perturb=False, precision="fp16x3", blur_idx=4)
    ro[5, 1] = 0.0
    with pytest.raises(_lib.NtxError) as e:
        rb(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0])
    assert e.value.code == _lib.NTX_E_UNSUPPORTED



def test_fp16x3_full_size_is_split_invariant():
    """BASELINE config 1 size at fp16x3: the lockstep kernel hands rays to waves through a compacted list whose order
    depends on the launch, yet each ray's result depends only on the ray -- bit-identical however the rays are split
    across calls (what sharding across GPUs relies on), alpha in [0,1], premultiplied colours below alpha."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    model, spec, w = make_model((1, 6), dense_media=True)
    n, S = 800 * 800, 64
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    t = t.copy(); t[::7] = np.inf                      # every 7th ray culled: the hit list is a real compaction
    params = to_dev(np.asarray([fam["params"]], np.float32))[0]
    r = Renderer(model=model, n_samples=S, perturb=False, precision="fp16x3")
    dro, drd, dt, dcone = to_dev(ro, rd, t, cone)
    full = r(dro[None], drd[None], dt[None], parameters=params, cone_scale=dcone[None])
    c, a = full["color_pred"][0], full["alpha_pred"][0]
    assert torch.isfinite(c).all() and torch.isfinite(a).all()
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-6 and bool((c <= a[:, None] + 1e-6).all())
    assert bool((a[::7] == 0).all()) and bool((c[::7] == 0).all())
    cuts = [0, 100_003, 400_001, n]
    pc, pa = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = r(dro[None, lo:hi], drd[None, lo:hi], dt[None, lo:hi], parameters=params, cone_scale=dcone[None, lo:hi])
        pc.append(o["color_pred"][0]); pa.append(o["alpha_pred"][0])
    assert torch.equal(torch.cat(pc), c) and torch.equal(torch.cat(pa), a)
    idx = np.random.default_rng(0).choice(np.where(np.isfinite(t[:, 0]))[0], 256, replace=False)
    ref = orc.render_rays(w, spec, ro[idx], rd[idx], t[idx], np.repeat(np.asarray([fam["params"]], np.float32), 256, 0), cone[idx],
                          S, False, (1, 1, 1.), dtype=np.float64)
    got = np.concatenate([c[idx].cpu().numpy(), a[idx].cpu().numpy()[:, None]], -1)
    want = np.concatenate([ref["color_pred"], ref["alpha_pred"][:, None]], -1)
    assert orc.rel_linf(got, want) <= TOL


@pytest.mark.parametrize("kind,npar", [("ParamNerf", (1, 6)), ("ParamNerf", (2, 3)), ("Nerf", (0, 0))])
@pytest.mark.parametrize("m", [1, 33, 4096 * 3 + 5])
def test_fp16x3_mlp_forward(kind, npar, m):
    """model((pos, dirs, params)) alone at fp16x3 (the lockstep stand-alone MLP kernel): raw colour and density against the
    float64 restatement at the float32 kernel's own gate."""
    from tests.common import random_samples
    model, spec, w = make_model(npar, kind=kind)
    model.precision = "fp16x3"
    pos, dirs, params = random_samples(m, sum(npar))
    c, a = model(tuple(to_dev(pos, dirs, params)))
    rc, ra = orc.model_forward(w, spec, pos, dirs, params, dtype=np.float64)
    got = np.concatenate([c.cpu().numpy(), a.cpu().numpy().reshape(m, 1)], -1)
    want = np.concatenate([rc, np.asarray(ra).reshape(m, 1)], -1)
    assert orc.rel_linf(got, want) <= 2e-5
    model.precision = "float32"
    c32, a32 = model(tuple(to_dev(pos, dirs, params)))
    got32 = np.concatenate([c32.cpu().numpy(), a32.cpu().numpy().reshape(m, 1)], -1)
    assert orc.rel_linf(got, got32) <= 5e-6 and (m < 33 or not np.array_equal(got, got32))
