"""The flex kernel family: architectures of network.model.ParamNerf / Nerf (model.py:58, :9) other than the reference configs'
8 x 256 / skips [4] / color_depth 1 -- depth, width, skips, color_depth as run-time facts of ONE kernel set (a loop over 256-wide
layers, ntx_layout.h "flex family") -- against the oracle's general restatement of model.py:104-123, through the C ABI.  `-m gpu`."""

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from tests.common import TOL, camera_rays, make_model, random_samples

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    return torch.device("cuda", 0)


def to_dev(*arrs):
    return [torch.as_tensor(a, device=dev()) for a in arrs]


def rgba_of(out, b=0):
    return np.concatenate([out["color_pred"][b].cpu().numpy(), out["alpha_pred"][b].cpu().numpy()[:, None]], -1)


ARCHS = [
    # (kind, n_parameters, arch)
    ("ParamNerf", (1, 6), dict(depth=4, width=128, skips=[2], color_depth=1)),
    ("ParamNerf", (1, 6), dict(depth=8, width=256, skips=[4], color_depth=0)),     # C2 straight on [dir_map, feature]
    ("ParamNerf", (1, 4), dict(depth=8, width=256, skips=[4], color_depth=2)),
    ("ParamNerf", (2, 3), dict(depth=10, width=256, skips=[3, 6], color_depth=1)),   # two skip layers: NTX_SKIP_MASK encoding
    ("ParamNerf", (1, 6), dict(depth=6, width=64, skips=[], color_depth=1)),         # no skip at all
    ("ParamNerf", (3, 2), dict(depth=2, width=200, skips=[0], color_depth=3)),       # skip into the last trunk layer; odd-ish width
    ("ParamNerf", (1, 6), dict(depth=1, width=256, skips=[4], color_depth=1)),       # a one-layer trunk
    ("ParamNerf", (0, 0), dict(depth=5, width=96, skips=[1, 2, 3], color_depth=4)),  # parameter-less ParamNerf
    ("ParamNerf", (1, 6), dict(depth=24, width=32, skips=[4, 9, 14, 19], color_depth=1)),   # the deepest trunk the family takes
    ("Nerf", (0, 0), dict(depth=6, width=128, skips=[3])),
    ("Nerf", (0, 0), dict(depth=8, width=256, skips=[2])),
]


@pytest.mark.parametrize("kind,npar,arch", ARCHS)
@pytest.mark.parametrize("m", [33, 4096 + 17])
def test_mlp_forward_any_architecture(kind, npar, arch, m):
    model, spec, w = make_model(npar, kind, arch=arch)
    assert model.layer_table() == orc.layer_table(spec)
    pos, dirs, params = random_samples(m, sum(npar))
    color, alpha = model(tuple(to_dev(pos, dirs, params)))
    rc, ra = orc.model_forward(w, spec, pos, dirs, params, np.float64)
    out = np.concatenate([color.cpu().numpy(), alpha.cpu().numpy()], -1)
    ref = np.concatenate([rc, ra], -1)
    err = orc.rel_linf(out, ref)
    assert err <= TOL, err          # north-star gate
    assert err <= 3e-5, err         # what exact-f32 MFMA reaches on glorot weights (deeper trunks accumulate a little more)


ARCHS_PD = [   # param_depth > 0 (model.py:88-101): Dense(param_width, relu) layers on the parameter features, per branch
    ((1, 6), dict(depth=8, width=256, skips=[4], color_depth=1, param_depth=1)),               # the reference architecture + one layer
    ((2, 3), dict(depth=8, width=256, skips=[4], color_depth=1, param_depth=2)),
    ((1, 4), dict(depth=4, width=128, skips=[1, 2], color_depth=0, param_depth=3, param_width=64)),   # colour half on [FF(dir) | A]
    ((4, 8), dict(depth=6, width=200, skips=[0, 4], color_depth=2, param_depth=4, param_width=100)),  # every parameter slot in use
    ((0, 5), dict(depth=5, width=256, skips=[2], color_depth=1, param_depth=2)),               # appearance branch only
    ((3, 0), dict(depth=5, width=256, skips=[2], color_depth=1, param_depth=2)),               # geometry branch only
    ((0, 0), dict(depth=5, width=256, skips=[2], color_depth=1, param_depth=2)),               # no parameters: no branches at all
    ((1, 6), dict(depth=1, width=96, skips=[], color_depth=3, param_depth=1, param_width=2)),
]


@pytest.mark.parametrize("npar,arch", ARCHS_PD)
def test_mlp_forward_with_parameter_branches(npar, arch):
    model, spec, w = make_model(npar, "ParamNerf", arch=arch)
    assert model.layer_table() == orc.layer_table(spec) and any(n.startswith("param_") for n, _, _ in model.layer_table()) == (sum(npar) > 0)
    for m in (33, 2000):
        pos, dirs, params = random_samples(m, sum(npar), seed=m)
        color, alpha = model(tuple(to_dev(pos, dirs, params)))
        rc, ra = orc.model_forward(w, spec, pos, dirs, params, np.float64)
        out = np.concatenate([color.cpu().numpy(), alpha.cpu().numpy()], -1)
        err = orc.rel_linf(out, np.concatenate([rc, ra], -1))
        assert err <= 3e-5, (err, npar, arch)


@pytest.mark.parametrize("blur,perturb", [(None, False), (0, True), (2, False)])
def test_render_rays_with_parameter_branches(blur, perturb):
    """Renderer.__call__ with param_depth = 2: blur_idx on a geometry parameter (its branch sees the scaled value per sample,
    renderer.py:155-158) and on an appearance parameter, in-kernel jitter."""
    from nerf_tex_amd.renderer import Renderer
    npar, S = (2, 3), 48
    model, spec, w = make_model(npar, "ParamNerf", dense_media=True, arch=dict(depth=6, width=256, skips=[2], color_depth=1, param_depth=2))
    (ro, rd, t, cone), _, _ = camera_rays("carpet", 14, 12)
    cone = (cone * 30).astype(np.float32)
    params = np.random.default_rng(3).uniform(0.1, 1.0, size=(1, 5)).astype(np.float32)
    r = Renderer(model=model, n_samples=S, perturb=perturb, blur_idx=blur)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0], seed=5)
    r.raise_if_nonfinite()
    got = rgba_of(out)
    hit = np.isfinite(t[:, 0])
    z = orc.z_values_perturbed(np.where(np.isfinite(t), t, 0).astype(np.float32), S, 5, np.float32)[hit] if perturb else None
    ref = {}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        o = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, int(hit.sum()), 0), cone[hit], S, False, (1, 1, 1.),
                            blur_idx=blur, z_override=z, dtype=dt)
        full = np.zeros((t.shape[0], 4)); full[hit] = np.concatenate([o["color_pred"], o["alpha_pred"][:, None]], -1)
        ref[name] = full
    assert np.all(got[~hit] == 0)
    assert orc.rel_linf(got, ref["f32"]) <= TOL
    assert orc.rel_linf(got, ref["f64"]) <= TOL + orc.rel_linf(ref["f32"], ref["f64"])


@pytest.mark.parametrize("seed", range(10))
def test_mlp_forward_random_architectures(seed):
    """Seeded random architectures inside the family's limits: depth 1..24, width 2..256, any set of skips below depth-1,
    color_depth 0..4, n_parameters up to [4, 8] (or plain Nerf)."""
    rng = np.random.default_rng(1000 + seed)
    depth = int(rng.integers(1, 25)) if seed % 3 else int(rng.integers(1, 7))
    width = int(rng.choice([2, 7, 32, 64, 100, 128, 192, 255, 256]))
    skips = sorted(int(i) for i in np.flatnonzero(rng.uniform(size=max(depth - 1, 0)) < 0.3))
    kind = "Nerf" if seed % 5 == 4 else "ParamNerf"
    npar = (0, 0) if kind == "Nerf" else (int(rng.integers(0, 5)), int(rng.integers(0, 9)))
    arch = dict(depth=depth, width=width, skips=skips, color_depth=int(rng.integers(0, 5)))
    if kind == "ParamNerf" and (depth, width, tuple(skips), arch["color_depth"]) == (8, 256, (4,), 1):
        arch["color_depth"] = 2                             # (that one is the tuned families' architecture)
    model, spec, w = make_model(npar, kind, seed=seed, arch=arch)
    pos, dirs, params = random_samples(700 + seed, sum(npar), seed=seed)
    color, alpha = model(tuple(to_dev(pos, dirs, params)))
    rc, ra = orc.model_forward(w, spec, pos, dirs, params, np.float64)
    out = np.concatenate([color.cpu().numpy(), alpha.cpu().numpy()], -1)
    err = orc.rel_linf(out, np.concatenate([rc, ra], -1))
    assert err <= 3e-5, (err, kind, npar, arch)


def test_flex_kernels_give_the_tuned_kernels_bits(monkeypatch):
    """The 8 x 256 / [4] / 1 model forced onto the flex kernels (NERFTEX_FORCE_FLEX, read by ntx_create): every accumulator sees the
    same bias and the same products in the same order as in the straight-line kernels, so the network outputs and the rendered
    image are the same BITS as the generic family's (which in turn equal the tuned [1,6] family's)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    ro, rd, t, cone = synthetic.all_hit_rays(1500, fam["b_0"], fam["b_1"], fam["cam"])
    params = to_dev(np.asarray([fam["params"]], np.float32))[0]
    args = to_dev(ro[None], rd[None], t[None])
    pos, dirs, pp = random_samples(3000, 7)

    def run():
        model, _, _ = make_model((1, 6), dense_media=True)
        c, a = model(tuple(to_dev(pos, dirs, pp)))
        img = rgba_of(Renderer(model=model, n_samples=64, perturb=False)(*args, parameters=params, cone_scale=to_dev(cone[None])[0]))
        return np.concatenate([c.cpu().numpy(), a.cpu().numpy()], -1), img

    net_t, img_t = run()
    monkeypatch.setenv("NERFTEX_FORCE_FLEX", "1")
    net_f, img_f = run()
    monkeypatch.delenv("NERFTEX_FORCE_FLEX")
    assert np.array_equal(net_t, net_f)
    assert np.array_equal(img_t, img_f)


@pytest.mark.parametrize("kind,npar,arch,blur", [("ParamNerf", (1, 6), dict(depth=4, width=128, skips=[2], color_depth=1), None),
                                                  ("ParamNerf", (2, 3), dict(depth=10, width=256, skips=[3, 6], color_depth=2), 0),
                                                  ("ParamNerf", (1, 4), dict(depth=8, width=256, skips=[4], color_depth=0), 3),
                                                  ("Nerf", (0, 0), dict(depth=6, width=128, skips=[3]), None)])
@pytest.mark.parametrize("S,perturb", [(40, False), (64, True)])
def test_render_rays_any_architecture(kind, npar, arch, blur, S, perturb):
    """Renderer.__call__ (renderer.py:47-213) with a flex model: camera grid with hits and misses, blur_idx on a geometry and on
    an appearance parameter, in-kernel jitter (depths restated by the oracle's generator)."""
    from nerf_tex_amd.renderer import Renderer
    model, spec, w = make_model(npar, kind, dense_media=True, arch=arch)
    (ro, rd, t, cone), _, _ = camera_rays("carpet", 14, 12)
    cone = (cone * 30).astype(np.float32)
    P = sum(npar)
    params = np.random.default_rng(P + 5).uniform(0.1, 1.0, size=(1, P)).astype(np.float32)
    args = to_dev(ro[None], rd[None], t[None])
    kw = dict(parameters=to_dev(params)[0] if P else torch.zeros((1, 0), device=dev()), cone_scale=to_dev(cone[None])[0])
    r = Renderer(model=model, n_samples=S, perturb=perturb, blur_idx=blur)
    out = r(*args, **kw, seed=77)
    r.raise_if_nonfinite()
    got = rgba_of(out)
    z = None
    if perturb:                                                                # the product's own stream, restated (ray index = index in the call)
        z = orc.z_values_perturbed(np.where(np.isfinite(t), t, 0).astype(np.float32), S, 77, np.float32)
    ref = {}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        if z is None:
            o = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.), blur, False, dtype=dt)
            ref[name] = np.concatenate([o["color_pred"][0], o["alpha_pred"][0][:, None]], -1).astype(np.float64)
        else:
            hit = np.isfinite(t[:, 0])
            o = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, int(hit.sum()), 0), cone[hit], S, False, (1, 1, 1.),
                                blur_idx=blur, z_override=z[hit], dtype=dt)
            full = np.zeros((t.shape[0], 4))
            full[hit] = np.concatenate([o["color_pred"], o["alpha_pred"][:, None]], -1)
            ref[name] = full
    miss = ~np.isfinite(t[:, 0])
    assert miss.any() and np.all(got[miss] == 0)
    assert orc.rel_linf(got, ref["f32"]) <= TOL
    floor = orc.rel_linf(ref["f32"], ref["f64"])          # the float32 restatement's own distance from the truth on these weights
    assert orc.rel_linf(got, ref["f64"]) <= TOL + floor


def test_instance_renderer_flex_model():
    """The InstanceRenderer tail (renderer.py:247-354) with a flex model: the per-sample instanced kernel."""
    from nerf_tex_amd.renderer import InstanceRenderer
    from tests.test_gpu_instance import FakeInstancer
    npar = (1, 6)
    model, spec, w = make_model(npar, dense_media=True, arch=dict(depth=5, width=192, skips=[1, 3], color_depth=2))
    P = sum(npar)
    S, n = 200, 61
    inst = FakeInstancer(P, seed=S + P, run_len=12)
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=0.09, step_size=0.002, blur_idx=None,
                         render_chunk=10_000, density_scale=400.0)
    rng = np.random.default_rng(1)
    ro = rng.normal(size=(1, n, 3)).astype(np.float32); rd = rng.normal(size=(1, n, 3)).astype(np.float32)
    t = np.tile(np.asarray([[1.0, 2.0]], np.float32), (1, n, 1))
    params = rng.uniform(0.2, 1, size=(1, P)).astype(np.float32)
    cone = rng.uniform(1e-3, 5e-3, size=(1, n, 1)).astype(np.float32)
    d = lambda a: torch.as_tensor(a, device=dev())
    out = r(d(ro), d(rd), d(t), parameters=d(params), cone_scale=d(cone))
    r.raise_if_nonfinite()
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map, hit = inst.last
    rc, ra = orc.instance_evaluate_model(w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id,
                                         hit, params_map, cone[0], None, 0.09, r.density_scale, r.density_reweighting,
                                         r.map_exr, False, (1., 1., 1.), r.instance_color, dtype=np.float64)
    got = rgba_of(out)
    want = np.concatenate([rc, ra[:, None]], -1)
    assert orc.rel_linf(got, want) <= TOL
    assert float(np.max(ra)) > 0.3


def test_what_the_flex_family_refuses():
    from nerf_tex_amd import _lib
    from nerf_tex_amd.renderer import Renderer
    # a skip behind the last trunk layer widens the alpha head and the feature layer (model.py:107-114): not built
    m, _, _ = make_model((1, 6), arch=dict(depth=4, width=128, skips=[3], color_depth=1))
    with pytest.raises(_lib.NtxError) as e:
        m.ctx(0)
    assert e.value.code == _lib.NTX_E_UNSUPPORTED
    for bad in (dict(depth=25, width=64, skips=[], color_depth=1), dict(depth=4, width=512, skips=[], color_depth=1),
                dict(depth=4, width=128, skips=[], color_depth=5), dict(param_depth=5), dict(param_depth=1, param_width=129),
                dict(param_depth=-1)):
        m, _, _ = make_model((1, 6), arch=bad)
        with pytest.raises(_lib.NtxError) as e:
            m.ctx(0)
        assert e.value.code == _lib.NTX_E_UNSUPPORTED, bad
    # fp16x3 is built for the 8 x 256 families only
    m, _, _ = make_model((1, 6), arch=dict(depth=4, width=128, skips=[2], color_depth=1))
    fam_rays = camera_rays("carpet", 6, 6)[0]
    ro, rd, t, cone = fam_rays
    params = torch.ones((1, 7), device=dev())
    r = Renderer(model=m, n_samples=32, perturb=False, precision="fp16x3")
    with pytest.raises(_lib.NtxError) as e:
        r(*to_dev(ro[None], rd[None], t[None]), parameters=params, cone_scale=to_dev(cone[None])[0])
    assert e.value.code == _lib.NTX_E_UNSUPPORTED and "fp16x3" in str(e.value)


FREQS = [   # (kind, n_parameters, (pos, dir, param) n_freq_bands, arch)
    ("ParamNerf", (1, 6), (6, 2, 3), None),                                   # tuned carpet kernels
    ("ParamNerf", (2, 3), (10, 4, 1), None),                                  # only the parameter embedding differs
    ("ParamNerf", (1, 4), (0, 0, 0), None),                                   # identity features only
    ("ParamNerf", (3, 2), (7, 1, 2), None),                                   # generic family
    ("Nerf", (0, 0), (5, 3, 0), None),
    ("IPE", (1, 3), (6, 4, 2), None),                                         # mip: [sin | cos] halves of 3 * 6 features each
    ("ParamNerf", (1, 6), (4, 3, 2), dict(depth=5, width=128, skips=[2], color_depth=2)),                                # flex
    ("ParamNerf", (2, 2), (9, 2, 3), dict(depth=4, width=256, skips=[1], color_depth=1, param_depth=2, param_width=64)),  # flex with branches
]


@pytest.mark.parametrize("kind,npar,freqs,arch", FREQS)
@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
def test_mlp_forward_with_fewer_frequency_bands(kind, npar, freqs, arch, precision):
    """`n_freq_bands` is a kwarg of the reference's embeddings (layer.py:11, :27); every config uses 10 / 4 / 4.  Models with FEWER
    bands run on the same kernels -- which evaluate all of their bands -- with zero weight rows for the bands the model does not
    have (host packers only): exact, so the same gate against the oracle of the model's own dimensions, every family, both precisions."""
    if precision == "fp16x3" and arch:
        pytest.skip("the flex family is float32 only")
    model, spec, w = make_model(npar, kind, arch=arch, freqs=freqs)
    assert model.layer_table() == orc.layer_table(spec)
    model.precision = precision
    m = 1000
    pos, dirs, params = random_samples(m, sum(npar))
    if kind == "IPE":
        pos = np.concatenate([pos, np.random.default_rng(4).uniform(0, 2e-3, size=(m, 3)).astype(np.float32)], -1)   # (mean, diagonal covariance)
    color, alpha = model(tuple(to_dev(pos, dirs, params)))
    rc, ra = orc.model_forward(w, spec, pos, dirs, params, np.float64)
    err = orc.rel_linf(np.concatenate([color.cpu().numpy(), alpha.cpu().numpy()], -1), np.concatenate([rc, ra], -1))
    assert err <= (3e-5 if precision == "float32" else TOL), err


@pytest.mark.parametrize("kind,npar,freqs,arch,blur", [("ParamNerf", (1, 6), (6, 2, 3), None, None), ("ParamNerf", (2, 3), (8, 3, 2), None, 0),
                                                        ("ParamNerf", (1, 6), (4, 3, 2), dict(depth=5, width=128, skips=[2], color_depth=2), 3)])
def test_render_rays_with_fewer_frequency_bands(kind, npar, freqs, arch, blur):
    """Renderer.__call__ on such a model: the hoisted per-ray rows (dir_block: direction segment, geometry blocks) come from the same
    zero-padded stream."""
    from nerf_tex_amd.renderer import Renderer
    model, spec, w = make_model(npar, kind, dense_media=True, arch=arch, freqs=freqs)
    (ro, rd, t, cone), _, _ = camera_rays("carpet", 14, 12)
    cone = (cone * 30).astype(np.float32)
    P = sum(npar)
    params = np.random.default_rng(P + 5).uniform(0.1, 1.0, size=(1, P)).astype(np.float32)
    S = 48
    r = Renderer(model=model, n_samples=S, perturb=False, blur_idx=blur)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0])
    r.raise_if_nonfinite()
    got = rgba_of(out)
    ref = {}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        o = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.), blur, False, dtype=dt)
        ref[name] = np.concatenate([o["color_pred"][0], o["alpha_pred"][0][:, None]], -1).astype(np.float64)
    assert orc.rel_linf(got, ref["f32"]) <= TOL
    assert orc.rel_linf(got, ref["f64"]) <= TOL + orc.rel_linf(ref["f32"], ref["f64"])
