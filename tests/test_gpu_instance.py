"""InstanceRenderer tail against the oracle, with a synthetic instancer standing in
for the reference's Embree one (its output contract: instancer.pyx:38-54).  `-m gpu`."""

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from tests.common import TOL, make_model

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# The instanced path multiplies the raw density by density_scale * alpha_weight (400 here) and by dists / patch_scale before
# the exponential (renderer.py:300, 339), which amplifies whatever error the density carries; the fp16x3 split is accurate
# enough (2^-22 per product) to stay inside the same 1e-4 gate as float32.
TOL_FP16X3_INSTANCED = 1e-4


class FakeInstancer:
    """Random in-patch segments per ray: exactly the buffer shapes/dtypes of instancer.get_model_input."""

    def __init__(self, n_params, seed=0, p_hit=0.8, p_in=0.35, n_inst=7):
        self.rng = np.random.default_rng(seed)
        self.n_params, self.p_hit, self.p_in, self.n_inst = n_params, p_hit, p_in, n_inst
        self.last = None

    def n_instances(self):
        return self.n_inst

    def get_model_input(self, rays_o, rays_d, parameters, n_samples, step_size):
        rng = self.rng
        n, S = rays_o.shape[0], n_samples
        rays_d_map = rng.normal(size=(n, S, 3)); rays_d_map /= np.linalg.norm(rays_d_map, axis=-1, keepdims=True)
        pts = rng.uniform(-1.2, 1.2, size=(n, S, 3))
        t = np.sort(rng.uniform(2, 8, size=(n, S)), -1)
        inside = rng.uniform(size=(n, S)) < self.p_in
        dists = np.where(inside, rng.uniform(0.5, 2.0, size=(n, S)) * step_size, 0.0)
        dists[rng.uniform(size=(n, S)) < 0.05] = -step_size          # "outside" can also be negative
        hit = rng.uniform(size=n) < self.p_hit
        if n > 3:
            hit[0] = True; inside[0] = False; dists[0] = 0.0          # hit ray with no in-patch sample
            hit[1] = False
        color_last = rng.uniform(0, 1, size=(n, 1, 3)); alpha_last = (rng.uniform(size=(n, 1)) < 0.5).astype(np.float64)
        alpha_weight = 1.0 / rng.integers(1, 4, size=(n, S))
        instance_id = rng.integers(0, self.n_inst, size=(n, S)).astype(np.int32)
        params_map = np.repeat(parameters[:, None, :], S, axis=1) * rng.uniform(0.5, 1.0, size=(n, S, 1))
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        out = (f(rays_d_map), f(pts), f(t), f(dists), f(color_last), f(alpha_last), f(alpha_weight), instance_id,
               np.nonzero(hit)[0][:, None], f(params_map))
        self.last = out + (hit,)
        return out


@pytest.mark.parametrize("npar,blur", [((1, 6), None), ((2, 3), 0), ((1, 4), None), ((2, 5), 1)])   # (2, 5): the generic family
@pytest.mark.parametrize("S", [40, 200])
@pytest.mark.parametrize("opts", [dict(), dict(composite_bkgd=True, map_exr=True), dict(density_reweighting=False, density_scale=30.0),
                                  dict(false_color=True)])
@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
def test_instance_renderer(npar, blur, S, opts, precision):
    from nerf_tex_amd.renderer import InstanceRenderer
    opts = dict(opts, precision=precision)
    bk = opts.pop("composite_bkgd", False)
    model, spec, w = make_model(npar, dense_media=True)
    P = sum(npar)
    inst = FakeInstancer(P, seed=S + P)
    patch_scale, step = 0.09, 0.002
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=patch_scale, step_size=step, blur_idx=blur,
                         render_chunk=10_000, density_scale=opts.pop("density_scale", 400.0), **opts)
    rng = np.random.default_rng(1)
    n = 61
    ro = rng.normal(size=(1, n, 3)).astype(np.float32); rd = rng.normal(size=(1, n, 3)).astype(np.float32)
    t = np.tile(np.asarray([[1.0, 2.0]], np.float32), (1, n, 1)); t[0, 5] = np.inf      # one ray culled by the proxy
    params = rng.uniform(0.2, 1, size=(1, P)).astype(np.float32)
    cone = rng.uniform(1e-3, 5e-3, size=(1, n, 1)).astype(np.float32)
    dv = torch.device("cuda", 0)
    d = lambda a: torch.as_tensor(a, device=dv)
    out = r(d(ro), d(rd), d(t), parameters=d(params), cone_scale=d(cone), composite_bkgd=bk, bkgd_color=[.3, .6, .9])
    r.raise_if_nonfinite()
    # oracle on the very buffers the fake instancer handed out (it saw the n-1 rays that survive the t cull)
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map, hit = inst.last
    keep = np.isfinite(t[0, :, 0])
    rc, ra = orc.instance_evaluate_model(w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id,
                                         hit, params_map, cone[0][keep], blur, patch_scale, r.density_scale, r.density_reweighting,
                                         r.map_exr, bk, (.3, .6, .9), r.instance_color, dtype=np.float64)
    want_c = np.zeros((n, 3)); want_a = np.zeros(n)
    want_c[keep] = rc; want_a[keep] = ra
    if bk:
        want_c[~keep] = (.3, .6, .9)                                  # renderer.py:85-86: only proxy-culled rays
    got = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
    want = np.concatenate([want_c, want_a[:, None]], -1)
    assert orc.rel_linf(got, want) <= (TOL if precision == "float32" else TOL_FP16X3_INSTANCED)
    kept = np.nonzero(keep)[0]
    assert np.all(got[kept[~hit]] == 0.0)                             # un-hit rays stay 0, even with background (:313-314)
    assert float(want_a.max()) > 0.3


def test_instance_renderer_fp16x3_many_rays_lockstep():
    """More rays than waves and very uneven rays (0 .. 300 in-patch samples): the fp16x3 instanced kernel runs its four
    waves per workgroup in lockstep rounds while each wave marches its own ray; result = the float32 kernel's within
    the tolerance, un-hit and sample-less rays included."""
    from nerf_tex_amd.renderer import InstanceRenderer
    model, spec, w = make_model((1, 6), dense_media=True)

    class Uneven(FakeInstancer):
        def get_model_input(self, rays_o, rays_d, parameters, n_samples, step_size):
            out = list(super().get_model_input(rays_o, rays_d, parameters, n_samples, step_size))
            dists = out[3]
            n = dists.shape[0]
            keep = self.rng.integers(0, n_samples + 1, size=n)        # ray r keeps only its first keep[r] samples' patches
            dists[np.arange(n_samples)[None, :] >= keep[:, None]] = 0.0
            out[3] = dists
            self.last = tuple(out) + (self.last[-1],)
            return tuple(out)

    n, S = 3000, 300
    rng = np.random.default_rng(2)
    ro = rng.normal(size=(1, n, 3)).astype(np.float32); rd = rng.normal(size=(1, n, 3)).astype(np.float32)
    t = np.tile(np.asarray([[1.0, 2.0]], np.float32), (1, n, 1))
    params = rng.uniform(0.2, 1, size=(1, 7)).astype(np.float32)
    cone = rng.uniform(1e-3, 5e-3, size=(1, n, 1)).astype(np.float32)
    dv = torch.device("cuda", 0)
    d = lambda a: torch.as_tensor(a, device=dv)
    res = {}
    for prec in ("float32", "fp16x3"):
        inst = Uneven(7, seed=11, p_in=0.9)
        r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=0.09, step_size=0.002, density_scale=400.0,
                             render_chunk=100_000, precision=prec)
        out = r(d(ro), d(rd), d(t), parameters=d(params), cone_scale=d(cone))
        r.raise_if_nonfinite()
        res[prec] = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
    assert orc.rel_linf(res["fp16x3"], res["float32"]) <= TOL_FP16X3_INSTANCED
    assert not np.array_equal(res["fp16x3"], res["float32"])
    hit = inst.last[-1]
    assert np.all(res["fp16x3"][~hit] == 0.0)


@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
def test_instance_renderer_plain_nerf(precision):
    """The instanced tail with a plain Nerf model (no material parameters: params_map is [n,S,0])."""
    from nerf_tex_amd.renderer import InstanceRenderer
    model, spec, w = make_model((0, 0), kind="Nerf", dense_media=True)
    inst = FakeInstancer(0, seed=5)
    S, n = 70, 45
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=0.09, step_size=0.002, density_scale=400.0,
                         precision=precision)
    rng = np.random.default_rng(3)
    ro = rng.normal(size=(1, n, 3)).astype(np.float32); rd = rng.normal(size=(1, n, 3)).astype(np.float32)
    t = np.tile(np.asarray([[1.0, 2.0]], np.float32), (1, n, 1))
    params = np.zeros((1, 0), np.float32)
    cone = rng.uniform(1e-3, 5e-3, size=(1, n, 1)).astype(np.float32)
    dv = torch.device("cuda", 0)
    d = lambda a: torch.as_tensor(a, device=dv)
    out = r(d(ro), d(rd), d(t), parameters=d(params), cone_scale=d(cone))
    r.raise_if_nonfinite()
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map, hit = inst.last
    rc, ra = orc.instance_evaluate_model(w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id,
                                         hit, params_map, cone[0], None, 0.09, r.density_scale, r.density_reweighting,
                                         r.map_exr, False, (1., 1., 1.), r.instance_color, dtype=np.float64)
    got = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
    want = np.concatenate([rc, ra[:, None]], -1)
    assert orc.rel_linf(got, want) <= TOL
    assert float(np.max(ra)) > 0.3


@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
def test_packed_tails_do_not_depend_on_the_company(precision):
    """The instanced kernels evaluate the tails (count % 32 samples) of successive rays of a wave in ONE packed batch; which
    rays meet there depends on the dynamic ray hand-out.  The image must not: every ray's result is bit-identical from run
    to run, under any permutation of the rays, and when the ray is rendered on its own."""
    from nerf_tex_amd import _lib
    model, spec, w = make_model((1, 6), dense_media=True)
    inst = FakeInstancer(7, seed=21, p_hit=0.9, p_in=0.3)
    n, S = 1500, 150                                      # in-patch counts ~45 +- 6: every kind of tail
    rng = np.random.default_rng(4)
    params = rng.uniform(0.2, 1, size=(n, 7)).astype(np.float32)
    bufs = inst.get_model_input(np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), params, S, 0.002)
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map = bufs
    hit = np.zeros(n, np.uint8); hit[idxs[:, 0]] = 1
    cone = rng.uniform(1e-3, 5e-3, size=n).astype(np.float32)
    dv = torch.device("cuda", 0)
    flags = _lib.PRECISIONS[precision]

    def render(order):
        d = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a[order]), device=dv).to(dt).contiguous()
        t_ = dict(rd=d(rays_d_map), pts=d(pts), t=d(tt), dists=d(dists), cl=d(color_last.reshape(n, 3)), al=d(alpha_last.reshape(n)),
                  aw=d(alpha_weight), iid=d(instance_id, torch.int32), hit=d(hit, torch.uint8), pm=d(params_map), cone=d(cone))
        k = len(order)
        col = torch.empty((k, 3), device=dv); alp = torch.empty((k,), device=dv)
        _lib.check(_lib.lib.ntx_render_instanced(
            model.ctx(0), t_["rd"].data_ptr(), t_["pts"].data_ptr(), t_["t"].data_ptr(), t_["dists"].data_ptr(), t_["cl"].data_ptr(),
            t_["al"].data_ptr(), t_["aw"].data_ptr(), t_["iid"].data_ptr(), t_["hit"].data_ptr(), t_["pm"].data_ptr(), t_["cone"].data_ptr(),
            k, S, -1, 0.09, 400.0, flags, _lib.f3([1, 1, 1.]), None, col.data_ptr(), alp.data_ptr(), None,
            torch.cuda.current_stream(dv).cuda_stream))
        torch.cuda.synchronize()
        return np.concatenate([col.cpu().numpy(), alp.cpu().numpy()[:, None]], -1)

    ident = np.arange(n)
    base = render(ident)
    for _ in range(3):
        assert np.array_equal(render(ident), base)                                   # run to run
    perm = np.random.default_rng(9).permutation(n)
    assert np.array_equal(render(perm), base[perm])                                  # any neighbours
    some = np.asarray([5, 17, 333, 1499])
    for r in some:
        assert np.array_equal(render(np.asarray([r]))[0], base[r])                   # alone
    counts = (dists > 0).sum(-1)
    assert len(set((counts[hit == 1] % 32).tolist())) > 20 and (counts[hit == 1] >= 32).any()
    rc, ra = orc.instance_evaluate_model(w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id,
                                         hit.astype(bool), params_map, cone[:, None], None, 0.09, 400.0, True, False, False, (1., 1., 1.),
                                         None, dtype=np.float64)
    assert orc.rel_linf(base, np.concatenate([rc, ra[:, None]], -1)) <= TOL
