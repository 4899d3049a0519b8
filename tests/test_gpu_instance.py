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

    def __init__(self, n_params, seed=0, p_hit=0.8, p_in=0.35, n_inst=7, run_len=None, n_geo=1):
        self.rng = np.random.default_rng(seed)
        self.n_params, self.p_hit, self.p_in, self.n_inst = n_params, p_hit, p_in, n_inst
        self.run_len, self.n_geo = run_len, n_geo
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
        if self.run_len is not None:
            # What the reference's instancer really hands out (instancer.cpp:889-960): the marching samples of a ray fall into RUNS,
            # one patch instance each; direction (getDir(ray, instance)), light direction and the other appearance parameters are
            # constant along a run, the texture-mapped geometry parameters and the position vary per sample.  Runs of 1 ..
            # run_len samples, in or out of a patch as a whole; a run may be followed by one of the SAME instance (same inputs).
            seg = np.zeros((n, S), np.int64)
            for r in range(n):
                k = 0; q = 0
                while k < S:
                    ln = int(rng.integers(1, self.run_len + 1))
                    seg[r, k:k + ln] = q; k += ln; q += 1
            nseg = int(seg.max()) + 1
            per = lambda shape: rng.uniform(size=(n, nseg) + shape)
            sd = rng.normal(size=(n, nseg, 3)); sd /= np.linalg.norm(sd, axis=-1, keepdims=True)
            same = rng.uniform(size=(n, nseg)) < 0.15                    # this run repeats its predecessor's instance
            for q in range(1, nseg):
                sd[:, q][same[:, q]] = sd[:, q - 1][same[:, q]]
            app = 0.5 + 0.5 * per((1,))
            for q in range(1, nseg):
                app[:, q][same[:, q]] = app[:, q - 1][same[:, q]]
            take = lambda a: np.take_along_axis(a, seg[..., None], 1)
            rays_d_map = take(sd)
            params_map = np.repeat(parameters[:, None, :], S, axis=1).astype(np.float64)
            params_map[..., self.n_geo:] *= take(app)                                       # appearance: per run
            params_map[..., :self.n_geo] *= rng.uniform(0.5, 1.0, size=(n, S, 1))            # geometry: per sample
            inside = np.take_along_axis(rng.uniform(size=(n, nseg)) < self.p_in, seg, 1)
            dists = np.where(inside, rng.uniform(0.5, 2.0, size=(n, S)) * step_size, 0.0)
            instance_id = np.take_along_axis(rng.integers(0, self.n_inst, size=(n, nseg)), seg, 1).astype(np.int32)
            if n > 3:
                dists[0] = 0.0
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


def _render_instanced_raw(model, bufs, hit, cone, S, precision="float32", blur=-1):
    from nerf_tex_amd import _lib
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map = bufs
    n = dists.shape[0]
    dv = torch.device("cuda", 0)
    d = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), device=dv).to(dt).contiguous()
    t_ = dict(rd=d(rays_d_map), pts=d(pts), t=d(tt), dists=d(dists), cl=d(color_last.reshape(n, 3)), al=d(alpha_last.reshape(n)),
              aw=d(alpha_weight), iid=d(instance_id, torch.int32), hit=d(hit, torch.uint8), pm=d(params_map), cone=d(cone))
    col = torch.full((n, 3), float('nan'), device=dv); alp = torch.full((n,), float('nan'), device=dv)   # a ray nobody renders shows
    model.reserve(0, n)
    _lib.check(_lib.lib.ntx_render_instanced(
        model.ctx(0), t_["rd"].data_ptr(), t_["pts"].data_ptr(), t_["t"].data_ptr(), t_["dists"].data_ptr(), t_["cl"].data_ptr(),
        t_["al"].data_ptr(), t_["aw"].data_ptr(), t_["iid"].data_ptr(), t_["hit"].data_ptr(), t_["pm"].data_ptr(), t_["cone"].data_ptr(),
        n, S, blur, 0.09, 400.0, _lib.PRECISIONS[precision], _lib.f3([1, 1, 1.]), None, None, col.data_ptr(), alp.data_ptr(), None,
        torch.cuda.current_stream(dv).cuda_stream))
    torch.cuda.synchronize()
    return np.concatenate([col.cpu().numpy(), alp.cpu().numpy()[:, None]], -1)


@pytest.mark.parametrize("npar,blur,run_len,S", [((1, 6), None, 40, 600), ((1, 6), None, 5, 300), ((2, 3), 0, 24, 400), ((1, 4), None, 100, 1024),
                                                 ((2, 5), 1, 16, 300), ((1, 6), 3, 16, 200)])
def test_instance_runs_share_their_direction_features(npar, blur, run_len, S, monkeypatch):
    """The instancer fills direction and appearance parameters per (ray, patch instance) (instancer.cpp:943-960), so they are
    constant along a run of in-patch samples; instance_kernel evaluates C1's direction segment once per run (leader_rows) and the
    samples start from their run's row.  Synthetic instancer output with that structure -- runs of 1 .. run_len samples, runs that
    cross batches, repeated instances, per-sample geometry parameters; blur_idx on a geometry parameter and (last case) on an
    APPEARANCE parameter, which makes every sample its own run -- against the float64 oracle, and bit for bit against a context
    created under NERFTEX_NO_DIR_HOIST, which treats every sample as its own run."""
    model, spec, w = make_model(npar, dense_media=True)
    P = sum(npar)
    inst = FakeInstancer(P, seed=run_len + S, p_hit=0.9, p_in=0.5, run_len=run_len, n_geo=npar[0])
    n = 700 if S < 1000 else 300                       # (the float64 restatement of 700 rays x 1024 steps takes 7 s of host time)
    rng = np.random.default_rng(8)
    params = rng.uniform(0.2, 1, size=(n, P)).astype(np.float32)
    bufs = inst.get_model_input(np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), params, S, 0.002)
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map = bufs
    hit = np.zeros(n, np.uint8); hit[idxs[:, 0]] = 1
    cone = rng.uniform(1e-3, 5e-3, size=n).astype(np.float32)
    b = -1 if blur is None else blur
    got = _render_instanced_raw(model, bufs, hit, cone, S, blur=b)
    rc, ra = orc.instance_evaluate_model(w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id,
                                         hit.astype(bool), params_map, cone[:, None], blur, 0.09, 400.0, True, False, False, (1., 1., 1.),
                                         None, dtype=np.float64)
    assert orc.rel_linf(got, np.concatenate([rc, ra[:, None]], -1)) <= TOL
    monkeypatch.setenv("NERFTEX_NO_DIR_HOIST", "1")
    model2, _, _ = make_model(npar, dense_media=True)
    assert np.array_equal(_render_instanced_raw(model2, bufs, hit, cone, S, blur=b), got)
    monkeypatch.delenv("NERFTEX_NO_DIR_HOIST")
    # the runs are really there: consecutive in-patch samples mostly share their direction
    ins = dists > 0
    same = (rays_d_map[:, 1:] == rays_d_map[:, :-1]).all(-1) & ins[:, 1:] & ins[:, :-1]
    assert same.sum() > 0.5 * (ins[:, 1:] & ins[:, :-1]).sum()


@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
def test_instance_renderer_raw_noise(precision):
    """InstanceRenderer.map_model_output adds raw_noise_std * N(0,1) to the (scaled) density before the relu (renderer.py:335-337);
    drawn inside the kernel from the restated generator, keyed by (seed, ray among the proxy-hit rays, marching sample)."""
    from nerf_tex_amd.renderer import InstanceRenderer
    model, spec, w = make_model((1, 6), dense_media=True)
    inst = FakeInstancer(7, seed=13, run_len=20)
    S, n, std, seed = 120, 90, 25.0, 4711                       # the density is scaled by 400 before the noise: std of that order
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=0.09, step_size=0.002, density_scale=400.0,
                         raw_noise_std=std, precision=precision)
    rng = np.random.default_rng(3)
    ro = rng.normal(size=(1, n, 3)).astype(np.float32); rd = rng.normal(size=(1, n, 3)).astype(np.float32)
    t = np.tile(np.asarray([[1.0, 2.0]], np.float32), (1, n, 1)); t[0, 7] = np.inf
    params = rng.uniform(0.2, 1, size=(1, 7)).astype(np.float32)
    cone = rng.uniform(1e-3, 5e-3, size=(1, n, 1)).astype(np.float32)
    dv = torch.device("cuda", 0)
    d = lambda a: torch.as_tensor(a, device=dv)
    out = r(d(ro), d(rd), d(t), parameters=d(params), cone_scale=d(cone), seed=seed)
    r.raise_if_nonfinite()
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map, hit = inst.last
    keep = np.isfinite(t[0, :, 0])
    noise = std * orc.noise_normals(int(keep.sum()), S, seed, dtype=np.float64)
    args = (w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, hit, params_map, cone[0][keep], None,
            0.09, 400.0, True, False, False, (1., 1., 1.), None)
    rc, ra = orc.instance_evaluate_model(*args, dtype=np.float64, noise=noise)
    got = np.concatenate([out["color_pred"][0].cpu().numpy()[keep], out["alpha_pred"][0].cpu().numpy()[keep][:, None]], -1)
    assert orc.rel_linf(got, np.concatenate([rc, ra[:, None]], -1)) <= TOL
    rc0, ra0 = orc.instance_evaluate_model(*args, dtype=np.float64)
    assert orc.rel_linf(got, np.concatenate([rc0, ra0[:, None]], -1)) > 10 * TOL


def test_instance_rays_longer_than_the_index_window():
    """A ray's compacted index list lives in the context's global scratch and is read through a 1024-entry window in LDS; rays
    with up to 4096 in-patch samples slide it (also at the tail, and while groups of runs look ahead)."""
    model, spec, w = make_model((1, 6), dense_media=True)
    S, n = 4096, 24
    inst = FakeInstancer(7, seed=77, p_hit=1.0, p_in=0.8, run_len=60)
    rng = np.random.default_rng(5)
    params = rng.uniform(0.2, 1, size=(n, 7)).astype(np.float32)
    bufs = list(inst.get_model_input(np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), params, S, 0.002))
    bufs[3][1] = np.abs(bufs[3][1]) + 1e-4                      # one ray with all 4096 samples inside
    bufs[3][2, 1100:] = 0.0                                     # one that just crosses the window
    bufs[3] = bufs[3] * 0.02                                    # thin media: the far samples still count
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map = bufs
    hit = np.zeros(n, np.uint8); hit[idxs[:, 0]] = 1
    cone = rng.uniform(1e-3, 5e-3, size=n).astype(np.float32)
    got = _render_instanced_raw(model, bufs, hit, cone, S)
    assert (dists > 0).sum(-1).max() == 4096 and ((dists > 0).sum(-1) > 1024).sum() > 10
    rc, ra = orc.instance_evaluate_model(w, spec, rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id,
                                         hit.astype(bool), params_map, cone[:, None], None, 0.09, 400.0, True, False, False, (1., 1., 1.),
                                         None, dtype=np.float64)
    assert orc.rel_linf(got, np.concatenate([rc, ra[:, None]], -1)) <= TOL
    assert np.array_equal(_render_instanced_raw(model, bufs, hit, cone, S), got)


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
        col = torch.full((k, 3), float('nan'), device=dv); alp = torch.full((k,), float('nan'), device=dv)   # a ray nobody renders shows
        _lib.check(_lib.lib.ntx_render_instanced(
            model.ctx(0), t_["rd"].data_ptr(), t_["pts"].data_ptr(), t_["t"].data_ptr(), t_["dists"].data_ptr(), t_["cl"].data_ptr(),
            t_["al"].data_ptr(), t_["aw"].data_ptr(), t_["iid"].data_ptr(), t_["hit"].data_ptr(), t_["pm"].data_ptr(), t_["cone"].data_ptr(),
            k, S, -1, 0.09, 400.0, flags, _lib.f3([1, 1, 1.]), None, None, col.data_ptr(), alp.data_ptr(), None,
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


def test_bundles_of_rays_give_the_single_ray_bits(monkeypatch):
    """The float32 instance kernel claims CHUNKS of the cost order (4 | 2 | 1 rays; inst_order_kernel says where) and compiles them
    into one execution list: whole batches of each ray, packed batches of tails that close inside the bundle or stay open across
    bundles, groups of run rows that span rays.  That only happens with many more rays than waves, so: 40 960 short rays (S = 64,
    ~13 in-patch samples each, i.e. mostly tails; a few rays with all 64 inside, rays with exactly 32, empty and un-hit rays) --
    the image bit for bit what single-ray claims give (NERFTEX_DEBUG_RUNS bit 3), under other chunk thresholds, under a permutation of
    the rays and with the run rows off; 400 of the rays against the float64 oracle."""
    from nerf_tex_amd import _lib
    model, spec, w = make_model((1, 6), dense_media=True)
    n, S = 40960, 64
    inst = FakeInstancer(7, seed=31, p_hit=0.97, p_in=0.2, run_len=8)
    rng = np.random.default_rng(12)
    params = rng.uniform(0.2, 1, size=(n, 7)).astype(np.float32)
    bufs = list(inst.get_model_input(np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), params, S, 0.002))
    dists = bufs[3]
    full = rng.choice(n, 60, replace=False)
    dists[full] = np.abs(dists[full]) + 1e-4                                       # the heavy tail: every marching sample inside
    half = rng.choice(n, 200, replace=False)
    dists[half, :32] = np.abs(dists[half, :32]) + 1e-4; dists[half, 32:] = 0.0     # exactly one whole batch and no tail
    rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map = bufs
    hit = np.zeros(n, np.uint8); hit[idxs[:, 0]] = 1
    cone = rng.uniform(1e-3, 5e-3, size=n).astype(np.float32)
    dv = torch.device("cuda", 0)
    model.reserve(0, n)

    def render(order):
        d = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a[order]), device=dv).to(dt).contiguous()
        t_ = dict(rd=d(rays_d_map), pts=d(pts), t=d(tt), dists=d(dists), cl=d(color_last.reshape(n, 3)), al=d(alpha_last.reshape(n)),
                  aw=d(alpha_weight), iid=d(instance_id, torch.int32), hit=d(hit, torch.uint8), pm=d(params_map), cone=d(cone))
        k = len(order)
        col = torch.full((k, 3), float('nan'), device=dv); alp = torch.full((k,), float('nan'), device=dv)   # a ray nobody renders shows
        _lib.check(_lib.lib.ntx_render_instanced(
            model.ctx(0), t_["rd"].data_ptr(), t_["pts"].data_ptr(), t_["t"].data_ptr(), t_["dists"].data_ptr(), t_["cl"].data_ptr(),
            t_["al"].data_ptr(), t_["aw"].data_ptr(), t_["iid"].data_ptr(), t_["hit"].data_ptr(), t_["pm"].data_ptr(), t_["cone"].data_ptr(),
            k, S, -1, 0.09, 400.0, 0, _lib.f3([1, 1, 1.]), None, None, col.data_ptr(), alp.data_ptr(), None,
            torch.cuda.current_stream(dv).cuda_stream))
        torch.cuda.synchronize()
        return np.concatenate([col.cpu().numpy(), alp.cpu().numpy()[:, None]], -1)

    ident = np.arange(n)
    base = render(ident)                                                           # chunks of 4 | 2 | 1
    assert np.array_equal(render(ident), base)
    for knob in (9,                                # single rays throughout
                 1 | (1 << 8) | (1 << 16),         # fours and pairs almost to the end
                 1 | (20 << 8) | (10 << 16),       # pairs only, single rays early
                 1 | 4,                            # groups of one batch
                 0, 3):                            # run rows off / flags computed, per-sample rows
        monkeypatch.setenv("NERFTEX_DEBUG_RUNS", str(knob))
        assert np.array_equal(render(ident), base), knob
    monkeypatch.delenv("NERFTEX_DEBUG_RUNS")
    perm = np.random.default_rng(2).permutation(n)
    assert np.array_equal(render(perm), base[perm])
    counts = (dists > 0).sum(-1)
    assert (counts[hit == 1] == 0).any() and (hit == 0).sum() > 100 and (counts == 64).sum() >= 60 and ((counts > 0) & (counts < 32)).mean() > 0.8
    sub = np.concatenate([full[:20], half[:20], rng.choice(n, 360, replace=False)])
    take = lambda a: a[sub]
    rc, ra = orc.instance_evaluate_model(w, spec, take(rays_d_map), take(pts), take(tt), take(dists), take(color_last), take(alpha_last),
                                         take(alpha_weight), take(instance_id), take(hit).astype(bool), take(params_map), take(cone)[:, None],
                                         None, 0.09, 400.0, True, False, False, (1., 1., 1.), None, dtype=np.float64)
    assert orc.rel_linf(base[sub], np.concatenate([rc, ra[:, None]], -1)) <= TOL


@pytest.mark.parametrize("kind,npar,arch,S", [("ParamNerf", (1, 4), None, 8), ("ParamNerf", (1, 4), None, 72), ("Nerf", (0, 0), None, 40),
                                              ("ParamNerf", (2, 5), None, 40), ("ParamNerf", (1, 2), dict(depth=3, width=64, skips=(1,)), 40)])
def test_chunked_hand_out_renders_every_ray_once(kind, npar, arch, S, monkeypatch):
    """inst_order_kernel cuts the cost order into chunks (single rays | pairs | fours | pairs | single rays; the ranks depend on the
    ray count, the number of waves and the cost histogram) and instance_kernel maps claim c to a chunk in closed form: at ray
    counts around every boundary of that table (1024 waves: 3 and 6 rays per wave, +-1, odd counts) every ray is written (the
    outputs start as NaN), and the image is bit for bit what single-ray claims give.  A tuned family at two sample counts (tails
    only / whole batches + tails), plain Nerf and the flex family (no run rows), the generic family."""
    model, spec, w = make_model(npar, kind=kind, dense_media=True, arch=arch)
    P = sum(npar)
    rng = np.random.default_rng(S)
    n_max = 20001
    inst = FakeInstancer(P, seed=S, p_hit=0.95, p_in=0.3)
    params = rng.uniform(0.2, 1, size=(n_max, P)).astype(np.float32)
    bufs = inst.get_model_input(np.zeros((n_max, 3), np.float32), np.zeros((n_max, 3), np.float32), params, S, 0.002)
    hit = np.zeros(n_max, np.uint8); hit[bufs[8][:, 0]] = 1
    cone = rng.uniform(1e-3, 5e-3, size=n_max).astype(np.float32)
    counts = (1, 2, 3, 5, 1023, 2047, 2048, 2049, 3071, 3073, 6143, 6144, 6147, 8190, 12289, 16383, 20001)
    for n in counts if (kind, npar, arch) == ("ParamNerf", (1, 4), None) else (3, 3073, 6147, 20001):
        sub = [b[:n] for b in bufs]
        monkeypatch.delenv("NERFTEX_DEBUG_RUNS", raising=False)
        got = _render_instanced_raw(model, sub, hit[:n], cone[:n], S)
        assert np.isfinite(got).all(), n
        assert np.all(got[hit[:n] == 0] == 0.0)
        monkeypatch.setenv("NERFTEX_DEBUG_RUNS", "9")
        assert np.array_equal(_render_instanced_raw(model, sub, hit[:n], cone[:n], S), got), n
    monkeypatch.delenv("NERFTEX_DEBUG_RUNS")
