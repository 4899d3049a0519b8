"""HIP paths of the C ABI that the family/parity suites do not reach: Frustum rays, the AABB's IEEE corner cases on
the kernel, the in-kernel stratified jitter, cone-filter conditioning at S = 128, full-size runs of the deep-sampling
families, error paths that must write nothing, per-call precision on a shared context, strided ray generation and the
RCCL gather.  `-m gpu`; everything goes through the C ABI."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from tests.common import TOL, camera_rays, make_model

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda", 0)


def to_dev(*arrs):
    return [torch.as_tensor(a, device=dev()) for a in arrs]


def rgba_of(out):
    return np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)


def rgba_ref(ref, b=None):
    c, a = (ref["color_pred"], ref["alpha_pred"]) if b is None else (ref["color_pred"][b], ref["alpha_pred"][b])
    return np.concatenate([c, a[..., None]], -1)


# ---------------------------------------------------------------------------------------------------------------
# ray_sampler.Frustum (ray_sampler.py:6-21): un-normalised rays_d, constant [near, far]
# ---------------------------------------------------------------------------------------------------------------
def test_frustum_rays_and_render():
    """mode 1 of ntx_generate_rays against the oracle, then the render of those rays: `pts` use the UN-normalised
    rays_d (renderer.py:114), the model gets rays_d / |rays_d| (:98), dists are scaled by |rays_d| (:180)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.ray_sampler import Frustum
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    h, w = 18, 26
    c2w = orc.look_at(fam["cam"], dtype=np.float32)
    focal = orc.focal_from_angle(w, fam["angle"])
    near, far = 4.5, 8.0
    o, d, t, cone = Frustum(h, w, focal, near, far)((0, h * w), c2w)
    ro, rd, rt, rcone = orc.frustum_rays(orc.full_pixels(h, w), h, w, focal, c2w, near, far, np.float64)
    assert np.max(np.abs(o.cpu().numpy() - ro)) <= 1e-6 and np.max(np.abs(d.cpu().numpy() - rd)) <= 1e-6
    assert np.array_equal(t.cpu().numpy(), rt.astype(np.float32))
    assert np.max(np.abs(cone.cpu().numpy() - rcone) / rcone) <= 1e-5
    norms = np.linalg.norm(d.cpu().numpy(), axis=-1)
    assert norms.min() >= 1.0 and norms.max() > 1.01                          # genuinely un-normalised away from the axis
    model, spec, wts = make_model((1, 6), dense_media=True)
    params = np.asarray([fam["params"]], np.float32)
    S = 64
    out = Renderer(model=model, n_samples=S, perturb=False)(o[None], d[None], t[None], parameters=to_dev(params)[0], cone_scale=cone[None])
    f32 = lambda x: x.cpu().numpy()
    ref = orc.renderer_call(wts, spec, f32(o)[None], f32(d)[None], f32(t)[None], params, f32(cone)[None], S, dtype=np.float64)
    assert orc.rel_linf(rgba_of(out), rgba_ref(ref, 0)) <= TOL
    # the same rays with rays_d normalised by the caller place the samples elsewhere: the distinction is observable
    dn = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    out_n = Renderer(model=model, n_samples=S, perturb=False)(o[None], dn[None], t[None], parameters=to_dev(params)[0], cone_scale=cone[None])
    assert orc.rel_linf(rgba_of(out_n), rgba_ref(ref, 0)) > 10 * TOL


# ---------------------------------------------------------------------------------------------------------------
# proxy.AABB (proxy.py:13-35) IEEE corner cases ON THE KERNEL
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("origin", [(0.3, -0.2, 5.0), (-1.5, 0.25, 5.0), (1.5, 1.5, 4.0), (0.0, 0.0, 1.0)])
def test_aabb_ieee_corners_on_kernel(origin):
    """An axis-aligned camera with odd image sizes: the centre column has d_x == 0 and the centre row d_y == 0 exactly, so
    1/d = +-inf; with the origin ON a slab plane (b - o) * inf is 0 * inf = NaN, and the reference's tf.where / reduce_max
    chain decides what a NaN does (proxy.py:22-33).  The kernel's hit mask must be bit-equal to the float32 restatement,
    and so must t on the special rays."""
    from nerf_tex_amd.proxy import AABB
    from nerf_tex_amd.ray_sampler import Proxy
    h, w = 33, 41
    b0, b1 = (-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = origin
    focal = np.float32(30.0)
    o, d, t, cone = Proxy(h, w, focal, AABB(b0, b1))((0, h * w), c2w)
    d, t = d.cpu().numpy(), t.cpu().numpy()
    with np.errstate(all="ignore"):
        ro, rd, rt, _ = orc.proxy_rays(orc.full_pixels(h, w), h, w, focal, c2w, b0, b1, np.float32)
    special = (rd[:, 0] == 0) | (rd[:, 1] == 0)
    assert special.sum() == h + w - 1 and np.array_equal((d[:, 0] == 0) | (d[:, 1] == 0), special)
    assert np.array_equal(d[special], rd[special])
    hit_k, hit_r = t[:, 0] != np.inf, rt[:, 0] != np.inf                       # renderer.py:58: NaN counts as a hit
    assert np.array_equal(hit_k[special], hit_r[special])
    assert np.array_equal(t[special], rt[special], equal_nan=True)             # incl. which of them are NaN
    # all other rays: same mask except where float32 rounding of a grazing ray may fall either way
    diff = hit_k != hit_r
    if diff.any():
        with np.errstate(all="ignore"):
            _, _, t64, _ = orc.proxy_rays(orc.full_pixels(h, w), h, w, focal, c2w, b0, b1, np.float64)
        span = np.where(np.isfinite(t64[:, 0]), t64[:, 1] - t64[:, 0], 0.0)
        assert np.all(np.abs(span[diff]) < 1e-4)
    assert diff.mean() < 0.01


# ---------------------------------------------------------------------------------------------------------------
# stratified jitter inside the kernel (renderer.py:106-111)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("npts", [2, 3, 64, 65, 129])
def test_sample_depths_match_the_restated_generator(npts):
    """ntx_sample_depths == the oracle's restatement of renderer.py:101-111 with the Philox4x32-10 draws (the generator
    itself is pinned to Random123's known-answer vectors in tests/test_oracle.py): bit for bit in float32."""
    from nerf_tex_amd.renderer import Renderer
    rng = np.random.default_rng(npts)
    n = 1000
    t0 = rng.uniform(1, 5, size=n).astype(np.float32)
    t = np.stack([t0, t0 + rng.uniform(0.1, 4, size=n).astype(np.float32)], -1)
    seed = 0x1234_5678_9ABC_DEF0 + npts
    z_lin = Renderer.sample_depths(to_dev(t)[0], npts).cpu().numpy()
    assert np.array_equal(z_lin, orc.z_values(t, npts, np.float32))
    z = Renderer.sample_depths(to_dev(t)[0], npts, perturb=True, seed=seed).cpu().numpy()
    assert np.array_equal(z, orc.z_values_perturbed(t, npts, seed, np.float32))
    # properties of renderer.py:107-111: every depth inside its own stratum, hence sorted, ends inside [t0, t1]
    lo, up = orc.jitter_bounds(z_lin)
    assert np.all(z >= lo) and np.all(z <= up) and np.all(np.diff(z, axis=-1) >= 0)
    u = (z - lo) / np.maximum(up - lo, 1e-30)
    assert abs(float(u.mean()) - 0.5) < 0.02 and float(u.std()) > 0.25          # uniform, not degenerate
    z2 = Renderer.sample_depths(to_dev(t)[0], npts, perturb=True, seed=seed + 1).cpu().numpy()
    assert not np.array_equal(z, z2)


@pytest.mark.parametrize("family,S,precision", [("carpet", 64, "float32"), ("grass", 100, "float32"), ("grass_filtered", 48, "float32"),
                                                ("carpet", 64, "fp16x3"), ("grass_filtered", 33, "fp16x3")])
def test_render_with_in_kernel_jitter(family, S, precision):
    """perturb=True (the reference's default, renderer.py:34): the fused kernel draws the jitter itself.  Checked against the
    oracle evaluated on the restated depths of the same (seed, ray, sample)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], dense_media=True)
    (ro, rd, t, cone), _, _ = camera_rays(family, 14, 18)
    n = ro.shape[0]
    params = np.asarray([fam["params"]], np.float32)
    seed = 987654321012345
    r = Renderer(model=model, n_samples=S, perturb=True, blur_idx=fam["blur_idx"], precision=precision)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0], seed=seed)
    r.raise_if_nonfinite()
    hit = np.isfinite(t[:, 0])
    assert hit.any() and (~hit).any()
    tz = np.where(np.isfinite(t), t, 0).astype(np.float32)
    z = orc.z_values_perturbed(tz, S, seed, np.float32)                       # ray index = index within the call
    ref = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, hit.sum(), 0), cone[hit], S, False, (1, 1, 1.),
                          fam["blur_idx"], z_override=z[hit], dtype=np.float64)
    got = rgba_of(out)
    assert orc.rel_linf(got[hit], rgba_ref(ref)) <= TOL
    assert np.all(got[~hit] == 0)
    # and it is not the un-jittered image
    plain = Renderer(model=model, n_samples=S, perturb=False, blur_idx=fam["blur_idx"], precision=precision)(
        *to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0])
    assert orc.rel_linf(rgba_of(plain)[hit], rgba_ref(ref)) > 10 * TOL
    # calls without `seed` draw from the renderer's private generator, seeded once from numpy's global state (main.py:30 seeds that
    # from the config) and never advancing it: reproducible run to run, a new seed per call, np.random untouched
    mk = lambda: Renderer(model=model, n_samples=S, perturb=True, blur_idx=fam["blur_idx"], precision=precision)
    args = (to_dev(ro[None], rd[None], t[None]), dict(parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0]))
    np.random.seed(5)
    r1 = mk()
    a = rgba_of(r1(*args[0], **args[1])); s1 = r1._last_seed
    a2 = rgba_of(r1(*args[0], **args[1])); s2 = r1._last_seed
    state = np.random.get_state()[1].copy()
    np.random.seed(5)
    assert np.array_equal(np.random.get_state()[1], state)                  # the global stream was not consumed
    r2 = mk()
    b = rgba_of(r2(*args[0], **args[1]))
    assert s1 == r2._last_seed and s1 != s2 and np.array_equal(a, b) and not np.array_equal(a, got) and not np.array_equal(a, a2)


@pytest.mark.parametrize("family,S,precision,perturb", [("carpet", 64, "float32", False), ("grass_filtered", 40, "float32", True),
                                                        ("fur", 33, "fp16x3", False)])
def test_render_with_raw_noise(family, S, precision, perturb):
    """raw_noise_std > 0 (renderer.py:190-192; configs/config_grass_filtered_train.py:99): sigma += std * N(0,1) before the relu,
    drawn inside the kernel.  Checked against the oracle fed with the restated draws of the same (seed, ray, sample) -- Box-Muller
    of two Philox words, counter word 3 = 1 -- on top of the restated jitter when perturb is on."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], dense_media=True)
    (ro, rd, t, cone), _, _ = camera_rays(family, 12, 16)
    n = ro.shape[0]
    params = np.asarray([fam["params"]], np.float32)
    seed, std = 20240929, 0.75
    kw = dict(parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0])
    r = Renderer(model=model, n_samples=S, perturb=perturb, raw_noise_std=std, blur_idx=fam["blur_idx"], precision=precision)
    out = r(*to_dev(ro[None], rd[None], t[None]), seed=seed, **kw)
    r.raise_if_nonfinite()
    hit = np.isfinite(t[:, 0])
    tz = np.where(np.isfinite(t), t, 0).astype(np.float32)
    z = orc.z_values_perturbed(tz, S, seed, np.float32)[hit] if perturb else None
    noise = std * orc.noise_normals(n, S, seed, dtype=np.float64)[hit]         # ray index = index within the call
    ref = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(params, hit.sum(), 0), cone[hit], S, False, (1, 1, 1.),
                          fam["blur_idx"], z_override=z, dtype=np.float64, noise=noise)
    got = rgba_of(out)
    assert orc.rel_linf(got[hit], rgba_ref(ref)) <= TOL
    assert np.all(got[~hit] == 0)
    quiet = Renderer(model=model, n_samples=S, perturb=perturb, blur_idx=fam["blur_idx"], precision=precision)(
        *to_dev(ro[None], rd[None], t[None]), seed=seed, **kw)
    assert orc.rel_linf(rgba_of(quiet)[hit], rgba_ref(ref)) > 10 * TOL      # the noise is really in
    again = r(*to_dev(ro[None], rd[None], t[None]), seed=seed, **kw)
    other = r(*to_dev(ro[None], rd[None], t[None]), seed=seed + 1, **kw)
    assert np.array_equal(rgba_of(again), got) and not np.array_equal(rgba_of(other), got)


def test_jitter_and_noise_are_keyed_by_the_global_ray():
    """the generators' counter used to be the ray index WITHIN THE CALL, so a chunked or sharded image repeated its
    pattern in every part and could not be reproduced on one GPU.  With the ray index map of ntx_render_opts (Renderer: ray_index)
    the image is bit-identical however it is split: contiguous chunks, and the interleaved pixel sets of a 3-rank shard map."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.dist import ShardMap
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    model, spec, w = make_model((1, 6), dense_media=True)
    H, W, S = 20, 26, 40
    (ro, rd, t, cone), _, _ = camera_rays("carpet", H, W)
    n = H * W
    params = to_dev(np.asarray([fam["params"]], np.float32))[0]
    r = Renderer(model=model, n_samples=S, perturb=True, raw_noise_std=0.4)
    sub = lambda idx, **kw: rgba_of(r(*to_dev(ro[None, idx], rd[None, idx], t[None, idx]), parameters=params,
                                      cone_scale=to_dev(cone[None, idx])[0], seed=99, **kw))
    whole = sub(np.arange(n))
    parts = np.concatenate([sub(np.arange(k0, min(k0 + 173, n)), ray_index=(k0, 173, 173)) for k0 in range(0, n, 173)])
    assert np.array_equal(parts, whole)
    assert not np.array_equal(np.concatenate([sub(np.arange(k0, min(k0 + 173, n))) for k0 in range(0, n, 173)]), whole)
    for run in (W, 7, None):                               # rows round-robin, ragged runs, bands
        shard = ShardMap(n, 3, run)
        img = np.zeros_like(whole)
        for rank in range(3):
            pix = shard.local_pixels(rank)
            img[pix] = sub(pix, ray_index=shard.ray_index(rank))
        assert np.array_equal(img, whole), run
    # the depths alone, against the restatement of the same map
    from nerf_tex_amd.renderer import Renderer as R
    shard = ShardMap(n, 4, 5)
    tz = np.where(np.isfinite(t), t, 1).astype(np.float32)
    for rank in range(4):
        pix = shard.local_pixels(rank)
        z = R.sample_depths(to_dev(tz[pix])[0], S, perturb=True, seed=31, ray_index=shard.ray_index(rank)).cpu().numpy()
        assert np.array_equal(z, orc.z_values_perturbed(tz[pix], S, 31, np.float32, ray_index=shard.ray_index(rank)))
        assert np.array_equal(z, orc.z_values_perturbed(tz, S, 31, np.float32)[pix])


def test_mip_renderer_with_in_kernel_jitter():
    """MipRenderer jitters its S+1 segment edges the same way (renderer.py:379-383)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import MipRenderer
    fam = synthetic.FAMILIES["grass_filtered"]
    model, spec, w = make_model((1, 3), kind="IPE", dense_media=True)
    (ro, rd, t, cone), _, _ = camera_rays("grass_filtered", 10, 12)
    hit = np.isfinite(t[:, 0])
    ro, rd, t, cone = ro[hit], rd[hit], t[hit], cone[hit]
    params = np.asarray([fam["params"]], np.float32)
    S, seed = 48, 4242
    out = MipRenderer(model=model, n_samples=S, perturb=True, blur_idx=0)(
        *to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0], seed=seed)
    z = orc.z_values_perturbed(t, S + 1, seed, np.float32)
    ref = orc.mip_render_rays(w, spec, ro, rd, t, np.repeat(params, ro.shape[0], 0), cone, S, 0, False, (1, 1, 1.), z_override=z,
                              dtype=np.float64)
    assert orc.rel_linf(rgba_of(out), rgba_ref(ref)) <= TOL


def test_hierarchical_sampling_sees_the_jittered_coarse_depths():
    """n_importance > 0 with perturb: ntx_sample_pdf regenerates the coarse depths of the same (flags, seed), so the merged
    depths contain exactly the jittered coarse ones (renderer.py:125-130)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    model, spec, w = make_model((1, 6), dense_media=True)
    n, S, NI, seed = 50, 32, 16, 777
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    params = np.asarray([fam["params"]], np.float32)
    r = Renderer(model=model, n_samples=S, n_importance=NI, perturb=True)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0], seed=seed)
    r.raise_if_nonfinite()
    z_all = r._last_z.cpu().numpy()
    zc = orc.z_values_perturbed(t, S, seed, np.float32)
    assert z_all.shape == (n, S + NI) and np.all(np.diff(z_all, axis=-1) >= 0)
    for k in range(n):
        assert np.all(np.isin(zc[k], z_all[k]))
    ref = orc.render_rays(w, spec, ro, rd, t, np.repeat(params, n, 0), cone, S + NI, False, (1, 1, 1.), z_override=z_all, dtype=np.float64)
    assert orc.rel_linf(rgba_of(out), rgba_ref(ref)) <= TOL
    refc = orc.render_rays(w, spec, ro, rd, t, np.repeat(params, n, 0), cone, S, False, (1, 1, 1.), z_override=zc, dtype=np.float64)
    got_c = np.concatenate([out["color_pred_coarse"][0].cpu().numpy(), out["alpha_pred_coarse"][0].cpu().numpy()[:, None]], -1)
    assert orc.rel_linf(got_c, rgba_ref(refc)) <= TOL


# ---------------------------------------------------------------------------------------------------------------
# cone-filter conditioning at BASELINE configs[4]'s sample count, and the deep-sampling families at full size
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
@pytest.mark.parametrize("sigma_blur", [10.0, 0.0])
def test_grass_filtered_blur_at_128_samples(precision, sigma_blur):
    """blur_idx = 0 scales the blur parameter per sample by cone_scale * z (renderer.py:155-158) -- at S = 128, the
    sample count of BASELINE configs[4] (config_grass_filtered_train.py:46-48: sigma_blur in {0, 10})."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["grass_filtered"]
    model, spec, w = make_model(fam["n_parameters"], dense_media=True)
    (ro, rd, t, cone), _, _ = camera_rays("grass_filtered", 16, 12)
    cone = (cone * 40).astype(np.float32)                                      # cone_scale * z * sigma_blur of order 1
    params = np.asarray([fam["params"]], np.float32); params[0, 0] = sigma_blur
    S = 128
    out = Renderer(model=model, n_samples=S, perturb=False, blur_idx=0, precision=precision)(
        *to_dev(ro[None], rd[None], t[None]), parameters=to_dev(params)[0], cone_scale=to_dev(cone[None])[0])
    ref = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.), 0, False, dtype=np.float64)
    assert orc.rel_linf(rgba_of(out), rgba_ref(ref, 0)) <= TOL
    if sigma_blur:                                                             # the conditioning is live
        ref0 = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.), None, False, dtype=np.float64)
        assert orc.rel_linf(rgba_ref(ref0, 0), rgba_ref(ref, 0)) > 10 * TOL


@pytest.mark.parametrize("family", ["grass", "grass_filtered"])
def test_full_size_properties_deep_sampling(family):
    """BASELINE configs[2] / [4] per-GPU size (800x800x128, all-hit rays): alpha in [0,1], premultiplied colour <= alpha,
    the image independent of how rays are split over calls (bit-identical: what sharding relies on), and a spot check of
    128 rays of the full-size render against the oracle."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES[family]
    model, spec, w = make_model(fam["n_parameters"], dense_media=True)
    n, S = 800 * 800, 128
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    params = np.asarray([fam["params"]], np.float32)
    r = Renderer(model=model, n_samples=S, perturb=False, blur_idx=fam["blur_idx"])
    dro, drd, dt, dcone = to_dev(ro, rd, t, cone)
    full = r(dro[None], drd[None], dt[None], parameters=to_dev(params)[0], cone_scale=dcone[None])
    r.raise_if_nonfinite()
    c, a = full["color_pred"][0], full["alpha_pred"][0]
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-6 and bool((c <= a[:, None] + 1e-6).all())
    cuts = [0, 33, 250_007, n]                                                 # incl. a shard smaller than one workgroup's block
    pc, pa = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = r(dro[None, lo:hi], drd[None, lo:hi], dt[None, lo:hi], parameters=to_dev(params)[0], cone_scale=dcone[None, lo:hi])
        pc.append(o["color_pred"][0]); pa.append(o["alpha_pred"][0])
    assert torch.equal(torch.cat(pc), c) and torch.equal(torch.cat(pa), a)
    idx = np.random.default_rng(1).choice(n, 128, replace=False)
    ref = orc.render_rays(w, spec, ro[idx], rd[idx], t[idx], np.repeat(params, 128, 0), cone[idx], S, False, (1, 1, 1.), fam["blur_idx"],
                          dtype=np.float64)
    got = np.concatenate([c[idx].cpu().numpy(), a[idx].cpu().numpy()[:, None]], -1)
    assert orc.rel_linf(got, rgba_ref(ref)) <= TOL


# ---------------------------------------------------------------------------------------------------------------
# error paths write nothing; scratch is reserved, never allocated per call; precision is per call
# ---------------------------------------------------------------------------------------------------------------
def _raw_render(model, n, S, blur_idx, flags, sentinel=-7.0):
    from nerf_tex_amd import _lib, synthetic
    fam = synthetic.FAMILIES["carpet"]
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    t[::3] = np.inf                                                            # culled rays are written by the FIRST launch
    dro, drd, dt, dcone = to_dev(ro, rd, t, cone[:, 0].copy())
    params = to_dev(np.asarray([fam["params"]], np.float32))[0]
    color = torch.full((n, 3), sentinel, device=dev()); alpha = torch.full((n,), sentinel, device=dev())
    rc = _lib.lib.ntx_render_rays(model.ctx(0), dro.data_ptr(), drd.data_ptr(), dt.data_ptr(), params.data_ptr(), n, dcone.data_ptr(), n, S,
                                  blur_idx, flags, _lib.f3([1, 1, 1.]), None, 0, None, color.data_ptr(), alpha.data_ptr(), None, None,
                                  torch.cuda.current_stream(dev()).cuda_stream)
    torch.cuda.synchronize()
    return rc, color, alpha


def test_failed_calls_write_nothing():
    from nerf_tex_amd import _lib
    model, _, _ = make_model((1, 6))
    # fp16x3 cannot scale an appearance parameter per sample: refused BEFORE the hit compaction touches the outputs
    rc, color, alpha = _raw_render(model, 300, 32, 6, _lib.FLAG_FP16X3)
    assert rc == _lib.NTX_E_UNSUPPORTED and b"appearance" in _lib.lib.ntx_last_error()
    assert bool((color == -7.0).all()) and bool((alpha == -7.0).all())
    # more rays than the context reserved: refused, nothing written, nothing allocated
    _lib.check(_lib.lib.ntx_reserve(model.ctx(0), 256))
    rc, color, alpha = _raw_render(model, 300, 32, -1, 0)
    assert rc == _lib.NTX_E_INVALID and b"ntx_reserve" in _lib.lib.ntx_last_error()
    assert bool((color == -7.0).all()) and bool((alpha == -7.0).all())
    _lib.check(_lib.lib.ntx_reserve(model.ctx(0), 300))
    rc, color, alpha = _raw_render(model, 300, 32, -1, 0)
    assert rc == 0 and bool((alpha[::3] == 0).all()) and bool((alpha[1::3] != -7.0).all())
    assert _lib.lib.ntx_reserve(model.ctx(0), -1) == _lib.NTX_E_INVALID
    # the python mirror reserves by itself (setup-time) when a call outgrows the default
    model.close()


def test_renderer_reserves_beyond_the_default():
    from nerf_tex_amd import _lib, synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    model, _, _ = make_model((1, 6), dense_media=True)
    n = _lib.DEFAULT_MAX_RAYS + 1000
    ro, rd, t, cone = synthetic.all_hit_rays(n, fam["b_0"], fam["b_1"], fam["cam"])
    t[:-5] = np.inf                                                            # all but 5 rays culled: cheap
    params = to_dev(np.asarray([fam["params"]], np.float32))[0]
    r = Renderer(model=model, n_samples=8, perturb=False)
    out = r(*to_dev(ro[None], rd[None], t[None]), parameters=params, cone_scale=to_dev(cone[None])[0])
    a = out["alpha_pred"][0]
    alone = r(*to_dev(ro[None, -5:], rd[None, -5:], t[None, -5:]), parameters=params, cone_scale=to_dev(cone[None, -5:])[0])
    assert bool((a[:-5] == 0).all()) and bool((a[-5:] > 0).any()) and torch.equal(a[-5:], alone["alpha_pred"][0])


def test_precision_is_per_call_on_a_shared_context():
    """Two renderers share one model context at different precisions on two streams: no mutable precision state."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    model, _, _ = make_model((1, 6), dense_media=True)
    ro, rd, t, cone = synthetic.all_hit_rays(5000, fam["b_0"], fam["b_1"], fam["cam"])
    args = to_dev(ro[None], rd[None], t[None])
    kw = dict(parameters=to_dev(np.asarray([fam["params"]], np.float32))[0], cone_scale=to_dev(cone[None])[0])
    r32 = Renderer(model=model, n_samples=64, perturb=False)
    r16 = Renderer(model=model, n_samples=64, perturb=False, precision="fp16x3")
    a32, a16 = rgba_of(r32(*args, **kw)), rgba_of(r16(*args, **kw))
    assert not np.array_equal(a32, a16) and orc.rel_linf(a16, a32) < 1e-4
    for _ in range(3):                                                         # interleaved: each keeps its own arithmetic
        assert np.array_equal(rgba_of(r16(*args, **kw)), a16)
        assert np.array_equal(rgba_of(r32(*args, **kw)), a32)


# ---------------------------------------------------------------------------------------------------------------
# multi-GPU plumbing that one GPU can exercise: strided ray generation, the RCCL gather on a 1-rank communicator
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,run", [(3, 40), (8, 40), (4, None), (3, 7)])
def test_strided_ray_generation_is_the_full_grid_resharded(world, run):
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.dist import ShardMap
    from nerf_tex_amd.pixel_sampler import Full
    from nerf_tex_amd.proxy import AABB
    from nerf_tex_amd.ray_sampler import Proxy
    fam = synthetic.FAMILIES["fur"]
    h, w = 30, 40
    c2w = orc.look_at(fam["cam"], dtype=np.float32)
    focal = orc.focal_from_angle(w, fam["angle"] * 2)
    sampler = Proxy(h, w, focal, AABB(fam["b_0"], fam["b_1"]))
    full = [x.cpu().numpy() for x in sampler(Full(h, w)(), c2w)]
    shard = ShardMap(h * w, world, run)
    for r in range(world):
        part = [x.cpu().numpy() for x in sampler(Full(h, w, shard=(shard, r))(), c2w)]
        px = shard.local_pixels(r)
        for a, b in zip(part, full):
            assert np.array_equal(a, b[px], equal_nan=True)


@pytest.mark.parametrize("n,run", [(800 * 10, None), (801, None), (800 * 10, 800), (4001, 16)])
def test_gather_image_on_a_one_rank_communicator(n, run):
    """ntx_comm_unique_id / ntx_comm_create / ntx_gather_image on real RCCL with one rank: the direct ncclGather path
    (bands) and the staging + unshard path (interleaved runs)."""
    from nerf_tex_amd.dist import Comm, ShardMap
    comm = Comm(0)
    assert comm.world == 1 and comm.rank == 0
    shard = ShardMap(n, 1, run)
    local = torch.rand((n, 4), device=dev())
    img = comm.gather_image(local, shard)
    torch.cuda.synchronize()
    assert img.shape == (n, 4) and torch.equal(img, local)
    with pytest.raises(ValueError):
        comm.gather_image(local[:-1], shard)
    comm.close()


@pytest.mark.parametrize("n,run", [(800 * 10, None), (801, None), (800 * 10, 800), (4001, 16), (640000, 800)])
def test_the_uneven_shards_branch_on_a_one_rank_communicator(n, run):
    """the branch of ntx_gather_image that uneven shards take -- grouped ncclSend / ncclRecv with exact counts into the staging
    buffer, then the un-shard kernel (csrc/ntx_comm.hip) -- had only ever run with one rank, where it did nothing.  NTX_GATHER_FORCE_EXCHANGE
    makes a communicator of ONE rank take it in full: the root receives its own block from itself through RCCL (send + recv in one group)
    instead of copying it, and the un-shard kernel deals the runs into pixel order.  Also: the librccl the calls went to and its version."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.dist import Comm, ShardMap
    comm = Comm(0)
    assert comm.version >= 20000 and comm.init_seconds >= 0 and "rccl" in comm.library
    shard = ShardMap(n, 1, run)
    local = torch.rand((n, 4), device=dev())
    img = comm.gather_image(local, shard, force_exchange=True)
    torch.cuda.synchronize()
    assert img.shape == (n, 4) and torch.equal(img, local)
    # the flag without a staging buffer is refused before any RCCL call, an unknown flag too
    image = torch.empty((n, 4), device=dev()); st = torch.cuda.current_stream(dev()).cuda_stream
    assert _lib.lib.ntx_gather_image_ex(comm.handle, local.data_ptr(), n, n, image.data_ptr(), None, 0, _lib.GATHER_FORCE_EXCHANGE, st) == _lib.NTX_E_INVALID
    assert _lib.lib.ntx_gather_image_ex(comm.handle, local.data_ptr(), n, n, image.data_ptr(), None, 0, 2, st) == _lib.NTX_E_INVALID
    comm.close()


def test_two_rank_sharded_render_is_bit_identical():
    """BASELINE configs[3] over 2 ranks (rows dealt round-robin, ntx_gather_image over RCCL): the gathered image equals the
    one-GPU image bit for bit.  Needs two GPUs; the one-GPU box skips it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    for shard in ("rows", "bands"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "fur_sharded", "--steps", "1",
                              "--warmup", "1", "--no-cpu-baseline", "--shard", shard], capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["sharded_image_bit_identical_to_1gpu"] is True


def test_sharded_workload_on_one_gpu():
    """`--workload fur_sharded` at N = 1: the true camera (hits and misses), rays generated on the device."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "fur_sharded", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    hits = d["config"]["hit_rays_total"]
    assert 0 < hits < 800 * 800 and d["scaling"] == "strong"
    assert abs(d["value"] - hits * 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


# ---------------------------------------------------------------------------------------------------------------
# the generic family: any ParamNerf n_parameters = [g <= 4, a <= 8] that has no tuned kernel family of its own
# ---------------------------------------------------------------------------------------------------------------
GENERIC = [(3, 3), (0, 2), (2, 0), (4, 8), (1, 1), (0, 0), (2, 5)]


@pytest.mark.parametrize("npar", GENERIC)
@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
def test_generic_family_mlp(npar, precision):
    """model.py:58-125 with parameter counts no shipped config uses: the kernels of Cfg<4, 8> with zero rows for the
    parameters the model does not have (and 0 fed into their slots) against the oracle of the model's own dimensions."""
    from tests.common import random_samples
    model, spec, w = make_model(npar)
    model.precision = precision
    m = 1000
    pos, dirs, params = random_samples(m, sum(npar), seed=sum(npar) + 5)
    color, alpha = model(tuple(to_dev(pos, dirs, params)) if sum(npar) else (to_dev(pos)[0], to_dev(dirs)[0], None))
    rc, ra = orc.model_forward(w, spec, pos, dirs, params, np.float64)
    err = orc.rel_linf(np.concatenate([color.cpu().numpy(), alpha.cpu().numpy()], -1), np.concatenate([rc, ra], -1))
    assert err <= 2e-5, err


@pytest.mark.parametrize("npar,blur", [((3, 3), None), ((3, 3), 1), ((3, 3), 4), ((0, 2), None), ((2, 0), 0), ((4, 8), None), ((4, 8), 3),
                                       ((1, 1), None), ((0, 0), None), ((2, 5), 6)])
def test_generic_family_render(npar, blur):
    """The fused render on the generic family: hits and misses, per-image parameters, a blur_idx on a geometry parameter
    (direction vectors still per ray), on an appearance parameter (float32 per-sample kernel; fp16x3 refuses), none."""
    from nerf_tex_amd import _lib, synthetic
    from nerf_tex_amd.renderer import Renderer
    model, spec, w = make_model(npar, dense_media=True)
    (ro, rd, t, cone), _, _ = camera_rays("carpet", 14, 12)
    cone = (cone * 30).astype(np.float32)
    P = sum(npar)
    rng = np.random.default_rng(P + 11)
    params = rng.uniform(0.1, 1.0, size=(1, P)).astype(np.float32)
    S = 40
    args = to_dev(ro[None], rd[None], t[None])
    kw = dict(parameters=to_dev(params)[0] if P else torch.zeros((1, 0), device=dev()), cone_scale=to_dev(cone[None])[0])
    ref = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.), blur, False, dtype=np.float64)
    ref32 = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1., 1., 1.), blur, False, dtype=np.float32)
    want, want32 = rgba_ref(ref, 0), rgba_ref(ref32, 0).astype(np.float64)
    # dense-media weights on a nearly empty image (max alpha 0.2 for the parameter-less models): the float32 restatement itself
    # sits ~1e-4 from float64 there.  The gate is the north star's -- 1e-4 against the float32 restatement, the
    # reference's own arithmetic -- and the float64 figure is asserted as what follows from it by the triangle inequality,
    # |got - f64| <= |got - f32| + |f32 - f64| (same normalisation up to max|f32| / max|f64|), not as a tolerance of its own
    floor = orc.rel_linf(want32, want)
    scale = float(np.abs(want32).max() / np.abs(want).max())

    def check(got):
        e32 = orc.rel_linf(got, want32)
        assert e32 <= TOL
        assert orc.rel_linf(got, want) <= e32 * scale + floor + 1e-12

    got = rgba_of(Renderer(model=model, n_samples=S, perturb=False, blur_idx=blur)(*args, **kw))
    check(got)
    miss = ~np.isfinite(t[:, 0])
    assert miss.any() and np.all(got[miss] == 0)
    r16 = Renderer(model=model, n_samples=S, perturb=False, blur_idx=blur, precision="fp16x3")
    if blur is not None and blur >= npar[0]:
        with pytest.raises(_lib.NtxError) as e:
            r16(*args, **kw)
        assert e.value.code == _lib.NTX_E_UNSUPPORTED
    else:
        check(rgba_of(r16(*args, **kw)))


def test_generic_family_matches_a_tuned_family_on_its_own_dimensions(monkeypatch):
    """[1,6] has a tuned family; forced onto the generic kernels (one geometry slot of four, six appearance slots of eight in
    use) it must give the same image -- bit for bit: see the last line.  (The knob is read by ntx_create.)"""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.renderer import Renderer
    fam = synthetic.FAMILIES["carpet"]
    ro, rd, t, cone = synthetic.all_hit_rays(3000, fam["b_0"], fam["b_1"], fam["cam"])
    params = to_dev(np.asarray([fam["params"]], np.float32))[0]
    args = to_dev(ro[None], rd[None], t[None])

    def render():
        model, _, _ = make_model((1, 6), dense_media=True)
        return rgba_of(Renderer(model=model, n_samples=64, perturb=False)(*args, parameters=params, cone_scale=to_dev(cone[None])[0]))

    tuned = render()
    monkeypatch.setenv("NERFTEX_FORCE_GENERIC", "1")
    generic = render()
    monkeypatch.delenv("NERFTEX_FORCE_GENERIC")
    # the zero rows contribute exact zeros and the non-zero k-steps keep their relative order: not merely close, the same bits
    assert np.array_equal(tuned, generic)


def test_gather_image_argument_errors():
    """ntx_gather_image refuses bad arguments before any RCCL call: NULL communicator, root out of range, an interleaved
    shard map without a staging buffer on the root."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.dist import Comm
    comm = Comm(0)
    n = 4000
    local = torch.rand((n, 4), device=dev()); image = torch.empty((n, 4), device=dev())
    st = torch.cuda.current_stream(dev()).cuda_stream
    call = lambda h, run, root, staging: _lib.lib.ntx_gather_image(h, local.data_ptr(), n, run, image.data_ptr(), staging, root, st)
    assert call(None, n, 0, None) == _lib.NTX_E_INVALID
    assert call(comm.handle, n, 1, None) == _lib.NTX_E_INVALID
    assert call(comm.handle, 0, 0, None) == _lib.NTX_E_INVALID
    assert call(comm.handle, 100, 0, None) == _lib.NTX_E_INVALID and b"staging" in _lib.lib.ntx_last_error()
    assert call(comm.handle, n, 0, None) == 0                      # bands on one rank: straight into image_out
    torch.cuda.synchronize()
    assert torch.equal(image, local)
    comm.close()


def test_main_entry_point_renders_the_example_config(tmp_path, monkeypatch):
    """`python -m nerf_tex_amd.main configs/example_carpet_render.py` (reference: main.py + render.py): two 128x128 views,
    perturb=True by default (in-kernel jitter), images returned and saved under <target_path>/media/test."""
    from nerf_tex_amd import main as m
    monkeypatch.chdir(tmp_path)
    imgs = m.main([os.path.join(ROOT, "configs", "example_carpet_render.py")])
    assert len(imgs) == 2 and tuple(imgs[0].shape) == (1, 128, 128, 4)
    a = imgs[0][0, ..., 3]
    assert bool(torch.isfinite(imgs[0]).all()) and float(a.max()) > 0.0 and float(a.min()) == 0.0      # object and empty background
    saved = sorted(os.listdir(tmp_path / "logs" / "example_carpet" / "media" / "test"))
    assert saved == ["0.npy", "0.png", "1.npy", "1.png"] and os.path.exists(tmp_path / "logs" / "example_carpet" / "config_render.py")
    # logger_config.write_exr (logger.py:132-142): the premultiplied float32 image as it is, one OpenEXR file a view
    from nerf_tex_amd import exr, util
    cfg = m.prepare(m.load_config(os.path.join(ROOT, "configs", "example_carpet_render.py")))
    cfg.target_path = str(tmp_path / "exr"); cfg.logger_config = dict(cfg.get("logger_config") or {}, write_exr=True)
    cfg.renderer_config = dict(cfg.renderer_config, perturb=False)
    again = util.instantiate(cfg)
    assert sorted(os.listdir(tmp_path / "exr" / "media" / "test")) == ["0.exr", "0.npy", "1.exr", "1.npy"]
    assert np.array_equal(exr.read_exr(str(tmp_path / "exr" / "media" / "test" / "1.exr")), again[1][0].cpu().numpy())


def test_rays_at_any_image_plane_locations():
    """ray_sampler.Proxy / Frustum called with a pixel TENSOR, as the reference's pixel samplers hand it over (pixel_sampler.py:15, 29,
    69 -> ray_sampler.py:17, 25): `ntx_generate_rays_at`.  The full grid as a tensor gives the bits of the pixel-range path; a random
    pixel list matches the float64 restatement of rays_from_camera + AABB."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.pixel_sampler import Full, Independent
    from nerf_tex_amd.proxy import AABB
    from nerf_tex_amd.ray_sampler import Frustum, Proxy
    fam = synthetic.FAMILIES["carpet"]
    H, W = 37, 53
    c2w = orc.look_at(fam["cam"], dtype=np.float32)
    focal = orc.focal_from_angle(W, fam["angle"] * 2.5)
    box = AABB(fam["b_0"], fam["b_1"])
    for sampler in (Proxy(H, W, focal, box), Frustum(H, W, focal, 2.0, 6.0)):
        by_range = sampler(Full(H, W)(), c2w, device=dev())
        by_tensor = sampler(Full(H, W).as_tensor(dev()), c2w)
        assert all(torch.equal(a, b) for a, b in zip(by_range, by_tensor))
    g = torch.Generator(device=dev()); g.manual_seed(3)
    pix = Independent(H, W, 500)(device=dev(), generator=g)
    assert pix.shape == (500, 2) and pix.dtype == torch.int32 and int(pix[:, 0].max()) < H and int(pix[:, 1].max()) < W and int(pix.min()) >= 0
    o, d, t, cone = Proxy(H, W, focal, box)(pix, c2w)
    ro, rd, tt, cc = orc.proxy_rays(pix.cpu().numpy(), H, W, focal, c2w, fam["b_0"], fam["b_1"], np.float64)
    assert np.max(np.abs(o.cpu().numpy() - ro)) <= 1e-6 and np.max(np.abs(d.cpu().numpy() - rd)) <= 1e-6
    hit = np.isfinite(tt[:, 0]); got_hit = np.isfinite(t.cpu().numpy()[:, 0])
    assert hit.any() and (~hit).any() and np.mean(hit == got_hit) > 0.99
    ok = hit & got_hit
    assert np.max(np.abs(t.cpu().numpy()[ok] - tt[ok]) / np.abs(tt[ok])) <= 1e-5
    assert np.max(np.abs(cone.cpu().numpy() - cc) / cc) <= 1e-5
    # sub-pixel locations are locations too (the reference casts image_plane_loc to float32 and never rounds it)
    loc = torch.rand((64, 2), device=dev()) * torch.tensor([H - 1.0, W - 1.0], device=dev())
    o2, d2, _, _ = Frustum(H, W, focal, 2.0, 6.0)(loc, c2w)
    _, rd2, _ = orc.rays_from_camera(loc.cpu().numpy(), H, W, focal, c2w, np.float64)
    assert np.max(np.abs(d2.cpu().numpy() - rd2)) <= 1e-6


def test_aabb_call_is_the_reference_slab_test():
    """proxy.AABB.__call__ (proxy.py:13-35) on the caller's rays through `ntx_aabb_intersect`: bit-equal to the float32 restatement,
    IEEE corners included (axis-parallel rays: 1/0 = inf, 0 * inf = NaN, the tf.where comparisons decide)."""
    from nerf_tex_amd.proxy import AABB
    rng = np.random.default_rng(0)
    n = 4000
    o = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    o[np.all(np.abs(o) < 1.6, -1)] += 3.0                          # origins outside the box, as the reference assumes
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:200, 0] = 0.0; d[100:300, 1] = 0.0; d[300:320] = [0.0, 0.0, 1.0]     # axis-parallel rays
    o[320:340, 0] = 1.5; d[320:340, 0] = 0.0                       # ... and on a face plane: (b - o) * inf = 0 * inf
    box = AABB([-1.5, -1.5, -1.5], [1.5, 1.5, 1.5])
    t = box(*to_dev(o, d)).cpu().numpy()
    want = orc.aabb(o, d, box.b_0, box.b_1, np.float32)
    assert np.array_equal(t, want, equal_nan=True)
    assert np.isinf(t[:, 0]).any() and np.isfinite(t[:, 0]).any()


def test_pixel_sampler_proxy_samples_hit_pixels():
    """pixel_sampler.Proxy (pixel_sampler.py:31-69): the sampled pixels are pixels of the nearest-neighbour upsampled hit mask of the
    coarse grid; every such pixel can be drawn; n_samples of them, all different (a permutation's head)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.pixel_sampler import Proxy as PixelProxy
    from nerf_tex_amd.proxy import AABB
    fam = synthetic.FAMILIES["carpet"]
    H, W, f = 100, 132, 8                                           # 100 is not a multiple of 8: the half-pixel-centre rule matters
    c2w = orc.look_at(fam["cam"], dtype=np.float32)
    focal = orc.focal_from_angle(W, fam["angle"] * 2.5)
    ps = PixelProxy(H, W, 300, AABB(fam["b_0"], fam["b_1"]), focal, downsample_factor=f)
    hd, wd = H // f, W // f
    _, _, t, _ = orc.frustum_rays(orc.full_pixels(hd, wd), hd, wd, focal // f, c2w, 0.0, 0.0, np.float32)
    ro, rd, _ = orc.rays_from_camera(orc.full_pixels(hd, wd), hd, wd, focal // f, c2w, np.float32)
    hit = np.isfinite(orc.aabb(ro, rd, fam["b_0"], fam["b_1"], np.float32)[:, 0]).reshape(hd, wd)
    src = lambda n_dst, n_src: np.minimum(np.floor((np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * (np.float32(n_src) / np.float32(n_dst))).astype(int), n_src - 1)
    up = hit[src(H, hd)][:, src(W, wd)]
    assert np.array_equal(ps.hit_mask(c2w, dev()).cpu().numpy(), up) and 0.05 < up.mean() < 0.95
    g = torch.Generator(device=dev()); g.manual_seed(1)
    pix = ps(c2w, device=dev(), generator=g).cpu().numpy()
    assert pix.shape == (300, 2) and up[pix[:, 0], pix[:, 1]].all() and len({tuple(p) for p in pix}) == 300
    seen = np.zeros_like(up)
    for s in range(40):
        g.manual_seed(s)
        q = ps(c2w, device=dev(), generator=g).cpu().numpy()
        seen[q[:, 0], q[:, 1]] = True
    assert seen[up].mean() > 0.5 and not seen[~up].any()
