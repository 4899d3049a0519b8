"""CPU tests of the oracle itself: against the committed goldens (drift pin), numpy vs the independent
C restatement, float32 vs float64, and properties of the reference's math."""

import hashlib
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import c_oracle, nerftex_oracle as orc
from nerf_tex_amd import synthetic

G = os.path.join(os.path.dirname(__file__), "golden")
FAMILIES = ["carpet", "grass", "fur", "grass_filtered"]


def load(family):
    g = np.load(os.path.join(G, f"golden_{family}.npz"))
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(int(v) for v in g["n_parameters"]))
    blob = synthetic.synthetic_weights(orc.layer_table(spec), seed=int(g["weights_seed"]), dense_media=bool(g["weights_dense_media"]))
    assert hashlib.sha256(blob.tobytes()).hexdigest() == str(g["weights_sha256"]), "seeded weights drifted"
    return g, spec, blob, orc.split_blob(spec, blob)


def test_macs_and_weight_counts_match_survey():
    # MACs / sample and bias counts per config family
    for npar, macs in [((1, 6), 680832), ((1, 4), 676224), ((2, 3), 678528)]:
        spec = orc.ModelSpec(n_parameters=npar)
        assert orc.macs_per_sample(spec) == macs
        assert orc.n_weight_floats(spec) == macs + 2692
    assert orc.macs_per_sample(orc.ModelSpec(kind="Nerf")) == 593408


def test_camera_fixtures_from_reference_modules():
    """The poses/params the reference's own TF-free code emitted (gen_golden.py part 1)."""
    cam = json.load(open(os.path.join(G, "cameras_carpet.json")))
    p = np.asarray([v["pose_dist_sample"] for v in cam["views"]])
    assert p.shape == (5, 3)
    np.testing.assert_allclose(p[:, 2], 0.4, atol=1e-12)                 # latitude u = .3 -> z = 1 - 2u
    np.testing.assert_allclose(np.linalg.norm(p, axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(p[0], [0.9165151389911681, 0.0, 0.4], atol=1e-12)
    np.testing.assert_allclose(p[1], [0.2832, 0.8717, 0.4], atol=1e-4)   # values recorded in SURVEY.md 8c
    assert cam["views"][0]["parameters"] == [1, 1, 1, .1, 0, 0, 1]
    assert cam["views"][0]["radius"] == 6.0
    # look_at restatement: orthonormal right-handed frame looking at the origin
    c2w = np.asarray(cam["views"][1]["c2w_oracle_look_at_f32"])
    R = c2w[:3, :3]
    np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-5)
    np.testing.assert_allclose(c2w[:3, 3], p[1] * 6.0, atol=1e-6)
    np.testing.assert_allclose(R[:, 2], p[1], atol=1e-5)                 # forward = normalised position


def test_product_look_at_matches_fixture():
    from nerf_tex_amd.dataset import look_at
    for fam in ("carpet", "grass", "plush"):
        cam = json.load(open(os.path.join(G, f"cameras_{fam}.json")))
        for v in cam["views"]:
            c2w = look_at(np.asarray(v["pose_dist_sample"]) * v["radius"])
            np.testing.assert_allclose(c2w, np.asarray(v["c2w_oracle_look_at_f32"]), atol=2e-7)


@pytest.mark.parametrize("family", FAMILIES)
def test_oracle_reproduces_golden(family):
    g, spec, blob, w = load(family)
    S = int(g["n_samples"]); bi = None if int(g["blur_idx"]) < 0 else int(g["blur_idx"])
    hit = g["hit"]
    ro, rd, t, cone = orc.proxy_rays(orc.full_pixels(int(g["height"]), int(g["width"])), int(g["height"]), int(g["width"]),
                                     float(g["focal"]), g["c2w"], g["b_0"], g["b_1"], np.float32)
    for a, b in ((ro, g["rays_o"]), (rd, g["rays_d"]), (t, g["t"]), (cone, g["cone_scale"])):
        np.testing.assert_array_equal(a, b)
    aux = orc.render_rays(w, spec, ro[hit], rd[hit], t[hit], np.repeat(g["parameters"], hit.sum(), 0), cone[hit], S,
                          False, (1, 1, 1.), bi, dtype=np.float64, return_aux=True)
    np.testing.assert_allclose(aux["z_vals"], g["z_vals"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(aux["raw_color"], g["raw_color"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(aux["weights"], g["weights"], rtol=1e-9, atol=1e-12)
    full = orc.renderer_call(w, spec, ro[None], rd[None], t[None], g["parameters"], cone[None], S, True, tuple(g["bkgd"]),
                             bi, dtype=np.float64)
    np.testing.assert_allclose(full["color_pred"], g["color_pred_bkgd"], rtol=1e-9, atol=1e-11)
    assert np.all(full["alpha_pred"][0][~hit] == 0) and np.all(full["color_pred"][0][~hit] == g["bkgd"])


@pytest.mark.parametrize("family", FAMILIES)
def test_numpy_and_c_restatements_agree(family):
    """Two independent readings of the reference source must give the same numbers."""
    g, spec, blob, w = load(family)
    S = int(g["n_samples"]); bi = None if int(g["blur_idx"]) < 0 else int(g["blur_idx"])
    c, a = c_oracle.model(spec, blob, g["m_pos"], g["m_dirs"], g["m_params"], np.float64)
    np.testing.assert_allclose(c, g["m_color"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a, g["m_alpha"], rtol=1e-10, atol=1e-12)
    hit = g["hit"]
    cc, aa = c_oracle.render_rays(spec, blob, g["rays_o"][hit], g["rays_d"][hit], g["t"][hit],
                                  np.repeat(g["parameters"], hit.sum(), 0), g["cone_scale"][hit], S, bi, dtype=np.float64)
    np.testing.assert_allclose(cc, g["color_pred"][0][hit], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(aa, g["alpha_pred"][0][hit], rtol=1e-9, atol=1e-12)
    # float32 builds of both restatements stay inside the north-star gate
    c32, a32 = c_oracle.render_rays(spec, blob, g["rays_o"][hit], g["rays_d"][hit], g["t"][hit],
                                    np.repeat(g["parameters"], hit.sum(), 0), g["cone_scale"][hit], S, bi, dtype=np.float32)
    ref = np.concatenate([g["color_pred"][0][hit], g["alpha_pred"][0][hit][:, None]], -1)
    assert orc.rel_linf(np.concatenate([c32, a32[:, None]], -1), ref) <= 1e-4
    n32 = orc.render_rays(w, spec, g["rays_o"][hit], g["rays_d"][hit], g["t"][hit], np.repeat(g["parameters"], hit.sum(), 0),
                          g["cone_scale"][hit], S, False, (1, 1, 1.), bi, dtype=np.float32)
    assert orc.rel_linf(np.concatenate([n32["color_pred"], n32["alpha_pred"][:, None]], -1), ref) <= 1e-4


def test_edge_golden_and_c_composite():
    g = np.load(os.path.join(G, "golden_edge.npz"))
    for exr in (0, 1):
        for bk in (0, 1):
            c, a, w, _ = orc.map_model_output(g["color"], g["sigma"], g["z"], g["rays_d"], bool(bk), tuple(g["bkgd"]), bool(exr), None, np.float64)
            np.testing.assert_allclose(c, g[f"color_exr{exr}_bk{bk}"], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose(a, g[f"alpha_exr{exr}_bk{bk}"], rtol=1e-12, atol=1e-14)
            cc, ca, cw = c_oracle.composite(g["color"], g["sigma"], g["z"], g["rays_d"], exr, bk, g["bkgd"], np.float64)
            np.testing.assert_allclose(cc, c, rtol=1e-10, atol=1e-13)
            np.testing.assert_allclose(cw, w, rtol=1e-10, atol=1e-13)
    a = g["alpha_exr0_bk0"]; w = g["weights_exr0_bk0"]
    assert abs(w[0, 0] - 1.0) < 1e-12 and abs(w[0, 1] - 1e-10) < 1e-18      # opaque ray: the 1e-10 floor
    assert a[1] == 0 and a[2] == 0                                            # relu(sigma) = 0
    assert np.all(w[6] == 0)                                                  # zero-length steps
    assert np.count_nonzero(w[3]) == 1 and w[3, -1] > 0                       # last sample uses the copied dist


def test_plumbing_image_crop():
    """BASELINE configs[0] (carpet 200x200x32): recompute a 10-row band in float32 and float64."""
    g = np.load(os.path.join(G, "golden_plumbing.npz"))
    spec = orc.ModelSpec(n_parameters=(1, 6))
    blob = synthetic.synthetic_weights(orc.layer_table(spec), seed=0, dense_media=True)
    assert hashlib.sha256(blob.tobytes()).hexdigest() == str(g["weights_sha256"])
    w = orc.split_blob(spec, blob)
    H, W, S = int(g["height"]), int(g["width"]), int(g["n_samples"])
    rows = slice(95 * W, 105 * W)
    ro, rd, t, cone = orc.proxy_rays(orc.full_pixels(H, W)[rows], H, W, float(g["focal"]), g["c2w"], g["b_0"], g["b_1"], np.float32)
    ref = g["rgba"].reshape(-1, 4)[rows]
    for dtype, tol in ((np.float64, 2e-7), (np.float32, 1e-4)):     # 2e-7: the fixture is stored as float32
        pred = orc.renderer_call(w, spec, ro[None], rd[None], t[None], g["parameters"], cone[None], S, dtype=dtype)
        got = np.concatenate([pred["color_pred"][0], pred["alpha_pred"][0][:, None]], -1)
        assert orc.rel_linf(got, ref) <= tol


# ---- properties -----------------------------------------------------------------------------
def test_fourier_layout():
    x = np.asarray([[0.25, -1.0, 2.0]], np.float64)
    o = orc.fourier_features(x, 3, np.float64)[0]
    assert o.shape == (21,)
    np.testing.assert_array_equal(o[:3], x[0])
    np.testing.assert_allclose(o[3:6], np.sin(x[0])); np.testing.assert_allclose(o[6:9], np.cos(x[0]))
    np.testing.assert_allclose(o[15:18], np.sin(4 * x[0])); np.testing.assert_allclose(o[18:21], np.cos(4 * x[0]))


def test_linspace_tf_endpoints_and_step():
    for n in (2, 3, 32, 64, 128, 257):
        v = orc.linspace_tf(n, np.float32)
        assert v.dtype == np.float32 and v[0] == 0 and v[-1] == 1 and np.all(np.diff(v) > 0)
        assert np.max(np.abs(v - np.linspace(0, 1, n))) <= 2e-7      # delta rounded once, then k*delta rounded


def test_aabb_hit_miss_and_ieee():
    o = np.asarray([[0, 0, 5.], [0, 0, 5.], [3, 0, 5.], [0, 0, 5.]], np.float32)
    d = np.asarray([[0, 0, -1.], [0, 1, 0.], [0, 0, -1.], [1e-3, 0, -1.]], np.float32)
    t = orc.aabb(o, d, [-1, -1, -1.], [1, 1, 1.], np.float32)
    np.testing.assert_allclose(t[0], [4, 6])        # axis-aligned: 1/0 = inf handled by the slab test
    assert np.all(np.isinf(t[1])) and np.all(np.isinf(t[2]))
    assert np.isfinite(t[3]).all()


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 70), st.integers(0, 2 ** 31 - 1), st.booleans())
def test_composite_properties(S, seed, exr):
    rng = np.random.default_rng(seed)
    n = 5
    color = rng.normal(size=(n, S, 3)) * 4; sigma = rng.normal(size=(n, S)) * 30
    z = np.sort(rng.uniform(0.5, 9, size=(n, S)), -1); d = rng.normal(size=(n, 3))
    c, a, w, _ = orc.map_model_output(color, sigma, z, d, False, (1, 1, 1.), exr, None, np.float64)
    assert np.all(w >= 0) and np.all(a <= 1 + S * 1e-10) and np.all(a >= 0)
    if not exr:
        assert np.all(c <= a[:, None] + 1e-12) and np.all(c >= 0)           # sigmoid colours, premultiplied
    cb, ab, _, _ = orc.map_model_output(color, sigma, z, d, True, (.3, .2, .1), exr, None, np.float64)
    np.testing.assert_allclose(cb, c + (1 - a)[:, None] * np.asarray([.3, .2, .1]), atol=1e-14)
    # scaling rays_d and dividing the densities leaves the result unchanged (dists are world-space, :180)
    c2, a2, _, _ = orc.map_model_output(color, sigma / 3.0, z, d * 3.0, False, (1, 1, 1.), exr, None, np.float64)
    np.testing.assert_allclose(a2, a, rtol=1e-9, atol=1e-12)


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_renderer_call_is_per_ray(seed):
    """Nothing couples two rays (what ray sharding relies on): rendering a permutation of the rays gives
    the permuted image; culled rays are zero."""
    rng = np.random.default_rng(seed)
    spec = orc.ModelSpec(kind="Nerf")
    w = orc.split_blob(spec, synthetic.synthetic_weights(orc.layer_table(spec), seed=1))
    n, S = 12, 8
    ro, rd, t, cone = synthetic.all_hit_rays(n, (-1, -1, -1), (1, 1, 1), (0, 0, 4), seed=seed % 1000)
    t[rng.integers(0, n, 3)] = np.inf
    prm = np.zeros((1, 0), np.float32)
    a = orc.renderer_call(w, spec, ro[None], rd[None], t[None], prm, cone[None], S, dtype=np.float64)
    perm = rng.permutation(n)
    b = orc.renderer_call(w, spec, ro[None, perm], rd[None, perm], t[None, perm], prm, cone[None, perm], S, dtype=np.float64)
    np.testing.assert_allclose(b["color_pred"][0], a["color_pred"][0][perm], rtol=1e-12, atol=1e-15)
    assert np.all(a["alpha_pred"][0][np.isinf(t[:, 0])] == 0)


def test_filtered_downsample_and_epilogue():
    """interpolate.py:68-82 / logger.py:128-144 restatement: normalisation, SAME output size, even-size shift."""
    for f, K in ((2, 6), (3, 9), (4, 12)):
        k1 = orc.gaussian_kernel_1d(K, f * .5, np.float64)
        assert k1.shape == (K,)
        if K % 2 == 0:
            assert np.argmax(k1) == K // 2 - 1 or k1[K // 2 - 1] > k1[K // 2]   # +0.5 shift: asymmetric taps (reference quirk)
    img = np.ones((37, 50, 4))
    out = orc.filtered_downsample(img, 2, dtype=np.float64)
    assert out.shape == (19, 25, 4)
    np.testing.assert_allclose(out[3:-3, 3:-3], 1.0, atol=1e-12)                # interior: kernel sums to 1
    assert out[0, 0, 0] < 1.0                                                    # zero padding at the border
    rgba = np.zeros((4, 4, 4)); rgba[..., :3] = 0.25; rgba[..., 3] = 0.5
    e = orc.image_epilogue(rgba, 1, False, np.float64)
    np.testing.assert_allclose(e[..., :3], 0.25 / (0.5 + 1e-5)); np.testing.assert_allclose(e[..., 3], 0.5)
    np.testing.assert_array_equal(orc.image_epilogue(rgba, 1, True, np.float64), rgba)
    np.testing.assert_array_equal(orc.to_uint8(np.asarray([0.0, 1.0, 0.5, -1.0, 2.0, 0.999])), [0, 255, 127, 0, 255, 255])


def test_instance_tail_properties():
    """InstanceRenderer tail (renderer.py:247-354): a ray with no in-patch sample returns its appended sample;
    skipped samples never contribute; un-hit rays are 0 even with a background."""
    spec = orc.ModelSpec(kind="Nerf")
    w = orc.split_blob(spec, synthetic.synthetic_weights(orc.layer_table(spec), seed=2))
    rng = np.random.default_rng(0)
    n, S = 6, 9
    rd = rng.normal(size=(n, S, 3)); pts = rng.normal(size=(n, S, 3)); t = rng.uniform(1, 2, size=(n, S))
    dists = rng.uniform(0.001, 0.003, size=(n, S)); dists[0] = 0; dists[1, ::2] = -1
    cl = rng.uniform(size=(n, 1, 3)); al = np.ones((n, 1)); aw = np.ones((n, S)); ids = np.zeros((n, S), np.int32)
    hit = np.array([1, 1, 1, 0, 1, 1], bool); prm = np.zeros((n, S, 0)); cone = np.zeros((n, 1))
    c, a = orc.instance_evaluate_model(w, spec, rd, pts, t, dists, cl, al, aw, ids, hit, prm, cone, None, 0.09, 50.0, True,
                                       False, True, (1, 1, 1.), None, dtype=np.float64)
    np.testing.assert_allclose(c[0], cl[0, 0], atol=1e-9); assert abs(a[0] - 1) < 1e-9     # only the opaque appended sample
    assert np.all(c[3] == 0) and a[3] == 0                                                # un-hit: 0 despite composite_bkgd
    pts2 = pts.copy(); pts2[1, ::2] = 1e3                                                 # moving skipped samples changes nothing
    c2, a2 = orc.instance_evaluate_model(w, spec, rd, pts2, t, dists, cl, al, aw, ids, hit, prm, cone, None, 0.09, 50.0, True,
                                         False, True, (1, 1, 1.), None, dtype=np.float64)
    np.testing.assert_array_equal(c2, c)


def test_sample_pdf_properties():
    """renderer.py:589-617: samples stay inside the bins, follow the mass, and det=True is monotone."""
    rng = np.random.default_rng(0)
    z = np.sort(rng.uniform(2, 6, (4, 20)), -1); mid = .5 * (z[:, 1:] + z[:, :-1])
    w = np.zeros((4, 20)); w[:, 7] = 1.0                         # all mass in one interior weight
    zs = orc.sample_pdf(mid, w[:, 1:-1], 64, det=True, dtype=np.float64)
    assert np.all(np.diff(zs, axis=-1) >= 0) and np.all(zs >= mid[:, :1]) and np.all(zs <= mid[:, -1:])
    inside = (zs >= mid[:, 6:7]) & (zs <= mid[:, 7:8])           # weight 7 spans bins[6]..bins[7]
    assert inside.mean() > 0.9
    u = rng.uniform(size=(4, 33))
    a = orc.sample_pdf(mid, w[:, 1:-1], 33, det=False, u=u, dtype=np.float64)
    b = orc.sample_pdf(mid, w[:, 1:-1], 33, det=False, u=u, dtype=np.float32)
    assert np.max(np.abs(a - b)) < 1e-4


def test_ipe_and_cone_gaussians():
    """layer.py:25-41 and renderer.py:411-437: zero covariance gives plain [sin | cos] (frequency-major), large covariance
    damps to 0; a degenerate cone (radius 0) has covariance only along the ray."""
    x = np.asarray([[0.3, -1.2, 2.0, 0, 0, 0]])
    e = orc.integrated_positional_encoding(x, 4, np.float64)[0]
    assert e.shape == (24,)
    y = (x[0, None, :3] * (2.0 ** np.arange(4))[:, None]).reshape(-1)
    np.testing.assert_allclose(e[:12], np.sin(y), atol=1e-15); np.testing.assert_allclose(e[12:], np.cos(y), atol=1e-12)
    x[0, 3:] = 50.0
    assert np.max(np.abs(orc.integrated_positional_encoding(x, 4, np.float64))) < 1e-10
    o = np.zeros((1, 3)); dd = np.asarray([[0., 0., 2.]]); tv = np.asarray([[1.0, 1.5, 2.5]])
    mean, cov = orc.cone_segment_gaussians(o, dd, tv, np.zeros((1, 1)), np.float64)
    assert mean.shape == (1, 2, 3) and np.all(cov[..., :2] == 0) and np.all(cov[..., 2] > 0)
    assert np.all(mean[0, :, 2] > 2 * tv[0, :-1]) and np.all(mean[0, :, 2] < 2 * tv[0, 1:])
    _, cov_r = orc.cone_segment_gaussians(o, dd, tv, np.full((1, 1), 0.01), np.float64)
    assert np.all(cov_r[..., :2] > 0) and np.allclose(cov_r[..., 2], cov[..., 2])    # radius only widens the null space of d


def test_philox_known_answer_vectors():
    """The counter-based generator behind the in-kernel jitter is pinned to the published known-answer vectors of
    Philox4x32-10 (Random123 kat_vectors: counter, key -> first two output words; the noise draw uses both)."""
    u32 = np.uint32
    for ctr, key, want, want1 in [((0, 0, 0, 0), (0, 0), 0x6627E8D5, 0xE169C58D),
                                  ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), 0x408F276D, 0x41C83B0E),
                                  ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), 0xD16CFE09, 0x94FDCCEB)]:
        assert int(orc.philox4x32_10(*[u32(c) for c in ctr], key[0], key[1])) == want
        w0, w1 = orc.philox4x32_10(*[u32(c) for c in ctr], key[0], key[1], words=2)
        assert (int(w0), int(w1)) == (want, want1)
    # TensorFlow's Uint32ToFloat: low 23 bits -> [1,2) - 1
    assert orc.uniform01_from_bits(np.uint32(0)) == 0.0 and orc.uniform01_from_bits(np.uint32(0xFFFFFFFF)) == np.float32(1.0 - 2.0 ** -23)


def test_perturbed_depths_stay_in_their_strata():
    """renderer.py:106-111: z = lower + (upper - lower) * u, u in [0,1) -> every depth inside its stratum, so sorted."""
    rng = np.random.default_rng(0)
    t0 = rng.uniform(1, 5, size=200).astype(np.float32)
    t = np.stack([t0, t0 + rng.uniform(0.1, 3, size=200).astype(np.float32)], -1)
    for n in (2, 3, 64):
        z0 = orc.z_values(t, n, np.float32)
        lo, up = orc.jitter_bounds(z0)
        z = orc.z_values_perturbed(t, n, 12345, np.float32)
        assert z.dtype == np.float32 and np.all(z >= lo) and np.all(z <= up) and np.all(np.diff(z, axis=-1) >= 0)
        assert not np.array_equal(z, orc.z_values_perturbed(t, n, 12346, np.float32))
        u = orc.jitter_uniforms(200, n, 12345)
        assert u.min() >= 0.0 and u.max() < 1.0


def test_noise_normals_are_standard_normal_and_keyed_by_the_global_ray():
    """raw_noise_std (renderer.py:190-192): Box-Muller of two Philox words -> N(0,1); the stream is apart from the jitter's and
    is a function of the GLOBAL ray index, so a shard draws what the whole image draws."""
    n = orc.noise_normals(4000, 64, 2024)
    assert n.dtype == np.float32 and abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
    assert abs((n ** 3).mean()) < 0.05 and abs((n ** 4).mean() - 3) < 0.1 and np.abs(n).max() < 6.0    # sqrt(-2 ln 1e-7) = 5.68
    assert np.abs(orc.noise_normals(100, 8, 2024, dtype=np.float64) - n[:100, :8]).max() < 1e-5
    whole = orc.noise_normals(96, 5, 7)
    for rank in range(4):                                     # rows of 8 pixels dealt round-robin over 4 ranks
        k = np.arange(24)
        pix = rank * 8 + (k // 8) * 32 + k % 8
        assert np.array_equal(orc.noise_normals(24, 5, 7, (rank * 8, 8, 32)), whole[pix])
        assert np.array_equal(orc.jitter_uniforms(24, 5, 7, (rank * 8, 8, 32)), orc.jitter_uniforms(96, 5, 7)[pix])
    assert not np.array_equal(orc.noise_normals(4, 4, 7), orc.noise_normals(4, 4, 8))
    # noise enters before the relu (renderer.py:195): sigma = -1 with noise +3 is as dense as sigma = 2
    col = np.zeros((1, 2, 3)); z = np.array([[1., 2.]]); d = np.array([[0., 0., 1.]])
    a = orc.map_model_output(col, np.array([[-1., -1.]]), z, d, False, (1, 1, 1), noise=np.array([[3., 0.]]), dtype=np.float64)
    b = orc.map_model_output(col, np.array([[2., -1.]]), z, d, False, (1, 1, 1), dtype=np.float64)
    assert np.array_equal(a[1], b[1]) and a[1][0] > 0.8


def test_torch_cpu_baseline_port_agrees_with_the_oracle():
    """oracle/torch_cpu.py (bench.py's cpu_baseline) computes the same render as the numpy restatement."""
    torch = pytest.importorskip("torch")
    from oracle import torch_cpu
    from nerf_tex_amd import synthetic
    for family, S in (("carpet", 33), ("grass_filtered", 48)):
        fam = synthetic.FAMILIES[family]
        spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
        w = orc.split_blob(spec, synthetic.synthetic_weights(orc.layer_table(spec), seed=0, dense_media=True))
        ro, rd, t, cone = synthetic.all_hit_rays(150, fam["b_0"], fam["b_1"], fam["cam"])
        params = np.asarray([fam["params"]], np.float32)
        ref = orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], S, False, (1, 1, 1.), fam["blur_idx"], False,
                                dtype=np.float64)
        c, a = torch_cpu.renderer_call(w, spec, ro, rd, t, params[0], cone, S, fam["blur_idx"], render_chunk=64, net_chunk=1000)
        got = np.concatenate([c, a[:, None]], -1)
        want = np.concatenate([ref["color_pred"][0], ref["alpha_pred"][0][:, None]], -1)
        assert orc.rel_linf(got, want) <= 2e-4          # float32 port vs float64 truth, dense-media weights
    assert "BLAS_INFO" in torch_cpu.blas_backend()


def test_layer_order_follows_keras_graph_depth_rule():
    """`layer_table` claims to be `tf.keras.Model.get_weights()` order.  TensorFlow cannot run here, so the rule is restated:
    keras/engine/functional.py `_map_graph_network` (TF 2.4) visits the graph depth-first from `outputs` in order (giving every
    layer a traversal index at its FIRST visit, before its inputs), assigns each layer depth = longest path to an output, and
    lists layers by decreasing depth, ties by traversal index.  Applied to the graph model.py:58-125 builds, with
    outputs=[color_outputs, alpha_outputs] (model.py:125), it must give the order of `layer_table`."""
    def order_of(kind, color_depth, depth=8, skips=(4,), param_depth=0, n_parameters=(1, 6)):
        inputs_of = {}                                    # layer -> list of input layers, as model.py wires them
        def dense(name, src):
            inputs_of[name] = [src]; return name
        for n in ("pos", "dir", "params"):
            inputs_of[n] = []
        inputs_of["pos_ff"], inputs_of["dir_ff"] = ["pos"], ["dir"]                         # FourierFeatures layers (weightless)
        inputs_of["pos_map"], inputs_of["dir_map"] = ["pos_ff"], ["dir_ff"]
        if kind == "ParamNerf":
            for branch, n, target in (("geo", n_parameters[0], "pos_map"), ("app", n_parameters[1], "dir_map")):
                if n > 0:                                                                   # model.py:88-93 / 96-101
                    inputs_of[f"slice_{branch}"] = ["params"]; inputs_of[f"ff_{branch}"] = [f"slice_{branch}"]
                    b = f"ff_{branch}"
                    for i in range(param_depth):
                        b = dense(f"param_{branch}{i}", b)
                    inputs_of[f"cat_{target}"] = [target, b]                                # tf.concat([pos_inputs_map, branch], -1)
        pos_map = "cat_pos_map" if "cat_pos_map" in inputs_of else "pos_map"
        dir_map = "cat_dir_map" if "cat_dir_map" in inputs_of else "dir_map"
        h = pos_map
        for i in range(depth):                                                            # model.py:104-108
            h = dense(f"trunk{i}", h)
            if i in skips:
                inputs_of[f"skipcat{i}"] = [pos_map, h]; h = f"skipcat{i}"
        alpha = dense("alpha", h)                                                         # :111
        f = dense("feature", h)                                                           # :114
        inputs_of["dircat"] = [dir_map, f]; h = "dircat"                                  # :115
        if kind == "ParamNerf":
            for i in range(color_depth):                                                  # :118-119
                h = dense(f"color_hidden{i}", h)
        h = dense("color_half", h)                                                        # :122
        color = dense("color", h)                                                         # :123
        outputs = [color, alpha]                                                          # :125
        index, depth_of = {}, {}
        def visit(layer):
            if layer not in index:
                index[layer] = len(index)
            for src in inputs_of[layer]:
                visit(src)
        for o in outputs:
            visit(o)
        def longest(layer, d):
            if depth_of.get(layer, -1) >= d:
                return
            depth_of[layer] = d
            for src in inputs_of[layer]:
                longest(src, d + 1)
        for o in outputs:
            longest(o, 0)
        layers = sorted(index, key=lambda l: (-depth_of[l], index[l]))
        weighted = lambda l: l.startswith(("trunk", "color", "param_")) or l in ("alpha", "feature")
        return [l for l in layers if weighted(l)]

    assert order_of("ParamNerf", 1) == [n for n, _, _ in orc.layer_table(orc.ModelSpec(kind="ParamNerf", n_parameters=(1, 6)))]
    assert order_of("Nerf", 0) == [n for n, _, _ in orc.layer_table(orc.ModelSpec(kind="Nerf", n_parameters=(0, 0)))]
    assert order_of("ParamNerf", 1)[-2:] == ["color", "alpha"] and order_of("ParamNerf", 1)[8] == "feature"
    # ... and for the architectures of the flex family (depth, skips, color_depth other than 8 / [4] / 1): the same rule, the same table
    for kind, cd, depth, skips in (("ParamNerf", 0, 8, (4,)), ("ParamNerf", 3, 8, (4,)), ("ParamNerf", 1, 3, (0, 1)), ("ParamNerf", 2, 10, (3, 6)),
                                   ("ParamNerf", 1, 1, ()), ("ParamNerf", 4, 24, tuple(range(0, 23, 2))), ("Nerf", 0, 5, (1, 2)), ("Nerf", 0, 6, ())):
        spec = orc.ModelSpec(kind=kind, n_parameters=(1, 6) if kind == "ParamNerf" else (0, 0), depth=depth, skips=skips, color_depth=cd)
        assert order_of(kind, cd, depth, skips) == [n for n, _, _ in orc.layer_table(spec)], (kind, cd, depth, skips)
    # ... and with param_depth > 0 (model.py:88-101): the Dense layers of the geometry branch sit above the whole trunk, those of the
    # appearance branch interleave with the trunk layers of the same graph depth and come first there (the traversal reaches
    # dir_map before the feature layer)
    for cd, depth, skips, pd, npar in ((1, 8, (4,), 1, (1, 6)), (1, 3, (0,), 2, (1, 6)), (0, 8, (4,), 3, (2, 3)), (2, 1, (), 4, (1, 4)),
                                       (1, 5, (1, 3), 2, (0, 3)), (1, 5, (1, 3), 2, (3, 0)), (4, 24, (), 4, (4, 8)), (1, 8, (4,), 2, (0, 0))):
        spec = orc.ModelSpec(kind="ParamNerf", n_parameters=npar, depth=depth, skips=skips, color_depth=cd, param_depth=pd)
        assert order_of("ParamNerf", cd, depth, skips, pd, npar) == [n for n, _, _ in orc.layer_table(spec)], (cd, depth, skips, pd, npar)
    t = dict((n, (i, o)) for n, i, o in orc.layer_table(orc.ModelSpec(param_depth=2)))
    assert t["param_geo0"] == (9, 128) and t["param_geo1"] == (128, 128) and t["param_app0"] == (54, 128) and t["trunk0"] == (63 + 128, 256) \
        and t["trunk5"] == (256 + 63 + 128, 256) and t["color_hidden0"] == (256 + 27 + 128, 256)
