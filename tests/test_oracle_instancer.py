"""The instancer restatement (oracle/instancer_oracle.py) against hand-computed cases of instancer.cpp:751-1037 and against
properties every output must have.  The reference ships no vectors for its instancer and cannot be built here (Embree): these
are known answers worked out from its code, not outputs of it -- parity stays unpinned (see the oracle's header)."""

import numpy as np
import pytest

from oracle import instancer_oracle as io

F = np.float32


def translate(x=0., y=0., z=0., s=1.):
    m = np.eye(4, dtype=F)
    m[:3, :3] *= s
    m[:3, 3] = (x, y, z)
    return m


def run(spec, o, d, S, h, u=0.5, params=None, uc=None):
    o = np.asarray(o, F).reshape(-1, 3); d = np.asarray(d, F).reshape(-1, 3)
    n = o.shape[0]
    params = np.zeros((n, spec.n_parameters), F) if params is None else np.asarray(params, F).reshape(n, -1)
    uo = np.full(n, u, F)
    uc = np.full((n, S), 0.0, F) if uc is None else np.asarray(uc, F).reshape(n, S)
    return io.get_model_input(spec, o, d, params, S, h, uo, uc)


UNIT = dict(b_0=[-1, -1, -1], b_1=[1, 1, 1])


def test_one_box_known_answer():
    spec = io.make_spec(transformations=[translate()], **UNIT)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(spec, [0, 0, -5], [0, 0, 1], 8, 0.5)
    assert hit.tolist() == [True]
    # crossings at t = 4 and 6: 2 / 0.5 = 4 steps, offset 0.5 * 0.5
    assert np.array_equal(dists[0], F([.5, .5, .5, .5, 0, 0, 0, 0]))
    assert np.array_equal(t[0], F([4.25, 4.75, 5.25, 5.75, 0, 0, 0, 0]))
    assert np.array_equal(pts[0, :4], F([[0, 0, -.75], [0, 0, -.25], [0, 0, .25], [0, 0, .75]]))
    assert np.array_equal(pts[0, 4:], np.zeros((4, 3), F))                  # instancer.pyx:44
    assert np.array_equal(rd[0], np.tile(F([0, 0, 1]), (8, 1)))
    assert np.array_equal(w[0], np.ones(8, F)) and np.array_equal(iid[0], np.zeros(8, np.int32))
    assert dens[0, 0] == 0 and np.array_equal(color[0, 0], np.zeros(3, F))  # no mesh: the closing sample is empty (:1024-1026)


def test_miss_keeps_the_defaults():
    spec = io.make_spec(transformations=[translate()], **UNIT)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(spec, [[5, 5, -5], [0, 0, 5]], [[0, 0, 1], [0, 0, 1]], 4, 0.5)
    assert hit.tolist() == [False, False]                                   # beside the box; behind the origin (t < 0)
    assert not dists.any() and not t.any() and not pts.any() and (w == 1).all() and not iid.any()


def test_origin_inside_a_box_toggles_at_the_exit_only():
    # one face crossing (the exit at t = 1): the toggle of instancer.cpp:812-824 ENTERS there and never leaves, so no length
    spec = io.make_spec(transformations=[translate()], **UNIT)
    *_, dists, color, dens, w, iid, hit, pm = run(spec, [0, 0, 0], [0, 0, 1], 4, 0.5)
    assert hit.tolist() == [True] and not dists.any()


def test_scaled_patch_maps_points_and_keeps_t_in_world_units():
    spec = io.make_spec(transformations=[translate(s=0.5)], **UNIT)          # a box of half-size 0.5 around the origin
    rd, pts, t, dists, *_ = run(spec, [0, 0, -5], [0, 0, 1], 4, 0.25)
    assert np.array_equal(dists[0], F([.25, .25, .25, .25]))
    assert np.array_equal(t[0], F([4.625, 4.875, 5.125, 5.375]))
    assert np.array_equal(pts[0, :, 2], F([-.75, -.25, .25, .75]))           # world z * 2
    assert np.array_equal(rd[0], np.tile(F([0, 0, 1]), (4, 1)))              # directions are normalised, not scaled (:136)


def test_two_disjoint_boxes_skip_the_gap():
    spec = io.make_spec(transformations=[translate(), translate(z=5)], **UNIT)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(spec, [0, 0, -5], [0, 0, 1], 16, 0.5)
    assert np.array_equal(dists[0, :8], np.full(8, .5, F)) and not dists[0, 8:].any()      # [4,6] and [9,11]: 4 / 0.5
    assert np.array_equal(t[0, :8], F([4.25, 4.75, 5.25, 5.75, 9.25, 9.75, 10.25, 10.75]))
    assert iid[0, :8].tolist() == [0] * 4 + [1] * 4
    assert np.array_equal(pts[0, 4:8, 2], F([-.75, -.25, .25, .75]))         # in the second patch's coordinates


def test_overlap_nearest_and_random():
    tr = [translate(), translate(z=1)]                                       # [4,6] and [5,7]: union 3 long, overlap [5,6]
    near = io.make_spec(transformations=tr, instance_sampling_method="nearest", **UNIT)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(near, [0, 0, -5], [0, 0, 1], 8, 0.5)
    assert np.array_equal(t[0, :6], F([4.25, 4.75, 5.25, 5.75, 6.25, 6.75]))
    # world z = -0.75 ... 1.75: the anchor of patch 0 is z = 0, of patch 1 z = 1; 0.25 -> 0, 0.75 -> 1
    assert iid[0, :6].tolist() == [0, 0, 0, 1, 1, 1] and (w[0] == 1).all()
    rnd = io.make_spec(transformations=tr, instance_sampling_method="random", **UNIT)
    uc = np.zeros((1, 8), F); uc[0, 3] = 0.75
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(rnd, [0, 0, -5], [0, 0, 1], 8, 0.5, uc=uc)
    assert iid[0, :6].tolist() == [0, 0, 0, 1, 1, 1]                         # steps 2, 3 lie in both: draw 0 -> first, 0.75 -> second
    assert w[0, :6].tolist() == [1, 1, 2, 2, 1, 1]                           # density_weight = number of patches (:676)


def test_nearest_blend_weights():
    tr = [translate(), translate(z=1)]
    spec = io.make_spec(transformations=tr, instance_sampling_method="nearest_blend", **UNIT)
    uc = np.zeros((1, 8), F)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(spec, [0, 0, -5], [0, 0, 1], 8, 0.5, uc=uc)
    # step 2 at world z = 0.25: distances 0.25 / 0.75, transition 0.2 -> weights (0.2, 0): patch 0 with probability 1
    assert iid[0, 2] == 0 and w[0, 2] == 1
    # a point 0.05 off the middle: weights (0.2, 0.1) -> draw 0.9 * 0.3 = 0.27 >= 0.2 picks patch 1 with probability 1/3
    uo = np.asarray([0.1], F)                                                 # step 2 at t = 5.05 -> z = 0.05 ... use a finer march
    out = io.get_model_input(spec, F([[0, 0, -5]]), F([[0, 0, 1]]), np.zeros((1, 0), F), 64, 0.05, np.asarray([0.0], F),
                             np.full((1, 64), 0.9, F))
    k = int(np.argmin(np.abs(out[2][0] - F(5.45))))                           # world z = 0.45: distances 0.45 / 0.55
    assert out[7][0, k] == 1 and abs(out[6][0, k] - 3.0) < 1e-4


def test_buffer_too_small_and_fewer_than_one_step():
    spec = io.make_spec(transformations=[translate()], **UNIT)
    *_, dists, color, dens, w, iid, hit, pm = run(spec, [0, 0, -5], [0, 0, 1], 3, 0.5)
    assert np.array_equal(dists[0], F([.5, .5, (F(.5) + F(2)) - F(3) * F(.5)]))             # 4 steps needed, 3 fit: the last takes the rest (:853)
    rd, pts, t, dists, *_ = run(spec, [0, 0, -5], [0, 0, 1], 3, 4.0)
    assert np.array_equal(dists[0], F([2, 0, 0])) and np.array_equal(t[0], F([5, 0, 0]))    # one sample, offset 0.5 * length (:846-849)


def test_mesh_ends_the_ray_with_an_opaque_sample():
    quad_v = [[-2, -2, .5], [2, -2, .5], [2, 2, .5], [-2, 2, .5]]
    spec = io.make_spec(transformations=[translate()], mesh=(quad_v, [[0, 1, 2], [0, 2, 3]]), **UNIT)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(spec, [[0.3, 0.2, -5], [5, 5, -5]], [[0, 0, 1]] * 2, 8, 0.5)
    assert np.array_equal(dists[0], F([.5, .5, .5, 0, 0, 0, 0, 0]))          # [4, 5.5]
    assert dens[:, 0].tolist() == [1, 0] and hit.tolist() == [True, False]
    # a mesh in front of the box: hit, nothing to march, opaque
    spec2 = io.make_spec(transformations=[translate()], mesh=([[-2, -2, -3], [2, -2, -3], [0, 2, -3]], [[0, 1, 2]]), **UNIT)
    *_, dists, color, dens, w, iid, hit, pm = run(spec2, [0, 0, -5], [0, 0, 1], 8, 0.5)
    assert hit.tolist() == [True] and not dists.any() and dens[0, 0] == 1


def test_lights():
    rot = np.eye(4, dtype=F); rot[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]          # patch x -> world y
    spec = io.make_spec(transformations=[rot], textures=["", "light"], **UNIT)
    assert (spec.n_parameters, spec.light_dir_idx, spec.light_strength_idx) == (4, 1, -1)
    par = F([[0.7, 0, 2, 0]])                                                            # light along world +y, not normalised
    *_, pm = run(spec, [0, 0, -5], [0, 0, 1], 4, 0.5, params=par)
    assert np.allclose(pm[0, :, 0], 0.7) and np.allclose(pm[0, :, 1:], [1, 0, 0])        # world y = patch x, normalised (:562)
    spec = io.make_spec(transformations=[translate()], textures=["point", ""], **UNIT)
    assert (spec.n_parameters, spec.light_dir_idx, spec.light_strength_idx) == (5, 1, 0)
    par = F([[10, 0, 0, 3, 0.3]])                                                        # strength 10 at (0, 0, 3)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run(spec, [0, 0, -5], [0, 0, 1], 4, 0.5, params=par)
    z = t[0] - 5                                                                         # world z of the samples
    assert np.allclose(pm[0, :, 1:4], [0, 0, 1])                                         # towards the light
    assert np.allclose(pm[0, :, 0], 10 / (4 * np.pi * (3 - z) ** 2 + 1e-6), rtol=1e-6)
    assert np.allclose(pm[0, :, 4], 0.3)


def test_mean_distance():
    spec = io.make_spec(transformations=[translate()], use_mean_distance=True, **UNIT)
    rd, pts, t, dists, *_ = run(spec, [0, 0, -5], [0, 0, 1], 8, 0.5)
    mu = t[0, :4].astype(np.float64)
    want = mu + 2 * mu * 0.25 / (3 * mu * mu + 0.25)                                     # instancer.cpp:746-748
    assert np.array_equal(t[0, :4], F([4.25, 4.75, 5.25, 5.75]))                         # t keeps t_mu (:885)
    assert np.allclose(pts[0, :4, 2], want - 5, atol=1e-6)


def test_is_shadowed_known_answers():
    """instancer.cpp:591-602 with the filter of :543-554: the top face of a patch box from outside, its bottom face either way, a
    mesh from its front; side faces and back faces let the light through."""
    spec = io.make_spec(transformations=[translate()], **UNIT)
    up, down = F([0, 0, 1]), F([0, 0, -2.5])
    assert not io.is_shadowed(spec, F([0, 0, 0]), up)            # inside: leaves through the top from INSIDE
    assert io.is_shadowed(spec, F([0, 0, 2]), down)              # above, light from below the box: enters through the top
    assert io.is_shadowed(spec, F([0, 0, -2]), up)               # below: crosses the bottom face
    assert io.is_shadowed(spec, F([0, 0, 0]), down)              # inside, looking down: the bottom face counts either way
    assert not io.is_shadowed(spec, F([3, 0, 0]), up)            # beside
    assert not io.is_shadowed(spec, F([3, 0, 0]), F([-1, 0, 0]))  # through the side faces only
    assert not io.is_shadowed(spec, F([0, 0, 102]), F([0, 0, -1]))          # the top face lies beyond tfar = 100
    tri = ([[-1, -1, 2], [1, -1, 2], [0, 1, 2]], [[0, 1, 2]])    # counter-clockwise seen from +z: Ng = +z
    spec = io.make_spec(transformations=[translate(z=-50)], mesh=tri, **UNIT)
    assert io.is_shadowed(spec, F([0, 0, 3]), F([0, 0, -1]))     # onto the front
    assert not io.is_shadowed(spec, F([0, 0, 0]), up)            # onto the back
    assert not io.is_shadowed(spec, F([5, 0, 3]), F([0, 0, -1]))


def test_auxiliary_mesh_is_shaded():
    """shadeMesh (instancer.cpp:716-743) on the closing sample: albedo 0.8 * min(max(n . l, 0) + 0.2, 1), 0.2 ambient only when the
    point is shadowed or faces away; the instancer mesh itself stays black."""
    quad_v = [[-2, -2, .5], [2, -2, .5], [2, 2, .5], [-2, 2, .5]]
    up = [[0, 0, 1]] * 4
    mesh = (quad_v, [[0, 1, 2], [0, 2, 3]], up, [1, 1])
    spec = io.make_spec(transformations=[translate(z=-30)], textures=["light"], mesh=mesh, **UNIT)
    o, d = F([[0.3, 0.2, 5]]), F([[0, 0, -1]])                    # from above onto the lit side
    light = F([[0, 3, 4]])                                       # n . l = 0.8
    *_, color, dens, w, iid, hit, pm = io.get_model_input(spec, o, d, light, 4, 0.5, F([0.5]), np.zeros((1, 4), F))
    assert dens[0, 0] == 1 and np.allclose(color[0, 0], 0.8 * 1.0)                      # min(0.8 + 0.2, 1)
    *_, color, dens, w, iid, hit, pm = io.get_model_input(spec, o, d, F([[0, 0.6, 0.3]]), 4, 0.5, F([0.5]), np.zeros((1, 4), F))
    assert np.allclose(color[0, 0], 0.8 * (0.3 / np.hypot(0.6, 0.3) + 0.2), atol=1e-6)
    *_, color, dens, w, iid, hit, pm = io.get_model_input(spec, o, d, F([[0, 0, -1]]), 4, 0.5, F([0.5]), np.zeros((1, 4), F))
    assert np.allclose(color[0, 0], 0.8 * 0.2)                                          # lit from behind: ambient only
    # a patch box over the point: its bottom face shadows it
    spec = io.make_spec(transformations=[translate(z=3)], textures=["light"], mesh=mesh, **UNIT)
    *_, color, dens, w, iid, hit, pm = io.get_model_input(spec, F([[0.3, 0.2, 1.5]]), d, F([[0, 0.3, 4]]), 4, 0.5, F([0.5]), np.zeros((1, 4), F))
    assert np.allclose(color[0, 0], 0.8 * 0.2) and dens[0, 0] == 1
    # kind 0 = the instancer mesh: black
    spec = io.make_spec(transformations=[translate(z=-30)], textures=["light"], mesh=(quad_v, mesh[1], up, [0, 0]), **UNIT)
    *_, color, dens, w, iid, hit, pm = io.get_model_input(spec, o, d, light, 4, 0.5, F([0.5]), np.zeros((1, 4), F))
    assert dens[0, 0] == 1 and not color.any()


def random_scene(seed, k=12, method="nearest", textures=(), mesh=False):
    rng = np.random.default_rng(seed)
    tr = []
    for _ in range(k):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m = np.eye(4)
        m[:3, :3] = q * rng.uniform(0.2, 0.5)
        m[:3, 3] = rng.uniform(-1, 1, size=3) * [1, 1, 0.2]
        tr.append(m.astype(F))
    msh = None
    if mesh:
        msh = ([[-3, -3, -.1], [3, -3, -.1], [3, 3, -.1], [-3, 3, -.1]], [[0, 1, 2], [0, 2, 3]])
    return io.make_spec([-1, -1, -.2], [1, 1, 1], tr, textures=textures, instance_sampling_method=method, mesh=msh)


def random_rays(seed, n):
    rng = np.random.default_rng(seed + 1000)
    o = rng.normal(size=(n, 3)); o = 4 * o / np.linalg.norm(o, axis=1, keepdims=True); o[:, 2] = np.abs(o[:, 2]) + .5
    target = rng.uniform(-1, 1, size=(n, 3)) * [1.2, 1.2, .3]
    d = target - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(F), d.astype(F)


@pytest.mark.parametrize("method", ["random", "nearest", "nearest_blend"])
def test_properties_on_random_scenes(method):
    spec = random_scene(3, method=method, mesh=True)
    o, d = random_rays(3, 40)
    S, h = 96, 0.02
    uo = io.offset_uniforms(40, 7); uc = io.choice_uniforms(40, S, 7)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = io.get_model_input(spec, o, d, np.zeros((40, 0), F), S, h, uo, uc)
    assert hit.any() and (dists > 0).any()
    for r in range(40):
        n_in = int((dists[r] > 0).sum())
        assert (dists[r, :n_in] > 0).all() and not dists[r, n_in:].any()                 # the in-patch steps lead the row
        emitted = int((t[r] > 0).sum())
        assert emitted <= n_in and (np.diff(t[r, :emitted]) > 0).all()
        # every emitted sample lies inside the box of the patch it was given to (up to rounding at the faces)
        p = pts[r, :emitted]
        assert (p >= spec.b_0 - 1e-4).all() and (p <= spec.b_1 + 1e-4).all()
        world = o[r] + t[r, :emitted, None] * d[r]
        back = np.einsum("kij,kj->ki", spec.inv[iid[r, :emitted]][:, :3, :3], world) + spec.inv[iid[r, :emitted]][:, :3, 3]
        assert np.allclose(back, p, atol=1e-5)
        assert np.allclose(np.linalg.norm(rd[r, :emitted], axis=-1), 1, atol=1e-5)
        assert (w[r, :emitted] >= 1).all()
        if dens[r, 0] == 1:                                                              # ended at the mesh: nothing behind it
            assert (world[:, 2] >= -.1 - 1e-4).all()


def test_draws_are_keyed_by_the_global_ray():
    a = io.offset_uniforms(8, 5, ray_index=(16, 4, 8))
    b = io.offset_uniforms(32, 5)
    assert np.array_equal(a, b[[16, 17, 18, 19, 24, 25, 26, 27]])
    c = io.choice_uniforms(2, 5, 5, ray_index=(3, 2, 2)); e = io.choice_uniforms(5, 5, 5)
    assert np.array_equal(c, e[3:5]) and (0 <= e).all() and (e < 1).all()
    assert not np.array_equal(io.offset_uniforms(4, 5), io.offset_uniforms(4, 6))


def test_prepare_instances_matches_add_instance():
    rng = np.random.default_rng(0)
    m = np.eye(4); m[:3, :3] = rng.normal(size=(3, 3)); m[:3, 3] = rng.normal(size=3)
    inv, dir_t, org = io.prepare_instances([m])
    assert np.allclose(inv[0] @ m, np.eye(4), atol=1e-5)
    assert np.allclose(org[0], m[:3, 3])
    cols = m[:3, :3].T
    assert np.allclose(dir_t[0], cols / np.linalg.norm(cols, axis=1, keepdims=True), atol=1e-6)   # :131
    with pytest.raises(ValueError):
        io.parse_textures(["meshes/smooth_checkerboard.png"])
