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
    with pytest.raises(OSError):                                                         # an image that is not there (the reference's are LFS pointers)
        io.parse_textures(["meshes/no_such_texture.png"])


# ------------------------------------------------------------------------------------------------------------------------------
# image textures (instancer.cpp:34-50, 605-667, 725-733, 910-927, 989-998): known answers worked out by hand
# ------------------------------------------------------------------------------------------------------------------------------

def test_texture_layout_known_answer():
    # stb's pixels [height 2, width 3], one channel, rows top-down:   10 20 30 / 40 50 60
    tex = io.texture_from_pixels(np.asarray([[10, 20, 30], [40, 50, 60]], np.uint8))
    assert len(tex) == 1 and tex[0].shape == (3, 2)                                      # (width, height): rows = x, columns = y
    # column-major Map(width, height): (x, y) = pixel row y, column x; rowwise().reverse() turns y over: column 0 = the BOTTOM image row
    assert np.array_equal(tex[0], (F([[40, 10], [50, 20], [60, 30]]) / F(255)))
    rgb = io.texture_from_pixels(np.arange(24, dtype=np.uint8).reshape(2, 4, 3))
    assert len(rgb) == 3 and rgb[1][0, 1] == F(1) / F(255) and rgb[2][3, 0] == F(23) / F(255)


def test_interpolate2d_known_answers():
    y = F([[0, 1], [2, 3], [4, 5]])                                                      # rows 3 (x), cols 2 (y)
    assert io.interpolate2d(F([0, 0]), y) == 0 and io.interpolate2d(F([1, 1]), y) == 5   # (1,1): idx + 1 would leave the matrix; its weight is 0
    assert io.interpolate2d(F([0.5, 0]), y) == 2                                         # x * (rows - 1) = 1: exactly on row 1
    assert io.interpolate2d(F([0.25, 0.5]), y) == F(1.5)                                 # between rows 0 / 1 (w 0.5) and columns 0 / 1 (w 0.5): (0 + 1 + 2 + 3) / 4
    assert io.interpolate2d(F([0.75, 1.0]), y) == F(4.0)                                 # rows 1 / 2 at w 0.5, column 1 (w 0): (3 + 5) / 2
    # the weights are x - floor(x), the indices truncate: below zero they part ways (-0.25 -> index 0, weight 0.75)
    assert io.interpolate2d(F([-0.125, 0]), y) == F(0.75) * 2


def test_closest_point_triangle_regions():
    a, b, c = F([0, 0, 0]), F([2, 0, 0]), F([0, 2, 0])
    cp = lambda p: io.closest_point_triangle(F(p), a, b, c)
    for p, q, w in [([-1, -1, 3], a, [1, 0, 0]), ([3, -1, 0], b, [0, 1, 0]), ([-1, 4, 1], c, [0, 0, 1]),          # the three corners
                    ([1, -2, 0], [1, 0, 0], [.5, .5, 0]), ([-3, .5, 0], [0, .5, 0], [.75, 0, .25]),              # edges ab, ac
                    ([2, 2, 5], [1, 1, 0], [0, .5, .5]), ([.5, .5, -7], [.5, .5, 0], [.5, .25, .25])]:           # edge bc, inside
        got_q, got_w = cp(p)
        assert np.allclose(got_q, q, atol=1e-7) and np.allclose(got_w, w, atol=1e-7), (p, got_q, got_w)
    v = F([[0, 0, 0], [2, 0, 0], [0, 2, 0], [2, 2, 0]]); f = np.asarray([[0, 1, 2], [1, 3, 2]])
    assert io.closest_point_on_mesh(v, f, F([1.5, 1.5, 1]), F(2))[0] == 1
    assert io.closest_point_on_mesh(v, f, F([1.5, 1.5, 1]), F(1))[0] is None            # d < radius is strict (:222)
    k, w = io.closest_point_on_mesh(v, f, F([1, 1, .5]), F(2))                           # on the shared edge: both at 0.5, the lower primID stays
    assert k == 0 and np.allclose(w, [0, .5, .5])


def textured_sheet_spec(textures, images, **kw):
    """Patches 0 and 1 side by side on the sheet z = 0 (|x|, |y| <= 2, texture coordinates (x + 2) / 4, (y + 2) / 4)."""
    v = F([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]]); f = [[0, 1, 2], [0, 2, 3]]
    uv = (v[:, :2] + 2) / 4
    return io.make_spec([-1, -1, 0], [1, 1, 1], [translate(x=-1), translate(x=1)], textures=textures, images=images,
                        instancer_mesh=(v, f, uv), mesh=(v, f), patch_scale=1.0, **kw)


def test_parameter_textures_known_answers():
    ramp = np.tile(F(np.linspace(0, 1, 5))[:, None], (1, 3))                             # 5 x 3 texels: value = u, constant in v
    const = np.full((2, 2), F(0.5))
    spec = textured_sheet_spec(["ramp", "", "light"], {"ramp": [ramp]})
    assert (spec.n_parameters, spec.light_dir_idx, spec.tex_idx) == (5, 2, [0]) and np.isclose(spec.patch_max_extent, np.sqrt(3))
    par = F([2.0, 7.0, 0, 0, 1])
    # straight above (x, y): u = (x + 2) / 4 -> parameter 0 = 2 u, the others as given
    for x in (-1.5, 0.0, 0.7):
        assert np.allclose(io.get_parameters(spec, F([x, 0.3, 0.8]), par), [2 * (x + 2) / 4, 7, 0, 0, 1], rtol=1e-6, atol=0)
    assert np.array_equal(io.get_parameters(spec, F([0, 0, 1.8]), par), par)             # further than patch_max_extent = sqrt(3): untouched (:657)
    assert np.allclose(io.get_parameters(spec, F([3, 0, 0.5]), par), [2.0, 7, 0, 0, 1], rtol=1e-6, atol=0)   # beside the sheet: its rim, u = 1
    # the reference's indexing (:656-662): file i multiplies its FIRST parameter by channel matrix i of the whole list
    rgb = [np.full((2, 2), F(c)) for c in (0.5, 0.25, 0.125)]
    spec = textured_sheet_spec(["rgb", "c"], {"rgb": rgb, "c": [const]})
    assert (spec.n_parameters, spec.tex_idx, len(spec.textures)) == (4, [0, 3], 4)
    assert np.array_equal(io.get_parameters(spec, F([0, 0, .5]), F([1, 1, 1, 1])), F([0.5, 1, 1, 0.25]))   # red on 0; GREEN of file 0 on 3; 1, 2 untouched


def test_texture_samples_along_the_ray():
    ramp = np.tile(F(np.linspace(0, 1, 5))[:, None], (1, 3))
    # a ray from x = -3 to the right, 0.5 above the sheet: inside the two patches for t in [1, 5] (one segment)
    o, d = F([[-3, 0, .5]]), F([[1, 0, 0]])
    par = F([[2.0, 7.0, 0, 0, 1]])
    per_step = textured_sheet_spec(["ramp", "", "light"], {"ramp": [ramp]}, n_texture_samples=100000)
    rd, pts, t, dists, color, dens, w, iid, hit, pm = io.get_model_input(per_step, o, d, par, 16, 0.5, F([0.5]), np.zeros((1, 16), F))
    assert np.array_equal(t[0, :8], F([1.25 + .5 * s for s in range(8)])) and not t[0, 8:].any()
    x = -3 + t[0, :8]
    assert np.allclose(pm[0, :8, 0], 2 * (x + 2) / 4, rtol=1e-6, atol=0) and (pm[0, :8, 1] == 7).all()     # a query per step (:926)
    assert np.array_equal(pm[0, 8:], np.tile(par, (8, 1)))                               # behind the last step: the row as given
    # 2 * 4 = 8 samples < 16: interpolated (:911-923).  The crossings at t = 3 sort (3, patch 0) before (3, patch 1): patch 0 is left before
    # patch 1 is entered, so the ray has TWO segments [1, 3] and [3, 5], each with max(4, 8 * 2 / 4) = 4 samples 2 / 3 apart; the ramp is
    # linear, so the blend gives 2 u again (rounded)
    interp = textured_sheet_spec(["ramp", "", "light"], {"ramp": [ramp]}, n_texture_samples=2, min_texture_samples=4)
    out = io.get_model_input(interp, o, d, par, 16, 0.5, F([0.5]), np.zeros((1, 16), F))
    assert np.allclose(out[9][0, :8, 0], 2 * (x + 2) / 4, atol=2e-6) and np.array_equal(out[2], t)
    sl = F(2) / F(3)
    for s in range(8):                                                                   # ... and exactly s0 * (1 - w) + s1 * w in EVERY column
        te = F(1) if s < 4 else F(3)
        k = int((t[0, s] - te) / sl)
        t0, t1 = te + F(k) * sl if k else te, te + F(k + 1) * sl
        s0 = io.get_parameters(interp, (o[0] + t0 * d[0]).astype(F), par[0]); s1 = io.get_parameters(interp, (o[0] + t1 * d[0]).astype(F), par[0])
        wgt = (t[0, s] - t0) / sl
        want = s0 * (F(1) - wgt) + s1 * wgt
        want[2:5] = [0, 0, 1]                                                            # the light direction is written afterwards (:943-949)
        assert np.array_equal(out[9][0, s], want), (s, out[9][0, s], want)
    # explicit transformations without DistributeInstancesOnMesh: the textures are loaded, counted and never applied (:911)
    plain = io.make_spec([-1, -1, 0], [1, 1, 1], [translate(x=-1), translate(x=1)], textures=["ramp", "", "light"], images={"ramp": [ramp]})
    assert plain.n_parameters == 5
    out = io.get_model_input(plain, o, d, par, 16, 0.5, F([0.5]), np.zeros((1, 16), F))
    assert (out[9][0, :, 0] == 2).all()


def test_auxiliary_albedo_and_the_second_triangle():
    quad_v = F([[-2, -2, .5], [2, -2, .5], [2, 2, .5], [-2, 2, .5]]); faces = [[0, 1, 2], [0, 2, 3]]
    up = [[0, 0, 1]] * 4
    uv = (quad_v[:, :2] + 2) / 4
    ramp = np.tile(F(np.linspace(0, 1, 5))[:, None], (1, 3))
    o, d, light = F([[1.0, 0.2, 5]]), F([[0, 0, -1]]), F([[0, 3, 4]])                    # n . l = 0.8 -> shade min(0.8 + 0.2, 1) = 1
    for tex, want in [([ramp], [.75] * 3), ([ramp, ramp * F(.5), ramp * F(.25)], [.75, .375, .1875]), ([ramp, ramp * F(.5)], [.75] * 3)]:
        spec = io.make_spec(transformations=[translate(z=-30)], textures=["light"], mesh=(quad_v, faces, up, [1, 1]), mesh_uv=uv,
                            mesh_tex=[0, 0], aux_textures=[tex], **UNIT)
        *_, color, dens, w, iid, hit, pm = io.get_model_input(spec, o, d, light, 4, 0.5, F([0.5]), np.zeros((1, 4), F))
        assert dens[0, 0] == 1 and np.allclose(color[0, 0], want, atol=1e-6), (color, want)    # u = 0.75; three channels or the first for all (:732)
    # the shadow filter's `primID == 1` (:553) does not ask which geometry: the second triangle of a mesh shadows from behind too
    spec = io.make_spec(transformations=[translate(z=-50)], mesh=(quad_v, faces), **UNIT)
    below_first, below_second = F([1, -1, 0]), F([-1, 1, 0])                             # under triangle 0 / under triangle 1, looking up at their backs
    assert not io.is_shadowed(spec, below_first, F([0, 0, 1])) and io.is_shadowed(spec, below_second, F([0, 0, 1]))
    assert io.is_shadowed(spec, F([1, -1, 3]), F([0, 0, -1])) and io.is_shadowed(spec, F([-1, 1, 3]), F([0, 0, -1]))
    two = io.make_spec(transformations=[translate(z=-50)], mesh=(np.concatenate([quad_v, quad_v + F([0, 0, 1])]), faces + [[4, 5, 6], [4, 6, 7]]),
                       mesh_prim=[0, 1, 0, 1], **UNIT)                                    # two meshes in one list: primIDs start over
    assert io.is_shadowed(two, F([-1, 1, 1]), F([0, 0, 1])) and not io.is_shadowed(two, F([1, -1, 1]), F([0, 0, 1]))


# ------------------------------------------------------------------------------------------------------------------------------
# the second restatement (plain C, oracle/c/ntx_instancer_oracle.c, written from instancer.cpp): element for element
# ------------------------------------------------------------------------------------------------------------------------------

NAMES = ["rays_d_map", "pts", "t", "dists", "color", "density", "density_weight", "instance_id", "hit", "params_map"]


def assert_twins_agree(spec, o, d, params, S, h, seed):
    from oracle import c_instancer as ci
    n = o.shape[0]
    uo, uc = io.offset_uniforms(n, seed), io.choice_uniforms(n, S, seed)
    a = io.get_model_input(spec, o, d, params, S, h, uo, uc)
    b = ci.get_model_input(spec, o, d, params, S, h, uo, uc)
    for name, x, y in zip(NAMES, a, b):
        assert x.shape == y.shape and x.dtype == y.dtype, (name, x.shape, y.shape, x.dtype, y.dtype)
        if not np.array_equal(x, y):
            bad = np.argwhere(x != y)
            raise AssertionError(f"{name}: {len(bad)} of {x.size} differ, first at {bad[0].tolist()}: python {x[tuple(bad[0])]!r}, C {y[tuple(bad[0])]!r}")
    return a


def sheet_mesh(n=7, extent=1.7, amp=0.07, z0=-0.12):
    xs = np.linspace(-extent, extent, n)
    x, y = np.meshgrid(xs, xs, indexing="ij")
    z = z0 + amp * np.sin(1.7 * x) * np.cos(1.3 * y)
    v = np.stack([x, y, z], -1).reshape(-1, 3).astype(F)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, e = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, e], -1)]).astype(np.int32)
    u = (x + extent) / (2 * extent); w = (y + extent) / (2 * extent)
    return v, f, np.stack([0.8 * u + 0.2 * w, 0.1 * u + 0.9 * w], -1).reshape(-1, 2).astype(F)


@pytest.mark.parametrize("seed", range(24))
def test_the_two_restatements_agree(seed):
    """Random scenes over every branch of GetModelInput -- the three patch choices, lights, mean distances, a culling mesh, shadow rays
    interpolated and per step, parameter textures interpolated and per step with one to four channels, shaded and textured auxiliary
    meshes, buffers too short -- through both restatements with the same draws: all ten buffers identical."""
    rng = np.random.default_rng(7000 + seed)
    method = ["random", "nearest", "nearest_blend"][seed % 3]
    textures = [(), ("", "light"), ("point",), ("", "", "light", "")][int(rng.integers(0, 4))]
    mean = bool(rng.integers(0, 2))
    shadows = bool(textures) and bool(rng.integers(0, 2))
    tex_mode = int(rng.integers(0, 3))
    names, images = list(textures), {}
    kw = {}
    if shadows:
        kw.update(cast_shadow_rays=True, min_shadow_samples=int(rng.integers(2, 7)), n_shadow_samples=int(rng.choice([16, 48, 100000])))
    mesh = bool(rng.integers(0, 2))
    spec0 = random_scene(300 + seed, k=int(rng.integers(3, 20)), method=method, textures=textures, mesh=mesh)
    msh = (spec0.mesh_v, spec0.mesh_f) if mesh else None
    patch_scale = 1.0
    if tex_mode:
        for q in range(int(rng.integers(1, 3))):
            key = f"#{q}"
            images[key] = io.texture_from_pixels(rng.integers(0, 256, size=(int(rng.integers(2, 9)), int(rng.integers(2, 9)), int(rng.choice([1, 3, 4]))), dtype=np.uint8))
            names.insert(int(rng.integers(0, len(names) + 1)), key)
        tm = sheet_mesh(n=int(rng.integers(3, 8)), amp=float(rng.uniform(0, 0.15)))
        patch_scale = float(rng.choice([0.3, 1.0]))
        kw.update(instancer_mesh=tm, patch_scale=patch_scale, min_texture_samples=int(rng.integers(2, 7)),
                  n_texture_samples=int(rng.choice([8, 32])) if tex_mode == 1 else 100000)
        msh = (tm[0], tm[1])
    aux = bool(textures) and bool(rng.integers(0, 2))
    if aux:                                                                      # an auxiliary quad over part of the scene, sometimes textured
        base_v = msh[0] if msh is not None else np.zeros((0, 3), F); base_f = msh[1] if msh is not None else np.zeros((0, 3), np.int32)
        z0, z1 = rng.uniform(0.3, 1.2, size=2)
        qv = F([[-1.5, -1.5, z0], [1.5, -1.5, z1], [1.5, 1.5, z1], [-1.5, 1.5, z0]]); qf = np.asarray([[0, 1, 2], [0, 2, 3]], np.int32) + len(base_v)
        nv = rng.normal(size=(4, 3)) * 0.2 + [0, 0, 1]
        nrm = np.concatenate([np.zeros((len(base_v), 3), F), (nv / np.linalg.norm(nv, axis=-1, keepdims=True)).astype(F)])
        msh = (np.concatenate([np.asarray(base_v, F).reshape(-1, 3), qv]), np.concatenate([np.asarray(base_f, np.int32).reshape(-1, 3), qf]), nrm,
               np.r_[np.zeros(len(base_f), np.uint8), np.ones(2, np.uint8)])
        kw["mesh_prim"] = np.r_[np.arange(len(base_f)), [0, 1]]
        if rng.integers(0, 2):
            kw.update(mesh_uv=np.concatenate([np.zeros((len(base_v), 2), F), rng.uniform(0, 1, size=(4, 2)).astype(F)]),
                      mesh_tex=np.r_[np.full(len(base_f), -1), [0, 0]],
                      aux_textures=[io.texture_from_pixels(rng.integers(0, 256, size=(4, 5, int(rng.choice([1, 3, 4]))), dtype=np.uint8))])
    spec = io.make_spec(spec0.b_0, spec0.b_1, None, textures=names, images=images, instance_sampling_method=method, use_mean_distance=mean, mesh=msh,
                        matrices=(spec0.inv, spec0.dir_t, spec0.origins), **kw)
    spec.patch_scale = patch_scale
    n = 24
    o, d = random_rays(300 + seed, n)
    d[3:6] = d[3:6] * F(rng.uniform(0.5, 2.0))
    S = int(rng.choice([7, 33, 64, 100]))
    h = float(rng.choice([0.01, 0.05, 0.7]))
    params = rng.uniform(0.1, 2.0, size=(n, spec.n_parameters)).astype(F)
    out = assert_twins_agree(spec, o, d, params, S, h, seed)
    assert out[8].any()


def test_float_inverse_of_add_instance():
    """AddInstance keeps `transform_mat.inverse()` (instancer.cpp:130), Eigen's float32 4x4 inverse; the product (and prepare_instances)
    invert in double and round.  Against the cofactor expansion in float32 (oracle/c: io_inverse4_float) the two differ by a few units
    in the last place on the patch matrices DistributeInstancesOnMesh builds (orthogonal frames times a scale, a translation)."""
    from oracle import c_instancer as ci
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(200):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m = np.eye(4); m[:3, :3] = q * rng.uniform(0.02, 0.5); m[:3, 3] = rng.uniform(-3, 3, size=3)
        m32 = m.astype(F)
        a = ci.inverse4_float(m32)
        b = np.linalg.inv(m32.astype(np.float64)).astype(F)
        scale = np.abs(b).max(axis=1, keepdims=True)                             # a row's entries share their magnitude (1 / scale; the translation row a few times that)
        worst = max(worst, float((np.abs(a - b) / (scale * np.finfo(F).eps)).max()))
    assert worst < 64, worst                                                    # within a few dozen ulps of the row's largest entry: rounding, not a different matrix


# ------------------------------------------------------------------------------------------------------------------------------
# DistributeInstancesOnMesh (instancer.cpp:233-390): the restatement's known answers, and the product's vectorised version against it
# ------------------------------------------------------------------------------------------------------------------------------

def test_distribute_known_answer_on_a_tilted_plane_with_sheared_uv():
    # the plane z = x: normal (-1, 0, 1) / sqrt 2.  Texture coordinates u = x + y / 2, v = y: at constant v, u grows along (1, 0, 1), so that is
    # the tangent whatever the shear; bitangent = n x t = (0, 1, 0)
    v = F([[0, 0, 0], [1, 0, 1], [0, 1, 0], [1, 1, 1]]); f = [[0, 1, 2], [1, 3, 2]]
    n = np.tile(F([-1, 0, 1]) * F(3.0), (4, 1))                                   # (not normalised: the reference normalises, :266)
    uv = np.stack([v[:, 0] + 0.5 * v[:, 1], v[:, 1]], -1)
    tr, scale = io.distribute_instances_on_mesh(v, f, n, uv, 0.25)
    r2 = np.sqrt(0.5)
    assert scale == 0.25 and tr.shape == (4, 4, 4)
    for k in range(4):
        assert np.allclose(tr[k, :3, 0], 0.25 * np.asarray([r2, 0, r2]), atol=1e-7) and np.allclose(tr[k, :3, 1], [0, 0.25, 0], atol=1e-7)
        assert np.allclose(tr[k, :3, 2], 0.25 * np.asarray([-r2, 0, r2]), atol=1e-7) and np.array_equal(tr[k, :3, 3], v[k]) and np.array_equal(tr[k, 3], [0, 0, 0, 1])
    # anchors: the frame at the closest point; one further than an average edge length away has no triangle (the reference indexes with -1)
    tr, _ = io.distribute_instances_on_mesh(v, f, n, uv, 0.25, patch_origins=F([[0.5, 0.25, 0.6]]))
    assert np.allclose(tr[0, :3, 2], 0.25 * np.asarray([-r2, 0, r2]), atol=1e-7) and np.array_equal(tr[0, :3, 3], F([0.5, 0.25, 0.6]))
    with pytest.raises(ValueError):
        io.distribute_instances_on_mesh(v, f, n, uv, 0.25, patch_origins=F([[0.5, 0.5, 9.0]]))
    # scale <= 0: the average edge length; 2 sides sqrt 2, 2 sides 1, the diagonal sqrt 3 (twice): (2 sqrt 2 + 2 + 2 sqrt 3) / 6
    _, scale = io.distribute_instances_on_mesh(v, f, n, uv, -1.0)
    assert np.isclose(scale, (2 * np.sqrt(2) + 2 + 2 * np.sqrt(3)) / 6, rtol=1e-6)
    # a vertex listed twice gets ONE patch (:348-349); jitter turns the frame about its normal by jitter_amount * U(0, pi) of mt19937(seed)
    v2 = np.concatenate([v, v[:1]]); n2 = np.concatenate([n, n[:1]]); uv2 = np.concatenate([uv, uv[:1]])
    tr, _ = io.distribute_instances_on_mesh(v2, f, n2, uv2, 0.25, jitter_amount=0.5, seed=3)
    assert tr.shape[0] == 4
    u0 = next(io.mt19937_canonical_floats(3))
    ang = 0.5 * float(u0) * np.pi
    assert np.allclose(tr[0, :3, 1] / 0.25, np.cos(ang) * np.asarray([0, 1, 0]) + np.sin(ang) * np.cross([-r2, 0, r2], [0, 1, 0]), atol=1e-6)
    assert np.allclose(np.einsum("kij,kil->kjl", tr[:, :3, :3], tr[:, :3, :3]), 0.0625 * np.eye(3), atol=1e-6)      # still orthogonal frames


@pytest.mark.parametrize("anchors,jitter", [(False, 0.0), (False, 0.7), (True, 0.0), (True, 1.0)])
def test_the_products_distribute_matches_the_restatement(anchors, jitter):
    """nerf_tex_amd.instancer.distribute_instances_on_mesh (vectorised numpy, a k-d tree for the anchors) against the sequential
    restatement on a CURVED mesh with sheared texture coordinates and a duplicated vertex: the same patches, to float32 rounding."""
    from nerf_tex_amd.instancer import distribute_instances_on_mesh as product
    v, f, uv = sheet_mesh(n=6, extent=1.0, amp=0.25, z0=0.0)
    xs = v[:, 0]; ys = v[:, 1]
    nrm = np.stack([-0.25 * 1.7 * np.cos(1.7 * xs) * np.cos(1.3 * ys), 0.25 * 1.3 * np.sin(1.7 * xs) * np.sin(1.3 * ys), np.ones_like(xs)], -1).astype(F)
    v = np.concatenate([v, v[7:8]]); nrm = np.concatenate([nrm, nrm[7:8]]); uv = np.concatenate([uv, uv[7:8]])     # vertex 7 once more (unused by faces)
    origins = None
    if anchors:
        rng = np.random.default_rng(5)
        origins = (v[rng.choice(36, size=11, replace=False)] + rng.normal(size=(11, 3)) * [0.05, 0.05, 0.01]).astype(F)
    want, s0 = io.distribute_instances_on_mesh(v, f, nrm, uv, 0.09, origins, jitter, seed=4)
    got, s1 = product(v, f, nrm, uv, 0.09, origins, jitter, 4)
    assert got.shape == want.shape == ((11 if anchors else 36), 4, 4) and np.isclose(s0, s1)
    assert np.allclose(got, want, rtol=0, atol=2e-7 * 0.09 + 1e-7), np.abs(got - want).max()
