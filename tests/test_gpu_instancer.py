"""The patch instancer on the GPU (`ntx_instancer_model_input`, ABI v4) against the restatement of the reference's
C_Instancer::GetModelInput (oracle/instancer_oracle.py <- instancer/src/instancer.cpp:751-1037), through the mirror of the
reference's Cython class (nerf_tex_amd.instancer.Instancer <- instancer/instancer.pyx).  `-m gpu`.

The oracle is fed the instance matrices the library holds (`ntx_instancer_matrices`) and restates the product's Philox draws,
and both sides spell their float32 operations in one order: every output is compared BIT FOR BIT."""

import json
import os

import numpy as np
import pytest

from oracle import instancer_oracle as io
from oracle import nerftex_oracle as orc
from tests.common import TOL, make_model
from tests.test_oracle_instancer import UNIT, random_rays, random_scene, translate

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = np.float32
NAMES = ["rays_d_map", "pts", "t", "dists", "color", "density", "density_weight", "instance_id", "hit", "params_map"]


def gpu_instancer(spec_kw, transformations, **kw):
    from nerf_tex_amd.instancer import Instancer
    return Instancer(spec_kw["b_0"], spec_kw["b_1"], transformations=[np.asarray(m).tolist() for m in transformations], **kw)


def run_gpu(inst, o, d, params, S, h, seed, ray_index=None):
    out = inst.get_model_input(o, d, params, S, h, seed=seed, ray_index=ray_index)
    torch.cuda.synchronize()
    res = [x.cpu().numpy() for x in out]
    hit = np.zeros(o.shape[0], bool); hit[res[8][:, 0]] = True                   # idxs = where(hit)
    res[8] = hit
    return res


def oracle_textures(textures):
    """The constructor's `textures` list for the restatement: pixel arrays become named images decoded by the ORACLE's loadTexture."""
    names, images = [], {}
    for i, t in enumerate(textures):
        if isinstance(t, str):
            names.append(t)
        else:
            names.append(f"#{i}"); images[f"#{i}"] = io.texture_from_pixels(t)
    return names, images


def run_oracle(inst, box, o, d, params, S, h, seed, method="random", textures=(), mean=False, mesh=None, ray_index=None, patch_scale=1.0,
               tex_mesh=None, aux_pixels=None, **kw):
    """`tex_mesh` = (vertices, faces, uv): DistributeInstancesOnMesh's state (parameter textures apply); `aux_pixels`: the pixel arrays
    of the auxiliary meshes' textures, in the order of the instancer's texture sets; kw: shadow / texture sample counts."""
    names, images = oracle_textures(textures)
    if mesh is not None and len(mesh) > 2 and inst.mesh_prim is not None:
        kw = dict(kw, mesh_prim=inst.mesh_prim)
    if aux_pixels is not None:
        kw = dict(kw, mesh_uv=inst.mesh_uv, mesh_tex=inst.mesh_tex, aux_textures=[io.texture_from_pixels(px) for px in aux_pixels])
    spec = io.make_spec(box["b_0"], box["b_1"], None, textures=names, images=images, instance_sampling_method=method, use_mean_distance=mean,
                        mesh=mesh, matrices=inst.matrices(), instancer_mesh=tex_mesh, patch_scale=patch_scale, **kw)
    spec.patch_scale = patch_scale
    n = o.shape[0]
    return list(io.get_model_input(spec, o, d, params, S, h, io.offset_uniforms(n, seed, ray_index), io.choice_uniforms(n, S, seed, ray_index)))


def wavy_sheet(n=9, extent=1.7, amp=0.07, z0=-0.12):
    """A waving sheet under the patches of `random_scene` as (vertices, faces, uv): the instancer mesh parameter textures are looked up
    on; the texture coordinates are sheared so that neither axis of the image follows an axis of the world."""
    xs = np.linspace(-extent, extent, n)
    x, y = np.meshgrid(xs, xs, indexing="ij")
    z = z0 + amp * np.sin(1.7 * x) * np.cos(1.3 * y)
    v = np.stack([x, y, z], -1).reshape(-1, 3).astype(F)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, e = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, e], -1)]).astype(np.int32)
    u = (x + extent) / (2 * extent); w = (y + extent) / (2 * extent)
    uv = np.stack([0.8 * u + 0.2 * w, 0.1 * u + 0.9 * w], -1).reshape(-1, 2).astype(F)
    return v, f, uv


def random_pixels(seed, h=12, w=16, c=1):
    """A smooth random image: neighbouring texels differ, so a wrong axis, flip or texel shows."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    img = np.stack([0.5 + 0.25 * np.sin(6 * xx * rng.uniform(.5, 1.5) + k) + 0.25 * np.cos(5 * yy * rng.uniform(.5, 1.5) - k) for k in range(c)], -1)
    return np.clip(img * 255 + rng.integers(-6, 7, size=img.shape), 0, 255).astype(np.uint8)


def assert_same(got, want):
    for name, g, w in zip(NAMES, got, want):
        assert g.shape == w.shape and g.dtype == w.dtype, (name, g.shape, w.shape, g.dtype, w.dtype)
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)
            raise AssertionError(f"{name}: {len(bad)} of {g.size} elements differ, first at {bad[0].tolist()}: {g[tuple(bad[0])]!r} != {w[tuple(bad[0])]!r}")


def test_matrices_are_add_instances():
    rng = np.random.default_rng(0)
    tr = []
    for _ in range(9):
        m = np.eye(4); m[:3, :3] = rng.normal(size=(3, 3)); m[:3, 3] = rng.normal(size=3)
        tr.append(m.astype(F))
    inst = gpu_instancer(UNIT, tr)
    assert inst.n_instances() == 9                                              # GetNumberOfInstances
    w2p, dirs, org = inst.matrices()
    inv, dir_t, origins = io.prepare_instances(tr)
    assert np.allclose(w2p, inv, rtol=2e-6, atol=1e-7) and np.array_equal(w2p[:, 3], np.tile(F([0, 0, 0, 1]), (9, 1)))
    assert np.allclose(dirs, dir_t, rtol=2e-7, atol=1e-7) and np.array_equal(org, origins)


def test_known_answers_on_the_kernel():
    """The hand-computed cases of tests/test_oracle_instancer.py, now from the kernel."""
    inst = gpu_instancer(UNIT, [translate(), translate(z=5)])
    o = F([[0, 0, -5], [5, 5, -5]]); d = F([[0, 0, 1], [0, 0, 1]])
    import nerf_tex_amd.instancer as mod
    rd, pts, t, dists, color, dens, w, iid, hit, pm = run_gpu(inst, o, d, np.zeros((2, 0), F), 16, 0.5, seed=3)
    u = io.offset_uniforms(2, 3)[0] * F(0.5)
    assert np.array_equal(dists[0, :8], np.full(8, .5, F)) and not dists[0, 8:].any() and not dists[1].any()
    want_t = np.asarray([(F(s) * F(.5) + u) + F(4) for s in range(4)] + [(F(s) * F(.5) + u) + (F(9) - F(2)) for s in range(4, 8)], F)
    assert np.array_equal(t[0, :8], want_t) and not t[0, 8:].any()
    assert iid[0, :8].tolist() == [0] * 4 + [1] * 4 and hit.tolist() == [True, False]
    assert np.allclose(pts[0, :8, 2], np.r_[want_t[:4] - 5, want_t[4:] - 10], atol=1e-6)
    assert (w == 1).all() and not dens.any() and not color.any() and np.array_equal(rd[1], np.tile(d[1], (16, 1)))
    assert mod.Instancer.device_native and inst.status() == 0


@pytest.mark.parametrize("method", ["random", "nearest", "nearest_blend"])
@pytest.mark.parametrize("mesh", [False, True])
@pytest.mark.parametrize("textures,mean", [((), False), (("", "light", ""), False), (("point", ""), True)])
def test_model_input_bit_for_bit(method, mesh, textures, mean):
    spec0 = random_scene(11, k=24, method=method, textures=textures, mesh=mesh)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    rng = np.random.default_rng(11)
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]      # patch -> world of the scene
    msh = (spec0.mesh_v, spec0.mesh_f) if mesh else None
    inst = gpu_instancer(box, tr, textures=list(textures), instance_sampling_method=method, use_mean_distance=mean, mesh=msh)
    n, S, h = 150, 160, 0.02
    o, d = random_rays(11, n)
    P = spec0.n_parameters
    params = rng.uniform(0.2, 3.0, size=(n, P)).astype(F)
    got = run_gpu(inst, o, d, params, S, h, seed=0x1234567812345678)
    want = run_oracle(inst, box, o, d, params, S, h, 0x1234567812345678, method, textures, mean, msh)
    assert want[8].sum() > 50 and (want[3] > 0).sum() > 2000 and len(np.unique(want[7])) > 10
    if method != "nearest":
        assert (want[6] > 1).any()                                              # overlaps were met
    if mesh:
        assert want[5].any()
    assert_same(got, want)
    assert inst.status() == 0


@pytest.mark.parametrize("textures", [("", "light", ""), ("point", "")])
@pytest.mark.parametrize("mesh", [False, True])
@pytest.mark.parametrize("n_shadow_samples,min_shadow", [(48, 4), (96, 8), (100000, 4)])
def test_shadow_rays_bit_for_bit(textures, mesh, n_shadow_samples, min_shadow):
    """cast_shadow_rays (instancer.cpp:591-602, 861, 945-961, 1018-1027): shadow samples spaced along every segment and the nearer of
    the two around a step (N * total < n_pts), or a query per step (the last case); a shadowed sample gets the light direction
    (0, 0, -1).  `point`: the reference hands isShadowed the light's POSITION as the direction (:956, 961) -- kept."""
    spec0 = random_scene(31, k=24, method="nearest", textures=textures, mesh=mesh)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    msh = (spec0.mesh_v, spec0.mesh_f) if mesh else None
    sh = dict(cast_shadow_rays=True, min_shadow_samples=min_shadow, n_shadow_samples=n_shadow_samples)
    inst = gpu_instancer(box, tr, textures=list(textures), instance_sampling_method="nearest", mesh=msh, **sh)
    n, S, h = 120, 128, 0.02
    o, d = random_rays(31, n)
    rng = np.random.default_rng(31)
    P = spec0.n_parameters
    params = rng.uniform(0.2, 1.0, size=(n, P)).astype(F)
    ld = 1 if textures[0] == "" else 1                          # ('', 'light', ''): light at 1..3; ('point', ''): position at 1..3
    light = rng.normal(size=(n, 3)); light[:, 2] = np.abs(light[:, 2]) * 0.7 + 0.1
    params[:, ld:ld + 3] = light
    got = run_gpu(inst, o, d, params, S, h, seed=5)
    want = run_oracle(inst, box, o, d, params, S, h, 5, "nearest", textures, False, msh, **sh)
    emitted = want[2] > 0
    dark = emitted & np.all(want[9][..., ld:ld + 3] == F([0, 0, -1]), axis=-1)
    assert dark.sum() > 100 and (emitted & ~dark).sum() > 100, (dark.sum(), emitted.sum())
    assert_same(got, want)
    assert inst.status() == 0


def test_rays_can_be_split_and_sharded():
    """The draws are keyed by the global ray index: two calls with an index map, and a call longer than the reserved
    workspace (cut into pieces inside the library), give the bits of one call."""
    spec0 = random_scene(5, k=10, method="random")
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    inst = gpu_instancer(box, tr)
    n, S, h = 96, 64, 0.03
    o, d = random_rays(5, n)
    par = np.zeros((n, 0), F)
    whole = run_gpu(inst, o, d, par, S, h, seed=9)
    a = run_gpu(inst, o[:32], d[:32], par[:32], S, h, seed=9, ray_index=(0, 32, 32))
    b = run_gpu(inst, o[32:], d[32:], par[32:], S, h, seed=9, ray_index=(32, 64, 64))
    assert_same([np.concatenate([x, y]) for x, y in zip(a, b)], whole)
    assert not np.array_equal(run_gpu(inst, o, d, par, S, h, seed=10)[2], whole[2])
    # 70 000 rays > NTX_INSTANCER_DEFAULT_MAX_RAYS = 65 536: two pieces
    big = 70_000
    ob, db = random_rays(6, big)
    outb = run_gpu(inst, ob, db, np.zeros((big, 0), F), 8, 0.2, seed=4)
    tail = run_gpu(inst, ob[65_000:], db[65_000:], np.zeros((5_000, 0), F), 8, 0.2, seed=4, ray_index=(65_000, 5_000, 5_000))
    assert_same([x[65_000:] for x in outb], tail)
    assert outb[8][65_536:].any()
    # a chunk of a larger call as its renderer passes it, (k0, n, n), is a plain offset and splits anywhere (render_chunk > 65 536)
    chunk = run_gpu(inst, ob, db, np.zeros((big, 0), F), 8, 0.2, seed=4, ray_index=(0, big, big))
    assert_same(chunk, outb)
    # a split call's pieces must start on run boundaries of the index map: refused before anything is launched
    from nerf_tex_amd import _lib
    with pytest.raises(_lib.NtxError) as e:
        inst.get_model_input(ob, db, np.zeros((big, 0), F), 8, 0.2, seed=4, ray_index=(0, 1000, 2000))
    assert e.value.code == _lib.NTX_E_INVALID and "ray_run_length" in str(e.value)


def test_overflow_flags_and_refusals(tmp_path):
    from nerf_tex_amd import _lib
    from nerf_tex_amd.instancer import Instancer
    nested = [translate(s=1 + 0.01 * k) for k in range(120)]                     # 240 face crossings on a ray through the middle
    inst = gpu_instancer(UNIT, nested)
    run_gpu(inst, F([[0, 0, -5]]), F([[0, 0, 1]]), np.zeros((1, 0), F), 8, 0.5, seed=1)
    assert inst.status() & 1
    inst = gpu_instancer(UNIT, nested[:90])                                      # 180 crossings: a point in 90 patches at once
    out = run_gpu(inst, F([[0, 0, -5]]), F([[0, 0, 1]]), np.zeros((1, 0), F), 8, 0.5, seed=1)
    want = run_oracle(inst, UNIT, F([[0, 0, -5]]), F([[0, 0, 1]]), np.zeros((1, 0), F), 8, 0.5, 1)
    assert inst.status() == 0
    assert_same(out, want)
    assert out[6].max() == 90                                                    # density_weight = patches the point lies in
    with pytest.raises(OSError):                                                 # an image file that is not there
        Instancer(UNIT["b_0"], UNIT["b_1"], transformations=[translate().tolist()], textures=["meshes/smooth_checkerboard.png"])
    for kw in (dict(textures=["a.jpg"]), dict(textures=[random_pixels(k) for k in range(5)])):     # not a PNG; five texture files
        with pytest.raises(_lib.NtxError) as e:
            Instancer(UNIT["b_0"], UNIT["b_1"], transformations=[translate().tolist()], **kw)
        assert e.value.code == _lib.NTX_E_UNSUPPORTED
    with pytest.raises(_lib.NtxError):                                           # a singular patch matrix
        Instancer(UNIT["b_0"], UNIT["b_1"], transformations=[np.zeros((4, 4)).tolist()])
    # the list the reference's transformation_export_path writes (instancer.cpp:1040-1061) comes back as the same instancer
    tr = [translate(x=.3, s=.5), translate(z=1)]
    a = Instancer(UNIT["b_0"], UNIT["b_1"], transformations=[m.tolist() for m in tr], transformation_export_path=str(tmp_path / "t.json"))
    assert np.allclose(np.asarray(json.load(open(tmp_path / "t.json")), F), np.stack(tr))
    b = Instancer(UNIT["b_0"], UNIT["b_1"], transformations_path=str(tmp_path / "t.json"), patch_scale=0.5)
    assert all(np.array_equal(x, y) for x, y in zip(a.matrices(), b.matrices())) and b.patch_scale == 0.5 and a.patch_scale == 1.0


@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
@pytest.mark.parametrize("npar,textures,blur", [((1, 6), ["", "", "", "", "light"], None), ((2, 3), ["", "", "light"], 0)])
def test_instance_renderer_end_to_end(npar, textures, blur, precision):
    """Rays -> GPU instancer -> ntx_render_instanced without leaving HBM, against oracle instancer -> float64 oracle of the
    InstanceRenderer tail (renderer.py:247-354), through the reference-shaped classes."""
    from nerf_tex_amd.renderer import InstanceRenderer
    model, mspec, wts = make_model(npar, dense_media=True)
    P = sum(npar)
    spec0 = random_scene(21, k=30, method="nearest", mesh=True)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    msh = (spec0.mesh_v, spec0.mesh_f)
    inst = gpu_instancer(box, tr, textures=textures, instance_sampling_method="nearest", mesh=msh)
    patch_scale, step, S = 0.35, 0.01, 256
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=patch_scale, step_size=step, blur_idx=blur,
                         render_chunk=64, density_scale=40.0, precision=precision)
    n = 150
    o, d = random_rays(21, n)
    tt = np.tile(F([[1.0, 2.0]]), (n, 1)); tt[7] = np.inf
    rng = np.random.default_rng(2)
    params = rng.uniform(0.2, 1, size=(1, P)).astype(F)
    cone = rng.uniform(1e-3, 5e-3, size=(n, 1)).astype(F)
    dv = torch.device("cuda", 0)
    dd = lambda a: torch.as_tensor(a, device=dv)[None]
    out = r(dd(o), dd(d), dd(tt), parameters=torch.as_tensor(params, device=dv), cone_scale=dd(cone), instancer_seed=77)
    r.raise_if_nonfinite()
    got = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
    keep = np.isfinite(tt[:, 0])
    ko = np.nonzero(keep)[0]
    pr = np.repeat(params, len(ko), 0)
    want = np.zeros((n, 4))
    for c0 in range(0, len(ko), 64):                                            # the renderer's chunks key the draws
        sl = ko[c0:c0 + 64]
        b = run_oracle(inst, box, o[sl], d[sl], pr[:len(sl)], S, step, 77, "nearest", textures, False, msh, ray_index=(c0, len(sl), len(sl)))
        rc, ra = orc.instance_evaluate_model(wts, mspec, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], cone[sl], blur,
                                             patch_scale, 40.0, True, False, False, (1., 1., 1.), None, dtype=np.float64)
        want[sl, :3] = rc; want[sl, 3] = ra
    assert orc.rel_linf(got, want) <= TOL
    assert want[:, 3].max() > 0.5 and (want[:, 3] > 0).sum() > 40 and np.all(got[7] == 0)


@pytest.mark.parametrize("precision", ["float32", "fp16x3"])
@pytest.mark.parametrize("shadows", [False, True])
def test_sparse_hand_off_is_the_dense_one(precision, shadows):
    """NTX_OPT_INSTANCER_SPARSE (what InstanceRenderer asks of this package's instancer): the rows behind a ray's last marching step stay
    unwritten in six of the ten buffers -- here they hold NaN from before the call -- and the image is the image of the dense buffers bit
    for bit, with finite numerics: the tail reads a row only where dists > 0 (renderer.py:284-288).  The buffers themselves: equal wherever
    dists > 0, dists equal everywhere, untouched NaN behind."""
    from nerf_tex_amd.renderer import InstanceRenderer
    model, mspec, wts = make_model((1, 6), dense_media=True)
    spec0 = random_scene(33, k=40, method="nearest", mesh=True)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    textures = ["", "", "", "", "light"]
    inst = gpu_instancer(box, tr, textures=textures, instance_sampling_method="nearest", mesh=(spec0.mesh_v, spec0.mesh_f), cast_shadow_rays=shadows)
    n, S, step = 300, 256, 0.01
    o, d = random_rays(5, n)
    params = np.tile(F([[1, 1, 1, .1, 0.3, 0.2, 1]]), (n, 1))
    dense = inst.get_model_input(o, d, params, S, step, seed=9)
    sparse = inst.get_model_input(o, d, params, S, step, seed=9, sparse=True, fill=float("nan"))
    live = (dense[3] > 0)
    assert torch.equal(dense[3], sparse[3]) and 0 < int(live.sum()) < live.numel() // 2
    for k in (0, 1, 2, 6, 7, 9):                                                  # rays_d_map, pts, t, density_weight, instance_id, params_map
        a, b = dense[k], sparse[k]
        m = live if a.dim() == 2 else live[..., None].expand_as(a)
        assert torch.equal(a[m], b[m])
        if a.dtype == torch.float32:
            assert torch.isnan(b[~m]).all()
    for k in (4, 5, 8):
        assert torch.equal(dense[k], sparse[k])
    r = InstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=0.35, step_size=step, render_chunk=128, density_scale=40.0, precision=precision)
    dv = torch.device("cuda", 0)
    dd = lambda a: torch.as_tensor(a, device=dv)[None]
    tt = np.tile(F([[1.0, 2.0]]), (n, 1))
    call = lambda **kw: r(dd(o), dd(d), dd(tt), parameters=torch.as_tensor(params[:1], device=dv), cone_scale=dd(np.full((n, 1), 2e-3, F)), instancer_seed=5, **kw)
    a = call(instancer_sparse=False)
    b = call(instancer_fill=float("nan"))                                         # sparse is the renderer's default with this instancer
    r.raise_if_nonfinite()
    assert torch.equal(a["color_pred"], b["color_pred"]) and torch.equal(a["alpha_pred"], b["alpha_pred"]) and float(a["alpha_pred"].max()) > 0.5


@pytest.mark.parametrize("seed", range(int(os.environ.get("NTX_INSTANCER_FUZZ_SEEDS", "16"))))     # a soak run sets more (profiles/r03/soak_instancer.txt)
def test_fuzz_scenes_bit_for_bit(seed):
    """Random scenes, ray sets and settings (patch count, box, scales, step size, buffer length incl. too short ones, choice rule,
    lights, mean distances, mesh, shadow rays in both modes, rays that start inside the patch layer, image textures as parameters --
    interpolated or per step, one or more channels -- and on auxiliary meshes): every buffer bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(3, 48))
    method = ["random", "nearest", "nearest_blend"][seed % 3]
    textures = [(), ("", "light"), ("point",), ("", "", "light", "")][int(rng.integers(0, 4))]
    mesh = bool(rng.integers(0, 2))
    mean = bool(rng.integers(0, 2))
    shadows = bool(textures) and bool(rng.integers(0, 2))
    spec0 = random_scene(200 + seed, k=k, method=method, textures=textures, mesh=mesh)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    msh = (spec0.mesh_v, spec0.mesh_f) if mesh else None
    sh = dict(cast_shadow_rays=True, min_shadow_samples=int(rng.integers(2, 9)), n_shadow_samples=int(rng.choice([16, 64, 100000]))) if shadows else {}
    aux = bool(textures) and bool(rng.integers(0, 2))                            # an auxiliary mesh (shaded closing sample) needs a light
    kw, okw = {}, {}
    # parameter textures (drawn after everything above, so that the scenes of earlier rounds' soaks stay what they were): one to three
    # image entries in front of / between the others, looked up on a waving sheet that then is the instancer mesh
    tex_mode = int(rng.integers(0, 3))                                           # 0: none, 1: interpolated, 2: per step
    patch_scale = 1.0
    textures = list(textures)
    if tex_mode:
        for _ in range(int(rng.integers(1, 4))):
            textures.insert(int(rng.integers(0, len(textures) + 1)), random_pixels(int(rng.integers(0, 1000)), h=int(rng.integers(2, 12)), w=int(rng.integers(2, 12)),
                                                                                 c=int(rng.choice([1, 1, 2, 3, 4]))))
        tex_mesh = wavy_sheet(n=int(rng.integers(3, 10)), amp=float(rng.uniform(0, 0.15)), z0=float(rng.uniform(-0.3, 0.1)))
        patch_scale = float(rng.choice([0.2, 0.45, 1.0]))                        # the lookup radius: some points find no triangle within reach
        tkw = dict(min_texture_samples=int(rng.integers(2, 9)), n_texture_samples=int(rng.choice([8, 40])) if tex_mode == 1 else 100000)
        kw.update(instancer_mesh=tex_mesh, patch_scale=patch_scale, **tkw)
        okw.update(tex_mesh=tex_mesh, **tkw)
        msh = (tex_mesh[0], tex_mesh[1])                                         # DistributeInstancesOnMesh's mesh is the culling mesh (:379)
        kw["mesh"] = None
    if aux:
        z0, z1 = rng.uniform(0.3, 1.2, size=2)
        nv = rng.normal(size=(4, 3)) * 0.2 + [0, 0, 1]
        quad = (F([[-1.5, -1.5, z0], [1.5, -1.5, z1], [1.5, 1.5, z1], [-1.5, 1.5, z0]]), [[0, 1, 2], [0, 2, 3]], (nv / np.linalg.norm(nv, axis=-1, keepdims=True)).astype(F))
        if tex_mode and rng.integers(0, 2):                                      # ... with its own texture
            px = random_pixels(int(rng.integers(0, 1000)), h=5, w=6, c=int(rng.choice([1, 3, 4])))
            kw["auxiliary_meshes"] = [(quad + (rng.uniform(0, 1, size=(4, 2)).astype(F),), px)]
            okw["aux_pixels"] = [px]
        else:
            kw["auxiliary_meshes"] = [(quad, "")]
    if "mesh" not in kw:
        kw["mesh"] = msh
    inst = gpu_instancer(box, tr, textures=textures, instance_sampling_method=method, use_mean_distance=mean, **sh, **kw)
    if aux:
        msh = inst.meshes                                                        # (vertices, faces, normals, kind) as the library holds them
    n = 48
    o, d = random_rays(200 + seed, n)
    if not shadows and not tex_mode:     # (a segment the ray never leaves has no length in the reference: instancer.cpp:990, 1001 read past its list)
        o[:6] = o[:6] * F(0.15)                                                  # six rays start inside the layer of patches
    d[6:9] = d[6:9] * F(rng.uniform(0.5, 2.0))                                   # rays_d is used as given (un-normalised rays march in their own units)
    S = int(rng.choice([7, 33, 64, 100, 257]))
    h = float(rng.choice([0.004, 0.01, 0.05, 0.7]))
    P = inst.n_parameters
    params = rng.uniform(0.1, 2.0, size=(n, P)).astype(F)
    seed64 = int(rng.integers(0, 2 ** 62))
    got = run_gpu(inst, o, d, params, S, h, seed=seed64)
    want = run_oracle(inst, box, o, d, params, S, h, seed64, method, textures, mean, msh, patch_scale=patch_scale, **sh, **okw)
    assert_same(got, want)
    assert inst.status() == 0


def test_the_measured_scene_bit_for_bit():
    """Parity where the numbers of DESIGN 4.5 are measured: the scene of tools/bench_instancer.py and bench.py's carpet_instanced_scene (synthetic.patch_sheet: 48 x 48 patches of the carpet
    config's box on a waving sheet that is also the mesh, 1024 steps of 0.002, 'nearest', a directional light) -- 40 seeded rays of a
    4096-ray call against the restatement, every buffer bit for bit (~260 steps per ray through ~35 overlapping patches)."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.dataset import look_at
    from nerf_tex_amd.instancer import Instancer
    bi = type("scene", (), {"B0": synthetic.PATCH_BOX[0], "B1": synthetic.PATCH_BOX[1]})
    tr, v, f = synthetic.patch_sheet(48)
    textures = ['', '', '', '', 'light']
    inst = Instancer(bi.B0, bi.B1, textures=textures, transformations=tr, instance_sampling_method="nearest", mesh=(v, f))
    fam = synthetic.FAMILIES["carpet"]
    c2w = look_at(np.asarray(fam["cam"], F))
    focal = 800 / np.tan(fam["angle"] / 2) / 2
    side, n, S, h = 64, 4096, 1024, 0.002
    r0 = (800 - side) // 2
    rows, cols = np.meshgrid(np.arange(r0, r0 + side), np.arange(r0, r0 + side), indexing="ij")
    ro, rd, t, cone = orc.proxy_rays(np.stack([rows.ravel(), cols.ravel()], -1), 800, 800, focal, c2w, [-1.7, -1.7, -.3], [1.7, 1.7, .4], F)
    params = np.tile(np.asarray([fam["params"]], F), (n, 1))
    got = run_gpu(inst, ro, rd, params, S, h, seed=1)
    pick = np.random.default_rng(0).choice(n, size=40, replace=False)
    box = dict(b_0=bi.B0, b_1=bi.B1)
    spec = io.make_spec(box["b_0"], box["b_1"], None, textures=textures, instance_sampling_method="nearest", mesh=(v, f), matrices=inst.matrices())
    want = io.get_model_input(spec, ro[pick], rd[pick], params[pick], S, h, io.offset_uniforms(n, 1)[pick], io.choice_uniforms(n, S, 1)[pick])
    assert (want[3] > 0).sum() > 40 * 150 and want[8].all() and want[5].all()          # long marches, every ray ends on the sheet
    assert_same([g[pick] for g in got], list(want))


def test_instanced_image_through_the_render_harness():
    """What the shipped render configs run, through the reference's plugin path: a config dict in the reference's format
    (network.render.Render -> Dataset -> ParamNerf -> network.renderer.InstanceRenderer with an instancer_config), remapped and
    instantiated like main.py does; the instancer is this package's.  A 40 x 40 image of 36 patches on a sheet against the whole
    pipeline restated: oracle rays -> oracle instancer -> float64 tail."""
    from nerf_tex_amd import synthetic, util
    from nerf_tex_amd.dataset import look_at
    bi = type("scene", (), {"B0": synthetic.PATCH_BOX[0], "B1": synthetic.PATCH_BOX[1]})
    tr, v, f = synthetic.patch_sheet(6, extent=0.35, scale=0.09)
    H = W = 40
    S, step, patch_scale, dscale = 192, 0.004, 0.09, 60.0
    textures = ['', '', '', '', 'light']
    params = [1, 1, 1, .1, 0.3, 0.2, 1]
    cam = 6. * np.asarray([0.9165, 0., 0.4])
    c2w = look_at(cam)
    emb = lambda n_: {'module': 'network.model.FourierFeatures', 'n_freq_bands': n_}
    aabb0, aabb1 = [-0.7, -0.7, -.2], [0.7, 0.7, .3]
    config = {
        'module': 'network.render.Render', 'target_path': None,
        'test_dataset_config': {
            'module': 'network.dataset.Dataset',
            'data_loader_config': {'module': 'nerf_tex_amd.dataset.FromViews', 'height': H, 'width': W, 'angle': 0.16,
                                   'views': [{'pose': c2w, 'parameters': params}]},
            'pixel_sampler_config': {'module': 'network.pixel_sampler.Full'},
            'ray_sampler_config': {'module': 'network.ray_sampler.Proxy'},
            'proxy_config': {'module': 'network.proxy.AABB', 'b_0': aabb0, 'b_1': aabb1},
            'n_epochs': 1},
        'model_config': {'module': 'network.model.ParamNerf', 'pos_embedding': emb(10), 'dir_embedding': emb(4), 'param_embedding': emb(4),
                         'n_parameters': [1, 6]},
        'renderer_config': {
            'module': 'network.renderer.InstanceRenderer', 'n_samples': S, 'render_chunk': 16384, 'density_scale': dscale,
            'instancer_config': {'module': 'nerf_tex_amd.instancer.Instancer', 'b_0': bi.B0, 'b_1': bi.B1, 'cast_shadow_rays': False,
                                 'textures': textures, 'transformations': [m.tolist() for m in tr], 'mesh': (v, f), 'patch_scale': patch_scale,
                                 'instance_sampling_method': 'nearest'},
            'density_reweighting': True, 'step_size': step},
        'logger_config': {'module': 'network.logger.Logger'},
    }
    model, mspec, wts = make_model((1, 6), dense_media=True)
    blob = synthetic.synthetic_weights(model.layer_table(), seed=0, dense_media=True)
    imgs = util.instantiate(dict(util.remap_reference_config(config), weights=blob, weights_order="keras_get_weights"))
    rgba = imgs[0][0].cpu().numpy().reshape(H * W, 4)
    # the same, restated
    focal = orc.focal_from_angle(W, 0.16)
    ro, rd, t, cone = orc.proxy_rays(orc.full_pixels(H, W), H, W, focal, c2w.astype(F), aabb0, aabb1, F)
    keep = np.isfinite(t[:, 0])
    k = int(keep.sum())
    inv, dir_t, org = io.prepare_instances(tr)
    from nerf_tex_amd.instancer import Instancer
    lib_side = Instancer(bi.B0, bi.B1, textures=textures, transformations=tr, mesh=(v, f))
    spec = io.make_spec(bi.B0, bi.B1, None, textures=textures, instance_sampling_method="nearest", mesh=(v, f), matrices=lib_side.matrices())
    par = np.tile(np.asarray([params], F), (k, 1))
    b = io.get_model_input(spec, ro[keep], rd[keep], par, S, step, io.offset_uniforms(k, 0, (0, k, k)), io.choice_uniforms(k, S, 0, (0, k, k)))
    rc, ra = orc.instance_evaluate_model(wts, mspec, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], cone[keep], None,
                                         patch_scale, dscale, True, False, False, (1., 1., 1.), None, dtype=np.float64)
    want = np.zeros((H * W, 4)); want[keep, :3] = rc; want[keep, 3] = ra
    assert orc.rel_linf(rgba, want) <= TOL
    # ... of an image with something in it: rays beside the sheet, rays through tens of patches, opaque and half-transparent pixels
    assert k > 0.8 * H * W and 0.3 * k < b[8].sum() < 0.9 * k and (b[3] > 0).sum() > 20 * b[8].sum()
    assert want[:, 3].max() > 0.9 and 0.2 < want[:, 3].mean()


def test_mip_instance_renderer_end_to_end():
    """MipInstanceRenderer (renderer.py:475-587) on the GPU instancer with use_mean_distance (instancer.cpp:746-748: the samples sit at
    the mean distance of their cone segment, what the integrated positional encoding expects): rays -> instancer -> fused IPE tail
    against oracle instancer -> float64 restatement of the mip tail."""
    from nerf_tex_amd.renderer import MipInstanceRenderer
    model, mspec, wts = make_model((1, 3), "IPE", dense_media=True)
    textures = ["", "", "light"]                                                 # blur parameter, one geometry parameter, light: 5 per row
    spec0 = random_scene(41, k=30, method="nearest", textures=textures, mesh=True)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    msh = (spec0.mesh_v, spec0.mesh_f)
    inst = gpu_instancer(box, tr, textures=textures, instance_sampling_method="nearest", mesh=msh, use_mean_distance=True)
    patch_scale, step, S = 0.35, 0.01, 256
    r = MipInstanceRenderer(model=model, n_samples=S, instancer=inst, patch_scale=patch_scale, step_size=step, blur_idx=0,
                            render_chunk=10_000, density_scale=40.0)
    n = 120
    o, d = random_rays(41, n)
    tt = np.tile(F([[1.0, 2.0]]), (n, 1))
    rng = np.random.default_rng(3)
    params = rng.uniform(0.2, 1, size=(1, 5)).astype(F); params[0, 0] = 6.0
    cone = rng.uniform(1e-3, 5e-3, size=(n, 1)).astype(F)
    dv = torch.device("cuda", 0)
    dd = lambda a: torch.as_tensor(a, device=dv)[None]
    out = r(dd(o), dd(d), dd(tt), parameters=torch.as_tensor(params, device=dv), cone_scale=dd(cone), instancer_seed=5)
    r.raise_if_nonfinite()
    got = np.concatenate([out["color_pred"][0].cpu().numpy(), out["alpha_pred"][0].cpu().numpy()[:, None]], -1)
    b = run_oracle(inst, box, o, d, np.repeat(params, n, 0), S, step, 5, "nearest", textures, True, msh, ray_index=(0, n, n))
    rc, ra = orc.mip_instance_evaluate_model(wts, mspec, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[8], b[9], cone, 0, patch_scale, 40.0,
                                             True, False, False, (1., 1., 1.), dtype=np.float64)
    assert orc.rel_linf(got, np.concatenate([rc, ra[:, None]], -1)) <= TOL
    assert ra.max() > 0.5 and (b[3] > 0).sum() > 2000


def test_degenerate_scenes():
    """No patches at all, a mesh and no patches, one marching step per ray, no parameters: the defaults of instancer.pyx:41-50 and the
    closing sample, bit for bit."""
    quad = ([[-2, -2, .5], [2, -2, .5], [2, 2, .5], [-2, 2, .5]], [[0, 1, 2], [0, 2, 3]])
    o = F([[0.3, 0.2, -5], [5, 5, -5], [0.1, 0.1, 3]]); d = F([[0, 0, 1], [0, 0, 1], [0, 0, 1]])
    for mesh in (None, quad):
        inst = gpu_instancer(UNIT, [], mesh=mesh)
        assert inst.n_instances() == 0
        for S in (1, 5):
            got = run_gpu(inst, o, d, np.zeros((3, 0), F), S, 0.5, seed=2)
            want = run_oracle(inst, UNIT, o, d, np.zeros((3, 0), F), S, 0.5, 2, mesh=mesh)
            assert_same(got, want)
            assert got[8].tolist() == ([True, False, False] if mesh else [False] * 3) and not got[3].any()
    inst = gpu_instancer(UNIT, [translate()], textures=["", "light"])
    par = F([[0.5, 0, 0, 1]] * 3)
    got = run_gpu(inst, o, d, par, 1, 0.5, seed=2)                                # four steps needed, one fits
    want = run_oracle(inst, UNIT, o, d, par, 1, 0.5, 2, textures=("", "light"))
    assert_same(got, want)
    assert got[3][0, 0] == (F(0.5) + F(2.0)) - F(1) * F(0.5) and got[8].tolist() == [True, False, False]


def test_failed_calls_write_nothing():
    """Every argument of ntx_instancer_model_input is checked before the first launch: a call that returns an error leaves the
    caller's buffers as they were (the convention of the other entry points)."""
    import ctypes as C
    from nerf_tex_amd import _lib
    inst = gpu_instancer(UNIT, [translate()])
    dv = torch.device("cuda", 0)
    n, S = 8, 16
    ro = torch.zeros((n, 3), device=dv); ro[:, 2] = -5
    rd = torch.zeros((n, 3), device=dv); rd[:, 2] = 1
    bufs = {k: torch.full(shape, 7, device=dv, dtype=dt) for k, shape, dt in [
        ("rays_d_map", (n, S, 3), torch.float32), ("pts", (n, S, 3), torch.float32), ("t", (n, S), torch.float32), ("dists", (n, S), torch.float32),
        ("color", (n, 3), torch.float32), ("alpha", (n,), torch.float32), ("weight", (n, S), torch.float32), ("iid", (n, S), torch.int32),
        ("hit", (n,), torch.uint8)]}
    p = lambda k: bufs[k].data_ptr()

    def call(n_pts=S, step=0.5, opts=None, pts=None, n_rays=n):
        return _lib.lib.ntx_instancer_model_input(inst._h, ro.data_ptr(), rd.data_ptr(), None, n_rays, n_pts, step, 1, opts, p("rays_d_map"),
                                                  pts if pts is not None else p("pts"), p("t"), p("dists"), p("color"), p("alpha"), p("weight"),
                                                  p("iid"), p("hit"), None, None, None)
    bad_opts = _lib.render_opts(ray_index=(0, 4, 2))                              # stride < run length
    for rc in (call(n_pts=0), call(n_pts=5000), call(step=0.0), call(step=float("inf")), call(opts=bad_opts), call(n_rays=-1)):
        assert rc == _lib.NTX_E_INVALID
    assert _lib.lib.ntx_instancer_model_input(None, *([None] * 3), n, S, 0.5, 1, None, *([None] * 12)) == _lib.NTX_E_INVALID
    assert _lib.lib.ntx_instancer_model_input(inst._h, ro.data_ptr(), rd.data_ptr(), None, n, S, 0.5, 1, None, p("rays_d_map"), None, *([None] * 10)) == _lib.NTX_E_INVALID
    torch.cuda.synchronize()
    assert all(bool((b == 7).all()) for b in bufs.values())
    assert call() == _lib.NTX_OK                                                  # ... and the same call with good arguments fills them
    torch.cuda.synchronize()
    assert not bool((bufs["dists"] == 7).any()) and bool(bufs["hit"].all())


@pytest.mark.parametrize("textures,shadows", [(("", "light", ""), False), (("point", ""), False), (("", "light"), True)])
def test_auxiliary_meshes_bit_for_bit(textures, shadows):
    """auxiliary_meshes (AddMesh + shadeMesh, instancer.cpp:393-417, 716-743): a ground sheet (the instancer mesh, black) and two
    shaded auxiliary meshes -- a tilted roof over half of the scene that also shadows the ground, and a wall -- every buffer incl.
    the closing sample's colour bit for bit."""
    spec0 = random_scene(51, k=20, method="nearest", textures=textures, mesh=True)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    rng = np.random.default_rng(51)
    unit = lambda v: (np.asarray(v, F) / np.linalg.norm(v)).astype(F)
    roof_v = F([[-1.5, -1.5, 0.9], [0.2, -1.5, 1.3], [0.2, 1.5, 1.3], [-1.5, 1.5, 0.9]])
    roof = (roof_v, [[0, 1, 2], [0, 2, 3]], np.stack([unit([-0.23, 0.05 * k, 0.97]) for k in range(4)]))          # smooth, slightly varying normals
    wall = (F([[0.9, -1.5, -0.1], [0.9, 1.5, -0.1], [0.9, 1.5, 0.8], [0.9, -1.5, 0.8]]), [[0, 1, 2], [0, 2, 3]], np.tile(unit([-1, 0, 0.1]), (4, 1)))
    sh = dict(cast_shadow_rays=True, min_shadow_samples=4, n_shadow_samples=64) if shadows else {}
    from nerf_tex_amd.instancer import Instancer
    inst = Instancer(box["b_0"], box["b_1"], textures=list(textures), transformations=[m.tolist() for m in tr], instance_sampling_method="nearest",
                     mesh=(spec0.mesh_v, spec0.mesh_f), auxiliary_meshes=[(roof, ""), (wall, "")], **sh)
    v, f, nrm, kind = inst.meshes
    assert kind.tolist() == [0, 0, 1, 1, 1, 1] and f.max() == 11
    n, S, h = 160, 96, 0.02
    o, d = random_rays(51, n)
    P = spec0.n_parameters
    params = rng.uniform(0.2, 1.0, size=(n, P)).astype(F)
    ld = 1
    light = rng.normal(size=(n, 3)); light[:, 2] = np.abs(light[:, 2]) * 0.7 + 0.2
    params[:, ld:ld + 3] = light
    got = run_gpu(inst, o, d, params, S, h, seed=6)
    want = run_oracle(inst, box, o, d, params, S, h, 6, "nearest", textures, False, (v, f, nrm, kind), **sh)
    shaded = want[4][:, 0, 0]
    assert (shaded > 0).sum() > 20 and len(np.unique(shaded)) > 10 and (want[5][:, 0] == 1).sum() > (shaded > 0).sum()   # shaded, black and open endings
    assert np.isclose(shaded[shaded > 0].min(), 0.16, atol=1e-6) and shaded.max() > 0.5     # some in shadow or facing away (ambient only), some lit
    assert_same(got, want)
    assert inst.status() == 0


def test_patches_distributed_on_a_mesh_file(tmp_path):
    """The shipped configs' way in: mesh_path + patch_origins_path + patch_scale + jitter_amount (config_carpet_render.py:87-94) ->
    DistributeInstancesOnMesh on the host -> the same instancer as the explicit transformation list; the mesh culls."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.instancer import Instancer, distribute_instances_on_mesh
    tr0, v, f = synthetic.patch_sheet(8, extent=0.5)
    nrm = tr0[:, :3, 2] / np.linalg.norm(tr0[:, :3, 2], axis=-1, keepdims=True)
    uv = (v[:, :2] + 0.5).astype(F)
    mesh = tmp_path / "sheet.ply"
    mesh.write_text("ply\nformat ascii 1.0\nelement vertex 64\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n"
                    "property float nz\nproperty float s\nproperty float t\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % f.shape[0]
                    + "".join("%r %r %r %r %r %r %r %r\n" % tuple(float(x) for x in (*a, *b, *c)) for a, b, c in zip(v, nrm, uv))
                    + "".join(f"3 {t[0]} {t[1]} {t[2]}\n" for t in f))
    origins = v[::3] + F([0.01, -0.01, 0.0])
    org = tmp_path / "anchors.ply"
    org.write_text("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(origins)
                   + "".join("%r %r %r\n" % tuple(float(x) for x in o) for o in origins))
    b_0, b_1 = synthetic.PATCH_BOX
    a = Instancer(b_0, b_1, textures=["", "light"], mesh_path=str(mesh), patch_origins_path=str(org), patch_scale=0.09, jitter_amount=1.0,
                  instance_sampling_method="nearest_blend")
    want, scale = distribute_instances_on_mesh(v, f, nrm, uv, 0.09, origins, 1.0, 0)
    assert a.n_instances() == len(origins) == 22 and a.patch_scale == pytest.approx(0.09) and scale == pytest.approx(0.09)
    b = Instancer(b_0, b_1, textures=["", "light"], transformations=[m.tolist() for m in want], mesh=(v, f), instance_sampling_method="nearest_blend")
    assert all(np.array_equal(x, y) for x, y in zip(a.matrices(), b.matrices()))
    R = np.linalg.inv(a.matrices()[0][:, :3, :3])                                 # patch -> world: orthogonal columns of length 0.09
    assert np.allclose(np.einsum("kij,kil->kjl", R, R), 0.09 ** 2 * np.eye(3), atol=1e-6)
    o = np.tile(F([[0.05, 0.02, 3.0]]), (4, 1)); d = np.tile(F([[0, 0, -1]]), (4, 1))
    out = run_gpu(a, o, d, np.tile(F([[1, 0, 0, 1]]), (4, 1)), 64, 0.01, seed=1)
    assert out[8].all() and (out[5] == 1).all() and (out[3] > 0).sum() > 8       # marches the patches, ends on the sheet
    c = Instancer(b_0, b_1, mesh_path=str(mesh), patch_scale=-1.0)                 # no anchors: a patch per vertex; scale = the average edge length
    assert c.n_instances() == 64 and 0.1 < c.patch_scale < 0.2


# ------------------------------------------------------------------------------------------------------------------------------
# image textures (ABI v5): parameter textures on the instancer mesh, textured auxiliary meshes
# ------------------------------------------------------------------------------------------------------------------------------

def test_texture_known_answers_on_the_kernel():
    """The hand-computed cases of tests/test_oracle_instancer.py (a ramp texture on a flat sheet, two patches side by side), from the
    kernel: a query per step, and the two-segment interpolation."""
    from nerf_tex_amd.instancer import Instancer
    v = F([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]]); f = [[0, 1, 2], [0, 2, 3]]
    uv = (v[:, :2] + 2) / 4
    ramp_px = np.tile(np.asarray([0, 64, 128, 191, 255], np.uint8)[None, :], (3, 1))       # [height 3, width 5]: value = u
    o, d, par = F([[-3, 0, .5]]), F([[1, 0, 0]]), F([[2.0, 7.0, 0, 0, 1]])
    tr = [translate(x=-1).tolist(), translate(x=1).tolist()]
    for n_tex in (100000, 2):
        inst = Instancer([-1, -1, 0], [1, 1, 1], textures=[ramp_px, "", "light"], transformations=tr, instancer_mesh=(v, f, uv), patch_scale=1.0,
                         n_texture_samples=n_tex, min_texture_samples=4)
        assert inst.n_parameters == 5 and np.isclose(inst.patch_max_extent, np.sqrt(3))
        got = run_gpu(inst, o, d, par, 16, 0.5, seed=3)
        t = got[2][0]
        x = -3 + t[:8]
        assert np.allclose(got[9][0, :8, 0], 2 * (x + 2) / 4, atol=6e-3) and (got[9][0, 8:] == par[0]).all()     # the 8-bit ramp is u within 1 / 255
        box = dict(b_0=[-1, -1, 0], b_1=[1, 1, 1])
        want = run_oracle(inst, box, o, d, par, 16, 0.5, 3, textures=[ramp_px, "", "light"], mesh=(v, f), tex_mesh=(v, f, uv), patch_scale=1.0,
                          n_texture_samples=n_tex, min_texture_samples=4)
        assert_same(got, want)


@pytest.mark.parametrize("n_texture_samples", [24, 100000])                    # interpolated between texture samples / a lookup per step
@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("shadows", [False, True])
def test_parameter_textures_bit_for_bit(n_texture_samples, channels, shadows):
    """getParameters (instancer.cpp:640-667) through GetModelInput: two texture files (the first of `channels` channels -- the
    reference multiplies ONE parameter per file, by entry i of its list of all channels, :656-662), a light, with and without shadow
    rays; the lookups at the closest point of a waving, sheared-uv instancer mesh; every buffer bit for bit."""
    spec0 = random_scene(61, k=24, method="nearest")
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    tex_mesh = wavy_sheet()
    textures = [random_pixels(1, c=channels), "", random_pixels(2, h=9, w=7), "light"]
    tkw = dict(n_texture_samples=n_texture_samples, min_texture_samples=5)
    sh = dict(cast_shadow_rays=True, min_shadow_samples=4, n_shadow_samples=48) if shadows else {}
    patch_scale = 0.35
    inst = gpu_instancer(box, tr, textures=textures, instance_sampling_method="nearest", instancer_mesh=tex_mesh, patch_scale=patch_scale, **tkw, **sh)
    P = inst.n_parameters
    assert P == channels + 1 + 1 + 3 and inst.tex_idx == [0, channels + 1]
    n, S, h = 140, 128, 0.02
    o, d = random_rays(61, n)
    rng = np.random.default_rng(61)
    params = rng.uniform(0.3, 2.0, size=(n, P)).astype(F)
    light = rng.normal(size=(n, 3)); light[:, 2] = np.abs(light[:, 2]) * 0.7 + 0.1
    params[:, P - 3:] = light
    got = run_gpu(inst, o, d, params, S, h, seed=8)
    want = run_oracle(inst, box, o, d, params, S, h, 8, "nearest", textures, False, (tex_mesh[0], tex_mesh[1]), tex_mesh=tex_mesh,
                      patch_scale=patch_scale, **tkw, **sh)
    emitted = want[2] > 0
    ratio = want[9][..., 0][emitted] / np.repeat(params[:, None, 0], S, 1)[emitted]
    assert emitted.sum() > 2000 and ratio.min() < 0.3 and ratio.max() > 0.7 and len(np.unique(ratio)) > 500      # the texture shows
    if channels == 3:
        assert np.array_equal(want[9][..., 1][emitted], np.repeat(params[:, None, 1], S, 1)[emitted]) or n_texture_samples < S   # green, blue: never multiplied
    assert_same(got, want)
    assert inst.status() == 0


@pytest.mark.parametrize("channels,shadows", [(1, False), (3, False), (3, True), (4, True)])
def test_textured_auxiliary_meshes_bit_for_bit(channels, shadows):
    """AddMesh's texture (instancer.cpp:404, 725-733): the albedo of the closing sample at the hit's texture coordinates, three
    channels or the first for all; a second auxiliary mesh without a texture keeps 0.8."""
    spec0 = random_scene(71, k=16, method="nearest", textures=("", "light"), mesh=True)
    box = dict(b_0=spec0.b_0.tolist(), b_1=spec0.b_1.tolist())
    tr = [np.linalg.inv(m.astype(np.float64)).astype(F) for m in spec0.inv]
    unit = lambda v: (np.asarray(v, F) / np.linalg.norm(v)).astype(F)
    roof = (F([[-1.5, -1.5, 0.9], [0.2, -1.5, 1.3], [0.2, 1.5, 1.3], [-1.5, 1.5, 0.9]]), [[0, 1, 2], [0, 2, 3]],
            np.stack([unit([-0.23, 0.05 * k, 0.97]) for k in range(4)]), F([[0.1, 0.05], [0.9, 0.2], [0.95, 0.85], [0.05, 0.9]]))
    wall = (F([[0.9, -1.5, -0.1], [0.9, 1.5, -0.1], [0.9, 1.5, 0.8], [0.9, -1.5, 0.8]]), [[0, 1, 2], [0, 2, 3]], np.tile(unit([-1, 0, 0.1]), (4, 1)))
    px = random_pixels(5, h=10, w=14, c=channels)
    sh = dict(cast_shadow_rays=True, min_shadow_samples=4, n_shadow_samples=64) if shadows else {}
    from nerf_tex_amd.instancer import Instancer
    inst = Instancer(box["b_0"], box["b_1"], textures=["", "light"], transformations=[m.tolist() for m in tr], instance_sampling_method="nearest",
                     mesh=(spec0.mesh_v, spec0.mesh_f), auxiliary_meshes=[(roof, px), (wall, "")], **sh)
    v, f, nrm, kind = inst.meshes
    assert kind.tolist() == [0, 0, 1, 1, 1, 1] and inst.mesh_tex.tolist() == [-1, -1, 0, 0, -1, -1] and inst.mesh_prim.tolist() == [0, 1] * 3
    n, S, h = 160, 64, 0.03
    o, d = random_rays(71, n)
    rng = np.random.default_rng(71)
    params = rng.uniform(0.2, 1.0, size=(n, 4)).astype(F)
    light = rng.normal(size=(n, 3)); light[:, 2] = np.abs(light[:, 2]) * 0.7 + 0.2
    params[:, 1:4] = light
    got = run_gpu(inst, o, d, params, S, h, seed=6)
    want = run_oracle(inst, box, o, d, params, S, h, 6, "nearest", ("", "light"), False, (v, f, nrm, kind), aux_pixels=[px], **sh)
    col = want[4][:, 0]
    lit = col.max(-1) > 0
    assert lit.sum() > 20 and len(np.unique(col[lit, 0])) > 10
    if channels == 3:
        assert (np.abs(col[lit, 0] - col[lit, 1]) > 1e-3).any()                   # an RGB texture colours the sample
    else:
        assert np.array_equal(col[:, 0], col[:, 1]) and np.array_equal(col[:, 0], col[:, 2])
    assert_same(got, want)


def write_ply(path, v, f, normals=None, uv=None):
    cols = ["x", "y", "z"] + (["nx", "ny", "nz"] if normals is not None else []) + (["s", "t"] if uv is not None else [])
    rows = np.concatenate([np.asarray(v, F)] + ([np.asarray(normals, F)] if normals is not None else []) + ([np.asarray(uv, F)] if uv is not None else []), -1)
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\n" % len(rows) + "".join(f"property float {c}\n" for c in cols)
                 + "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % len(f)
                 + "".join(" ".join(repr(float(x)) for x in r) + "\n" for r in rows) + "".join(f"3 {t[0]} {t[1]} {t[2]}\n" for t in f))


def stand_in_scene(fam, root):
    """Stand-ins for the files a shipped render config names (LFS pointers in the reference's repository), written under root/meshes:
    a small waving sheet with normals and texture coordinates for the mesh, a few anchor points near it, a smooth one-channel image."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.png import write_png
    os.makedirs(os.path.join(root, "meshes"), exist_ok=True)
    grid, extent = (7, 0.30) if fam != "plush" else (6, 0.10)
    scale = 0.09 if fam != "plush" else 0.04
    tr0, v, f = synthetic.patch_sheet(grid, extent=extent, scale=scale)
    if fam == "plush":                                                            # a finer relief for the smaller patches
        v = v.copy(); v[:, 2] *= 0.25; tr0 = tr0.copy()
    nrm = tr0[:, :3, 2] / np.linalg.norm(tr0[:, :3, 2], axis=-1, keepdims=True)
    uv = ((v[:, :2] + extent) / (2 * extent)).astype(F)
    files = {"carpet": ("cloth_mesh.ply", "cloth_anchor_points.ply", "smooth_checkerboard.png"), "grass": ("terrain_mesh.ply", "terrain_anchor_points.ply", None),
             "grass_filtered": ("terrain_mesh.ply", "terrain_anchor_points.ply", None), "plush": ("stanford_bunny.ply", None, "checkerboard.png")}[fam]
    write_ply(os.path.join(root, "meshes", files[0]), v, f, nrm, uv)
    origins = None
    if files[1]:
        rng = np.random.default_rng(7)
        origins = (v[rng.choice(len(v), size=14, replace=False)] + rng.normal(size=(14, 3)) * [0.01, 0.01, 0.002]).astype(F)
        write_ply(os.path.join(root, "meshes", files[1]), origins, [])
    if files[2]:
        write_png(os.path.join(root, "meshes", files[2]), random_pixels(3, h=16, w=16)[..., 0])
    return v, f, nrm, uv, origins, extent


@pytest.mark.parametrize("fam", ["carpet", "grass", "grass_filtered", "plush"])
def test_shipped_render_configs_run_as_written(fam, tmp_path, monkeypatch):
    """The `renderer_config` block of every shipped render config (tests/golden/renderer_configs.json: the reference's own
    configs/config_<fam>_render.py, dumped by oracle/gen_golden.py), VERBATIM -- instancer.instancer.Instancer with its mesh_path,
    patch_origins_path, image textures, shadow rays, jitter -- through the reference's plugin path (Render -> Dataset -> ParamNerf ->
    InstanceRenderer -> Instancer) on stand-in files, against the whole pipeline restated: oracle rays -> oracle instancer (fed the
    library's instance matrices) -> float64 tail."""
    from nerf_tex_amd import synthetic, util
    from nerf_tex_amd.dataset import look_at
    from nerf_tex_amd.instancer import Instancer
    monkeypatch.chdir(tmp_path)                                                   # the configs name their files relative to the repository
    v, f, nrm, uv, origins, extent = stand_in_scene(fam, str(tmp_path))
    with open(os.path.join(os.path.dirname(__file__), "golden", "renderer_configs.json")) as fh:
        doc = json.load(fh)[fam]
    rc = doc["renderer_config"]
    npar = tuple(doc["model_n_parameters"])
    ic = rc["instancer_config"]
    assert ic["module"] == "instancer.instancer.Instancer"
    H = W = 14
    P = sum(npar)
    params = {"carpet": [1, 1, 1, .1, 0.3, 0.2, 1], "grass": [1, 6.0, 0.3, 0.4, 1.5], "grass_filtered": [4.0, 1, 0.3, 0.2, 1], "plush": [1, 1, 0.3, 0.2, 1]}[fam]
    cam = (6. if fam != "plush" else 2.5) * np.asarray([0.9165, 0., 0.4])
    c2w = look_at(cam)
    angle = 0.13 if fam != "plush" else 0.11
    emb = lambda n_: {'module': 'network.model.FourierFeatures', 'n_freq_bands': n_}
    hi = 0.3 if fam != "plush" else 0.08
    aabb0, aabb1 = [-extent * 1.6, -extent * 1.6, -.2 * hi / 0.3], [extent * 1.6, extent * 1.6, hi]
    config = {
        'module': 'network.render.Render', 'target_path': None,
        'test_dataset_config': {
            'module': 'network.dataset.Dataset',
            'data_loader_config': {'module': 'nerf_tex_amd.dataset.FromViews', 'height': H, 'width': W, 'angle': angle,
                                   'views': [{'pose': c2w, 'parameters': params}]},
            'pixel_sampler_config': {'module': 'network.pixel_sampler.Full'},
            'ray_sampler_config': {'module': 'network.ray_sampler.Proxy'},
            'proxy_config': {'module': 'network.proxy.AABB', 'b_0': aabb0, 'b_1': aabb1},
            'n_epochs': 1},
        'model_config': {'module': 'network.model.ParamNerf', 'pos_embedding': emb(10), 'dir_embedding': emb(4), 'param_embedding': emb(4),
                         'n_parameters': list(npar)},
        'renderer_config': dict(rc, density_scale=60.0),                          # (the one addition: the seeded stand-in weights are thin media)
        'logger_config': {'module': 'network.logger.Logger'},
    }
    model, mspec, wts = make_model(npar, dense_media=True)
    blob = synthetic.synthetic_weights(model.layer_table(), seed=0, dense_media=True)
    imgs = util.instantiate(dict(util.remap_reference_config(config), weights=blob, weights_order="keras_get_weights"))
    rgba = imgs[0][0].cpu().numpy().reshape(H * W, 4)
    # the same, restated
    S, step, patch_scale = rc["n_samples"], rc["step_size"], ic["patch_scale"]
    focal = orc.focal_from_angle(W, angle)
    ro, rd, t, cone = orc.proxy_rays(orc.full_pixels(H, W), H, W, focal, c2w.astype(F), aabb0, aabb1, F)
    keep = np.isfinite(t[:, 0])
    k = int(keep.sum())
    lib_side = Instancer(**{kk: vv for kk, vv in ic.items() if kk != "module"})
    assert lib_side.n_instances() == (len(origins) if origins is not None else len(v))
    sh = dict(cast_shadow_rays=True, min_shadow_samples=ic["min_shadow_samples"], n_shadow_samples=ic["n_shadow_samples"]) if ic["cast_shadow_rays"] else {}
    tkw = dict(min_texture_samples=ic.get("min_texture_samples", 4), n_texture_samples=ic.get("n_texture_samples", 512))
    par = np.tile(np.asarray([params], F), (k, 1))
    b = run_oracle(lib_side, dict(b_0=ic["b_0"], b_1=ic["b_1"]), ro[keep], rd[keep], par, S, step, 0, ic["instance_sampling_method"], ic["textures"],
                   False, (v, f), ray_index=(0, k, k), patch_scale=patch_scale, tex_mesh=(v, f, uv), **sh, **tkw)
    hitmask = b[8]
    rcol, ra = orc.instance_evaluate_model(wts, mspec, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], hitmask, b[9], cone[keep], rc.get("blur_idx"),
                                           patch_scale, 60.0, rc.get("density_reweighting", True), False, False, (1., 1., 1.), None, dtype=np.float64)
    want = np.zeros((H * W, 4)); want[keep, :3] = rcol; want[keep, 3] = ra
    assert orc.rel_linf(rgba, want) <= TOL
    emitted = b[2] > 0
    assert k > 0.7 * H * W and hitmask.sum() > 0.3 * k and emitted.sum() > 10 * hitmask.sum()
    assert want[:, 3].max() > 0.5
    if any(isinstance(x, str) and x.endswith(".png") for x in ic["textures"]):   # the texture modulates its parameter along the rays
        col = [i for i, x in enumerate(ic["textures"]) if x.endswith(".png")][0]
        assert len(np.unique(b[9][..., col][emitted])) > 50
    if ic["cast_shadow_rays"]:
        ld = lib_side._light_dir_idx
        assert (np.all(b[9][..., ld:ld + 3] == F([0, 0, -1]), axis=-1) & emitted).sum() > 0


@pytest.mark.parametrize("mode", ["interpolated", "per_step"])
def test_the_bench_scene_with_shadows_and_textures_against_the_c_restatement(mode):
    """The scene the instancer's numbers are measured on (48 x 48 patches on a waving sheet, 1024 steps of 0.002), now with shadow rays under a
    low sun AND a parameter texture, 1024 rays -- every ray, every buffer, bit for bit against the SECOND restatement (plain C,
    oracle/c/ntx_instancer_oracle.c; fast enough for a thousand rays through 2304 patches and 4418 triangles).  This is where the two-level
    cull, the shadow intervals and the candidate lists of the texture lookups all have work to do."""
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.dataset import look_at
    from nerf_tex_amd.instancer import Instancer
    from oracle import c_instancer as ci
    B0, B1 = synthetic.PATCH_BOX
    tr, v, f = synthetic.patch_sheet(48)
    yy, xx = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    px = (127.5 + 127.5 * np.sin(xx * 0.4) * np.sin(yy * 0.4)).astype(np.uint8)
    uv = ((v[:, :2] + 1.5) / 3.0).astype(F)
    n_sh, n_tx = (128, 256) if mode == "interpolated" else (100000, 100000)
    kw = dict(cast_shadow_rays=True, min_shadow_samples=8, n_shadow_samples=n_sh, min_texture_samples=8, n_texture_samples=n_tx)
    textures = [px, '', '', '', 'light']
    inst = Instancer(B0, B1, textures=textures, transformations=tr, instance_sampling_method="nearest", instancer_mesh=(v, f, uv), patch_scale=0.09, **kw)
    fam = synthetic.FAMILIES["carpet"]
    c2w = look_at(np.asarray(fam["cam"], F))
    focal = 800 / np.tan(fam["angle"] / 2) / 2
    side, S, h = (32 if mode == "interpolated" else 20), 1024, 0.002      # (every-step shadow and texture lookups: 16 ms of the C restatement a ray)
    r0 = (800 - side) // 2
    rows, cols = np.meshgrid(np.arange(r0, r0 + side), np.arange(r0, r0 + side), indexing="ij")
    ro, rd, t, cone = orc.proxy_rays(np.stack([rows.ravel(), cols.ravel()], -1), 800, 800, focal, c2w, [-1.7, -1.7, -.3], [1.7, 1.7, .4], F)
    n = side * side
    par = np.tile(np.asarray([fam["params"]], F), (n, 1)); par[:, 4:7] = (0.9, 0.2, 0.08)
    got = run_gpu(inst, ro, rd, par, S, h, seed=1)
    names, images = oracle_textures(textures)
    spec = io.make_spec(B0, B1, None, textures=names, images=images, instance_sampling_method="nearest", mesh=(v, f), matrices=inst.matrices(),
                        instancer_mesh=(v, f, uv), patch_scale=0.09, **kw)
    want = ci.get_model_input(spec, ro, rd, par, S, h, io.offset_uniforms(n, 1), io.choice_uniforms(n, S, 1))
    emitted = want[2] > 0
    dark = emitted & np.all(want[9][..., 4:7] == F([0, 0, -1]), axis=-1)
    assert emitted.sum() > n * 150 and 0.02 < dark.sum() / emitted.sum() < 0.6 and len(np.unique(want[9][..., 0][emitted])) > (10000 if side == 32 else 4000)
    assert_same(got, list(want))
    assert inst.status() == 0
