"""HIP path against the committed fixtures in tests/golden/ (float64 oracle outputs).  `-m gpu`."""

import json
import os

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from nerf_tex_amd import synthetic
from tests.common import EMB, TOL

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
G = os.path.join(os.path.dirname(__file__), "golden")


def dev():
    return torch.device("cuda", 0)


def d(a):
    return torch.as_tensor(np.asarray(a), device=dev())


def model_for(g):
    from nerf_tex_amd.model import ParamNerf
    m = ParamNerf(EMB(10), EMB(4), EMB(4), [int(v) for v in g["n_parameters"]])["model"]
    m.set_blob(synthetic.synthetic_weights(m.layer_table(), seed=int(g["weights_seed"]), dense_media=bool(g["weights_dense_media"])))
    return m


@pytest.mark.parametrize("family", ["carpet", "grass", "fur", "grass_filtered"])
def test_golden_family(family):
    from nerf_tex_amd.layer import FourierFeatures
    from nerf_tex_amd.proxy import AABB
    from nerf_tex_amd.ray_sampler import Proxy
    from nerf_tex_amd.renderer import Renderer
    g = np.load(os.path.join(G, f"golden_{family}.npz"))
    m = model_for(g)
    H, W, S = int(g["height"]), int(g["width"]), int(g["n_samples"])
    bi = None if int(g["blur_idx"]) < 0 else int(g["blur_idx"])
    # a2-a4: rays
    ro, rd, t, cone = Proxy(H, W, float(g["focal"]), AABB(g["b_0"], g["b_1"]))((0, H * W), g["c2w"])
    assert np.max(np.abs(ro.cpu().numpy() - g["rays_o"])) <= 1e-6 and np.max(np.abs(rd.cpu().numpy() - g["rays_d"])) <= 1e-6
    hit = g["hit"]
    assert np.array_equal(np.isfinite(t.cpu().numpy()[:, 0]), hit)
    assert np.max(np.abs(t.cpu().numpy()[hit] - g["t"][hit])) <= 1e-5
    # a8: encoding of the golden sample positions (first 8 rows, position block of pos_map)
    ff = FourierFeatures(10)(d(g["m_pos"][:8])).cpu().numpy()
    assert np.max(np.abs(ff - g["m_pos_map"][:, :63])) <= 2.5e-7
    # a9: MLP
    col, alp = m((d(g["m_pos"]), d(g["m_dirs"]), d(g["m_params"])))
    got = np.concatenate([col.cpu().numpy(), alp.cpu().numpy()], -1)
    assert orc.rel_linf(got, np.concatenate([g["m_color"], g["m_alpha"]], -1)) <= 2e-5
    # a5-a7, a10: fused render from the golden rays, with and without background
    r = Renderer(model=m, n_samples=S, perturb=False, blur_idx=bi)
    args = (d(g["rays_o"])[None], d(g["rays_d"])[None], d(g["t"])[None])
    for bk, ck, ak, col_bk in ((False, "color_pred", "alpha_pred", (1, 1, 1.)), (True, "color_pred_bkgd", "alpha_pred_bkgd", g["bkgd"])):
        out = r(*args, parameters=d(g["parameters"]), cone_scale=d(g["cone_scale"])[None], composite_bkgd=bk, bkgd_color=list(col_bk))
        r.raise_if_nonfinite()
        got = np.concatenate([out["color_pred"].cpu().numpy(), out["alpha_pred"].cpu().numpy()[..., None]], -1)
        want = np.concatenate([g[ck], g[ak][..., None]], -1)
        assert orc.rel_linf(got, want) <= TOL
        assert np.all(got[0][~hit][:, 3] == 0)


def test_golden_edge_composite():
    from nerf_tex_amd.renderer import Renderer
    g = np.load(os.path.join(G, "golden_edge.npz"))
    for exr in (0, 1):
        for bk in (0, 1):
            r = Renderer(model=None, map_exr=bool(exr), perturb=False)
            c, a, w = r.map_model_output(d(g["color"]), d(g["sigma"]), d(g["z"]), d(g["rays_d"]), bool(bk), list(g["bkgd"]))
            rc, ra, rw = g[f"color_exr{exr}_bk{bk}"], g[f"alpha_exr{exr}_bk{bk}"], g[f"weights_exr{exr}_bk{bk}"]
            scale = max(1.0, float(np.abs(rc).max()))           # elu+1 colours reach ~80
            assert np.max(np.abs(w.cpu().numpy() - rw)) <= 2e-6
            assert np.max(np.abs(a.cpu().numpy() - ra)) <= 2e-6
            assert np.max(np.abs(c.cpu().numpy() - rc)) / scale <= 2e-6


def test_golden_plumbing_image_through_render_harness():
    """BASELINE configs[0] (carpet 200x200x32) end to end through the reference-style config:
    Render -> Dataset(FromViews, Full, Proxy, AABB) -> ParamNerf -> Renderer -> RGBA."""
    from nerf_tex_amd import util
    g = np.load(os.path.join(G, "golden_plumbing.npz"))
    cam = json.load(open(os.path.join(G, "cameras_carpet.json")))
    blob = synthetic.synthetic_weights(orc.layer_table(orc.ModelSpec(n_parameters=(1, 6))), seed=0, dense_media=True)
    config = {
        "module": "network.render.Render", "target_path": None,
        "test_dataset_config": {
            "module": "network.dataset.Dataset",
            "data_loader_config": {"module": "nerf_tex_amd.dataset.FromViews", "height": 200, "width": 200, "angle": cam["angle"],
                                   "views": [{"pose": g["c2w"], "parameters": g["parameters"][0]}]},
            "pixel_sampler_config": {"module": "network.pixel_sampler.Full"},
            "ray_sampler_config": {"module": "network.ray_sampler.Proxy"},
            "proxy_config": {"module": "network.proxy.AABB", "b_0": cam["b_0"], "b_1": cam["b_1"]},
            "n_epochs": 1},
        "model_config": {"module": "network.model.ParamNerf", "pos_embedding": EMB(10), "dir_embedding": EMB(4),
                         "param_embedding": EMB(4), "n_parameters": [1, 6]},
        "renderer_config": {"module": "network.renderer.Renderer", "n_samples": 32, "perturb": False},
        "logger_config": {"module": "network.logger.Logger"},
    }
    with pytest.raises(ValueError, match="weights_order"):      # a flat blob must say which order it is in (ABI v1's differed, same size)
        util.instantiate(dict(util.remap_reference_config(config), weights=blob))
    imgs = util.instantiate(dict(util.remap_reference_config(config), weights=blob, weights_order="keras_get_weights"))
    assert len(imgs) == 1
    rgba = imgs[0][0].cpu().numpy()
    assert rgba.shape == (200, 200, 4)
    # North-star gate: match the reference's FLOAT32 render within 1e-4 rel-Linf.  TensorFlow cannot run, so
    # the float32 numpy restatement stands in for it (same float32 sample positions and encoder arguments).
    err32 = orc.rel_linf(rgba, g["rgba_f32"])
    # The arithmetic the kernel answers for, against the truth: the float64 network and composite on the float32 sample
    # points (elementwise float32 ray arithmetic has no summation order: every float32 run of the reference feeds its network
    # these bits; oracle render_rays: points_dtype).  Same strict gate.
    err_net = orc.rel_linf(rgba, g["rgba_net64"])
    # Against the all-float64 image there is, on top, the rounding of the INPUTS -- sample positions |x| <= 6 rounded to
    # float32 (|dx| <= 2^-24 |x| per operation of o + d z) in front of sin(2^9 x): an error model of the reference's own
    # float32, measured between the two float64 images (no kernel involved): 1.0e-4 on this image.
    err64 = orc.rel_linf(rgba, g["rgba"])
    input_floor = float(g["input_floor_rel_linf"])
    print(f"plumbing image: HIP vs f32 restatement {err32:.3e}, vs f64 network on f32 points {err_net:.3e}, vs all-f64 {err64:.3e} "
          f"(input-rounding floor {input_floor:.3e})")
    assert err32 <= TOL, err32
    assert err_net <= TOL, err_net
    assert err64 <= err_net + input_floor * (1 + 1e-6) + 1e-9, (err64, err_net, input_floor)   # the triangle inequality of that model, nothing looser
    assert abs(float(rgba.astype(np.float64).sum()) - float(g["rgba_f64_sum"])) / float(g["rgba_f64_sum"]) <= 1e-5
