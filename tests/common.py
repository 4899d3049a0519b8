"""Shared helpers of the test-suite: seeded inputs for both sides, spec conversion."""

import numpy as np

from oracle import nerftex_oracle as orc
from nerf_tex_amd import synthetic
from nerf_tex_amd.model import ParamNerf, Nerf

EMB = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}   # as in the reference configs

TOL = 1e-4   # BASELINE.json north_star: <= 1e-4 relative L-inf (float32) vs the reference render


def make_model(n_parameters=(1, 6), kind="ParamNerf", seed=0, dense_media=False, arch=None, freqs=None):
    """(product model with synthetic weights, oracle spec, oracle weight list).  `arch`: depth / width / skips / color_depth other
    than the reference configs' 8 / 256 / [4] / 1 (model.py:58, :9); `freqs`: n_freq_bands of the position / direction / parameter
    embeddings other than the configs' 10 / 4 / 4 (layer.py:11)."""
    if freqs:
        pf, df, qf = freqs
        a = dict(arch or {})
        spec_kw = dict(depth=a.get("depth", 8), width=a.get("width", 256), skips=tuple(a.get("skips", (4,))), pos_freq=pf, dir_freq=df)
        if kind == "Nerf":
            model = Nerf(EMB(pf), EMB(df), depth=spec_kw["depth"], width=spec_kw["width"], skips=list(spec_kw["skips"]))["model"]
            spec = orc.ModelSpec(kind="Nerf", n_parameters=(0, 0), **spec_kw)
        else:
            ipe = kind == "IPE"
            pos = {"module": "network.layer.IntegratedPositionalEncoding", "n_freq_bands": pf} if ipe else EMB(pf)
            pk = dict(param_depth=a.get("param_depth", 0), param_width=a.get("param_width", 128))
            model = ParamNerf(pos, EMB(df), EMB(qf), list(n_parameters), n_pos=6 if ipe else 3, depth=spec_kw["depth"], width=spec_kw["width"],
                              skips=list(spec_kw["skips"]), color_depth=a.get("color_depth", 1), **pk)["model"]
            spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(n_parameters), color_depth=a.get("color_depth", 1), param_freq=qf,
                                 **(dict(n_pos=6, pos_encoding="ipe") if ipe else {}), **spec_kw, **pk)
    elif arch:
        a = dict(arch)
        spec_kw = dict(depth=a.get("depth", 8), width=a.get("width", 256), skips=tuple(a.get("skips", (4,))))
        if kind == "Nerf":
            model = Nerf(EMB(10), EMB(4), depth=spec_kw["depth"], width=spec_kw["width"], skips=list(spec_kw["skips"]))["model"]
            spec = orc.ModelSpec(kind="Nerf", n_parameters=(0, 0), **spec_kw)
        else:
            pk = dict(param_depth=a.get("param_depth", 0), param_width=a.get("param_width", 128))
            model = ParamNerf(EMB(10), EMB(4), EMB(4), list(n_parameters), depth=spec_kw["depth"], width=spec_kw["width"],
                              skips=list(spec_kw["skips"]), color_depth=a.get("color_depth", 1), **pk)["model"]
            spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(n_parameters), color_depth=a.get("color_depth", 1), **spec_kw, **pk)
    elif kind == "IPE":                      # mip variant: IntegratedPositionalEncoding on (mean, covariance)
        ipe = {"module": "network.layer.IntegratedPositionalEncoding", "n_freq_bands": 10}
        model = ParamNerf(ipe, EMB(4), EMB(4), list(n_parameters), n_pos=6)["model"]
        spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(n_parameters), n_pos=6, pos_encoding="ipe")
    elif kind == "Nerf":
        model = Nerf(EMB(10), EMB(4))["model"]
        spec = orc.ModelSpec(kind="Nerf", n_parameters=(0, 0))
    else:
        model = ParamNerf(EMB(10), EMB(4), EMB(4), list(n_parameters))["model"]
        spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(n_parameters))
    assert model.layer_table() == orc.layer_table(spec)
    blob = synthetic.synthetic_weights(model.layer_table(), seed=seed, dense_media=dense_media)
    model.set_blob(blob)
    return model, spec, orc.split_blob(spec, blob)


def random_samples(m, n_params, seed=3, box=1.5):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-box, box, size=(m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    params = rng.uniform(0, 1, size=(m, max(n_params, 0))).astype(np.float32)
    return pos, d.astype(np.float32), params


def camera_rays(family, height, width, dtype=np.float32, angle_scale=2.5):
    """The camera grid of a BASELINE config family; the field of view is widened by `angle_scale`
    so that the grid holds rays that miss the AABB as well as rays that hit it."""
    fam = synthetic.FAMILIES[family]
    c2w = orc.look_at(fam["cam"], dtype=np.float32)
    focal = orc.focal_from_angle(width, fam["angle"] * angle_scale)
    loc = orc.full_pixels(height, width)
    return orc.proxy_rays(loc, height, width, focal, c2w, fam["b_0"], fam["b_1"], dtype), c2w, focal


def importance_depths(z_merged, z_coarse):
    """The n_importance depths of a merged, sorted [n, S + NI] array (renderer.py:130) given the S coarse ones it contains
    (multiset difference per ray, order kept)."""
    out = []
    for m, c in zip(np.asarray(z_merged), np.asarray(z_coarse)):
        res, i = [], 0
        for v in m:
            if i < len(c) and v == c[i]:
                i += 1
            else:
                res.append(v)
        assert i == len(c), "the merged depths do not contain the coarse ones"
        out.append(res)
    return np.asarray(out)
