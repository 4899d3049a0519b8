"""ctypes wrapper of the plain-C restatement (oracle/c/).  TEST INFRASTRUCTURE ONLY.

Second, independent reading of the reference source next to the numpy one in nerftex_oracle.py;
`tests/test_oracle.py` requires the two to agree.  PARITY UNPINNED, see nerftex_oracle.py."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libntx_oracle_c.so")


def _lib():
    if not os.path.exists(_SO):
        subprocess.run(["make", "-C", _HERE], check=True)
    return C.CDLL(_SO)


def _desc(spec):
    kind = 1 if spec.kind == "Nerf" else 0
    assert len(spec.skips) == 1
    return (C.c_int * 10)(kind, spec.n_geo, spec.n_app, spec.pos_freq, spec.dir_freq, spec.param_freq, spec.depth,
                          spec.width, spec.skips[0], spec.color_depth)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def model(spec, blob, pos, dirs, params, dtype=np.float64):
    sfx = "_f64" if dtype == np.float64 else "_f32"
    pos, dirs, params, blob = _f32(pos), _f32(dirs), _f32(params), _f32(blob)
    m = pos.shape[0]
    color = np.empty((m, 3), dtype); alpha = np.empty((m, 1), dtype)
    getattr(_lib(), "ntxo_model" + sfx)(_desc(spec), _p(blob), _p(pos), _p(dirs), _p(params), C.c_long(m), _p(color), _p(alpha))
    return color, alpha


def composite(color, sigma, z, rays_d, map_exr, composite_bkgd, bkgd, dtype=np.float64):
    sfx = "_f64" if dtype == np.float64 else "_f32"
    color, sigma, z, rays_d = _f32(color), _f32(sigma), _f32(z), _f32(rays_d)
    bkgd = np.ascontiguousarray(bkgd, dtype=np.float64)
    n, S = sigma.shape
    c = np.empty((n, 3), dtype); a = np.empty((n,), dtype); w = np.empty((n, S), dtype)
    getattr(_lib(), "ntxo_composite" + sfx)(_p(color), _p(sigma), _p(z), _p(rays_d), C.c_long(n), S, int(map_exr),
                                            int(composite_bkgd), _p(bkgd), _p(c), _p(a), _p(w))
    return c, a, w


def render_rays(spec, blob, rays_o, rays_d, t, params_per_ray, cone, n_samples, blur_idx=None, map_exr=False,
                composite_bkgd=False, bkgd=(1., 1., 1.), dtype=np.float64):
    """All rays must hit (no culling here: that is Renderer.__call__'s job)."""
    sfx = "_f64" if dtype == np.float64 else "_f32"
    rays_o, rays_d, t, pp, cone, blob = map(_f32, (rays_o, rays_d, t, params_per_ray, cone, blob))
    bkgd = np.ascontiguousarray(bkgd, dtype=np.float64)
    n = rays_o.shape[0]
    c = np.empty((n, 3), dtype); a = np.empty((n,), dtype)
    getattr(_lib(), "ntxo_render_rays" + sfx)(_desc(spec), _p(blob), _p(rays_o), _p(rays_d), _p(t), _p(pp), _p(cone),
                                              C.c_long(n), n_samples, -1 if blur_idx is None else blur_idx, int(map_exr),
                                              int(composite_bkgd), _p(bkgd), _p(c), _p(a))
    return c, a
