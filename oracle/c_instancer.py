"""ctypes wrapper of the plain-C second restatement of the patch instancer (oracle/c/ntx_instancer_oracle.c).  TEST INFRASTRUCTURE ONLY.

`get_model_input(spec, ...)` takes the `InstancerSpec` of oracle/instancer_oracle.py and the same draws and returns the same tuple:
tests/test_oracle_instancer.py requires the two restatements to agree element for element.  PARITY UNPINNED (see both files' headers)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libntx_instancer_oracle_c.so")
_fp, _ip, _up = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
_fpp = C.POINTER(_fp)


class _Spec(C.Structure):
    _fields_ = [("n_inst", C.c_int), ("inv", _fp), ("dir_t", _fp), ("origins", _fp), ("b0", C.c_float * 3), ("b1", C.c_float * 3),
                ("n_parameters", C.c_int), ("light_dir_idx", C.c_int), ("light_strength_idx", C.c_int), ("sample_method", C.c_int),
                ("use_mean_distance", C.c_int), ("patch_scale", C.c_float),
                ("n_mesh_v", C.c_int), ("n_mesh_f", C.c_int), ("mesh_v", _fp), ("mesh_f", _ip), ("mesh_n", _fp), ("mesh_kind", _up),
                ("mesh_prim", _ip), ("mesh_uv", _fp), ("mesh_tex", _ip),
                ("cast_shadow_rays", C.c_int), ("min_shadow_samples", C.c_int), ("n_shadow_samples", C.c_int),
                ("n_tex_files", C.c_int), ("tex_idx", _ip), ("n_textures", C.c_int), ("tex_data", _fpp), ("tex_rows", _ip), ("tex_cols", _ip),
                ("has_instancer", C.c_int), ("n_inst_v", C.c_int), ("n_inst_f", C.c_int), ("inst_v", _fp), ("inst_f", _ip), ("inst_uv", _fp),
                ("patch_max_extent", C.c_float), ("min_texture_samples", C.c_int), ("n_texture_samples", C.c_int),
                ("n_aux_sets", C.c_int), ("aux_first", _ip), ("aux_channels", _ip), ("aux_data", _fpp), ("aux_rows", _ip), ("aux_cols", _ip)]


def _lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "c", "ntx_instancer_oracle.c")):
        subprocess.run(["make", "-C", _HERE], check=True)
    lib = C.CDLL(_SO)
    lib.io_get_model_input.restype = C.c_int
    lib.io_inverse4_float.restype = C.c_int
    return lib


def _tex_arrays(mats, keep):
    """(pointer array, rows, cols) over channel matrices [rows, cols] float32."""
    mats = [np.ascontiguousarray(m, dtype=np.float32) for m in mats]
    keep += mats
    ptrs = (_fp * max(len(mats), 1))(*[m.ctypes.data_as(_fp) for m in mats])
    rows = np.asarray([m.shape[0] for m in mats] or [0], np.int32); cols = np.asarray([m.shape[1] for m in mats] or [0], np.int32)
    keep += [ptrs, rows, cols]
    return ptrs, rows, cols


def get_model_input(spec, rays_o, rays_d, parameters, n_samples, step_size, u_offset, u_choice=None):
    keep = []
    f32 = lambda a, shape=None: (lambda x: (keep.append(x), x)[1])(np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape) if shape else np.asarray(a, np.float32)))
    i32 = lambda a: (lambda x: (keep.append(x), x)[1])(np.ascontiguousarray(np.asarray(a, np.int32)))
    fp = lambda a: a.ctypes.data_as(_fp) if a is not None else None
    ip = lambda a: a.ctypes.data_as(_ip) if a is not None else None
    s = _Spec()
    k = spec.inv.shape[0]
    inv, dir_t, org = f32(spec.inv, (k, 16)), f32(spec.dir_t, (k, 9)), f32(spec.origins, (k, 3))
    s.n_inst, s.inv, s.dir_t, s.origins = k, fp(inv), fp(dir_t), fp(org)
    s.b0 = (C.c_float * 3)(*[float(v) for v in spec.b_0]); s.b1 = (C.c_float * 3)(*[float(v) for v in spec.b_1])
    s.n_parameters, s.light_dir_idx, s.light_strength_idx = spec.n_parameters, spec.light_dir_idx, spec.light_strength_idx
    s.sample_method, s.use_mean_distance, s.patch_scale = spec.sample_method, int(spec.use_mean_distance), float(spec.patch_scale)
    if spec.mesh_v is not None:
        mv, mf = f32(spec.mesh_v), i32(spec.mesh_f)
        s.n_mesh_v, s.n_mesh_f, s.mesh_v, s.mesh_f = mv.shape[0], mf.shape[0], fp(mv), ip(mf)
        if spec.mesh_n is not None:
            mn = f32(spec.mesh_n); mk = np.ascontiguousarray(spec.mesh_kind, dtype=np.uint8); keep.append(mk)
            s.mesh_n, s.mesh_kind = fp(mn), mk.ctypes.data_as(_up)
        if spec.mesh_prim is not None:
            s.mesh_prim = ip(i32(spec.mesh_prim))
        if spec.mesh_tex is not None:
            s.mesh_uv, s.mesh_tex = fp(f32(spec.mesh_uv)), ip(i32(spec.mesh_tex))
            flat, first, nch = [], [], []
            for tex in spec.aux_textures:
                first.append(len(flat)); nch.append(len(tex)); flat += list(tex)
            s.n_aux_sets = len(first)
            s.aux_first, s.aux_channels = ip(i32(first or [0])), ip(i32(nch or [0]))
            s.aux_data, rows, cols = _tex_arrays(flat, keep)
            s.aux_rows, s.aux_cols = ip(rows), ip(cols)
    s.cast_shadow_rays, s.min_shadow_samples, s.n_shadow_samples = int(spec.cast_shadow_rays), spec.min_shadow_samples, spec.n_shadow_samples
    s.n_tex_files = len(spec.tex_idx or [])
    s.tex_idx = ip(i32(spec.tex_idx or [0]))
    s.n_textures = len(spec.textures or [])
    s.tex_data, rows, cols = _tex_arrays(spec.textures or [], keep)
    s.tex_rows, s.tex_cols = ip(rows), ip(cols)
    s.has_instancer = int(spec.inst_v is not None)
    if spec.inst_v is not None:
        iv, jf, uv = f32(spec.inst_v), i32(spec.inst_f), f32(spec.inst_uv)
        s.n_inst_v, s.n_inst_f, s.inst_v, s.inst_f, s.inst_uv = iv.shape[0], jf.shape[0], fp(iv), ip(jf), fp(uv)
    s.patch_max_extent = float(spec.patch_max_extent)
    s.min_texture_samples, s.n_texture_samples = spec.min_texture_samples, spec.n_texture_samples
    # the buffers as instancer.pyx:41-50 makes them
    rays_o = f32(rays_o); rays_d = np.asarray(rays_d, np.float32)
    n, S, P = rays_o.shape[0], int(n_samples), spec.n_parameters
    parameters = np.asarray(parameters, np.float32).reshape(n, -1)
    rays_d_map = np.ascontiguousarray(np.repeat(rays_d[:, None, :], S, axis=1))
    t = np.zeros((n, S), np.float32); dists = np.zeros((n, S), np.float32); pts = np.zeros((n, S, 3), np.float32)
    color = np.zeros((n, 1, 3), np.float32); density = np.zeros((n, 1), np.float32)
    weight = np.ones((n, S), np.float32); iid = np.zeros((n, S), np.int32); hit = np.zeros(n, np.uint8)
    pm = np.ascontiguousarray(np.repeat(parameters[:, None, :], S, axis=1))
    uo = f32(u_offset)
    uc = f32(u_choice if u_choice is not None else np.zeros((n, S), np.float32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = _lib().io_get_model_input(C.byref(s), n, S, C.c_float(step_size), p(rays_o), p(rays_d_map), p(t), p(dists), p(pts), p(color), p(density),
                                   p(weight), p(iid), p(hit), p(pm), p(uo), p(uc))
    if rc:
        raise IndexError("a ray starts a segment that has no length (instancer.cpp:990, 1001 read past segment_lengths)")
    return rays_d_map, pts, t, dists, color, density, weight, iid, hit.astype(bool), pm


def inverse4_float(m):
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(4, 4)
    out = np.zeros((4, 4), np.float32)
    rc = _lib().io_inverse4_float(m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("singular")
    return out
