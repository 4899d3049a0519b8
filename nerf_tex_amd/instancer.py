"""The patch instancer on the GPU (reference: instancer/instancer.pyx `Instancer` over instancer/src/instancer.cpp `C_Instancer`).

Same constructor keywords and the same `get_model_input` / `n_instances` contract as the reference's Cython class, so a render
config names `nerf_tex_amd.instancer.Instancer` where it named `instancer.instancer.Instancer`; everything runs in
`ntx_instancer_model_input` (include/nerftex.h) and the ten buffers stay in HBM, where `InstanceRenderer` hands them to
`ntx_render_instanced` -- the reference walks its rays through Embree on one CPU thread and uploads ~70 bytes per (ray, step).

What is built: explicit `transformations` (instancer.pyx:19-20), the JSON file the reference's `transformation_export_path`
writes (instancer.cpp:1040-1061), or `mesh_path` [+ `patch_origins_path`, `patch_scale`, `jitter_amount`]: DistributeInstancesOnMesh
(instancer.cpp:233-390) on the host, from a PLY with vertex normals and texture coordinates; a culling mesh as arrays or a PLY file;
`auxiliary_meshes` (flat-shaded; a PLY with vertex normals, or arrays); the three `instance_sampling_method`s;
`use_mean_distance`; '' / 'light' / 'point' / image entries of `textures` (PNG files: parameter textures looked up on the instancer
mesh, instancer.cpp:640-667, with `min_texture_samples` / `n_texture_samples`; they apply when `mesh_path` is given, as in the
reference, :911); `cast_shadow_rays` with `min_shadow_samples` / `n_shadow_samples`; textured auxiliary meshes (:725-733).  All four
shipped render configs' `instancer_config` blocks construct as written (their mesh and texture files are LFS pointers in the
reference's repository: bring the files).  Refused (NtxError, NTX_E_UNSUPPORTED): more than four texture files; images that are not PNG.
"""

from __future__ import annotations

import ctypes as C
import json
from typing import Optional, Sequence

from . import _lib

SAMPLING_METHODS = {"random": 0, "nearest": 1, "nearest_blend": 2}        # instancer.pyx:14


def load_texture(source):
    """loadTexture (instancer.cpp:34-50): the channels of an image as float32 matrices [width, height] of value / 255, element (x, y
    counted from the bottom row) -- stb's pixels [height * width, channels] mapped column-major to (width, height) per channel, then
    `.rowwise().reverse()`.  `source`: a PNG file, or the pixels themselves [height, width(, channels)] uint8."""
    import numpy as np
    if isinstance(source, str):
        if not source.lower().endswith(".png"):
            raise _lib.NtxError(_lib.NTX_E_UNSUPPORTED, f"texture {source!r}: only PNG files are read (nerf_tex_amd/png.py)")
        from .png import read_png
        px = read_png(source)
    else:
        px = np.asarray(source, np.uint8)
        px = px[:, :, None] if px.ndim == 2 else px
    f = px.astype(np.float32) / np.float32(255.0)
    return [np.ascontiguousarray(f[::-1, :, c].T) for c in range(px.shape[2])]


def parse_textures(textures: Sequence):
    """(n_parameters, light_dir_parameter_idx, light_strength_parameter_idx, texture_parameter_idxs, channel matrices) of a `textures`
    list (instancer.cpp:74-92): '' = one parameter, 'light' = three, 'point' = four, anything else an image whose channels take one
    parameter each (an entry may also be the pixels themselves, [height, width(, channels)] uint8)."""
    n, light_dir, light_strength = 0, -1, -1
    idx, mats = [], []
    for path in textures:
        if isinstance(path, str) and path == "light":
            light_dir = n; n += 3
        elif isinstance(path, str) and path == "point":
            light_strength = n; light_dir = n + 1; n += 4
        elif not isinstance(path, str) or path != "":
            tex = load_texture(path)
            mats += tex
            idx.append(n)
            n += len(tex)
        else:
            n += 1
    return n, light_dir, light_strength, idx, mats


def _textures_struct(mats):
    """A ctypes array of ntx_texture over float32 matrices (kept alive by the caller)."""
    arr = (_lib.Texture * max(len(mats), 1))()
    for i, m in enumerate(mats):
        arr[i].texels = m.ctypes.data_as(C.POINTER(C.c_float)); arr[i].rows, arr[i].cols = m.shape
    return arr


class Instancer:
    """instancer.instancer.Instancer (instancer.pyx:6-54)."""

    device_native = True          # InstanceRenderer: inputs and outputs are torch tensors on the GPU, nothing goes through the host

    def __init__(self, b_0, b_1, cast_shadow_rays: bool = False, textures: Sequence[str] = (), transformations=(),
                 mesh_path: Optional[str] = None, patch_scale: float = 1., patch_origins_path: str = '',
                 min_shadow_samples: int = 4, n_shadow_samples: int = 512, min_texture_samples: int = 4,
                 n_texture_samples: int = 512, jitter_amount: float = 0, instance_sampling_method: str = 'random',
                 use_mean_distance: bool = False, auxiliary_meshes=(), transformation_export_path: Optional[str] = None,
                 transformations_path: Optional[str] = None, mesh=None, instancer_mesh=None, seed: int = 0, device: int = 0) -> None:
        """Beyond the reference's keywords: `transformations_path` (a list ExportTransformations wrote), `mesh` = (vertices, faces) of a
        culling mesh as arrays, `instancer_mesh` = (vertices, faces, uv): the state DistributeInstancesOnMesh leaves behind (the mesh
        culls, parameter textures are looked up on it, `patch_scale` scales the lookup radius, :246) without placing patches -- they
        come from `transformations`; `seed` (the reference's C++ default 0, not exposed by its Cython class), `device`."""
        import numpy as np
        if instance_sampling_method not in SAMPLING_METHODS:
            raise ValueError(f"instance_sampling_method must be one of {sorted(SAMPLING_METHODS)}")
        n_par, light_dir, light_strength, tex_idx, tex_mats = parse_textures(textures)
        if len(tex_idx) > 4:
            raise _lib.NtxError(_lib.NTX_E_UNSUPPORTED, f"{len(tex_idx)} texture files in `textures`: at most 4 are built")
        tr = [np.asarray(m, np.float32).reshape(4, 4) for m in transformations]
        distributed = False
        tex_mesh = None                                # (vertices, faces, uv) parameter textures are looked up on
        if transformations_path is not None:           # what ExportTransformations wrote (patch -> world, instancer.cpp:1040-1061)
            with open(transformations_path) as f:
                tr += [np.asarray(m, np.float32).reshape(4, 4) for m in json.load(f)]
            distributed = True
        elif mesh_path is not None:                    # DistributeInstancesOnMesh (instancer.cpp:233-390), on the host
            v_, f_, n_, uv_ = read_ply(mesh_path, normals=True, uv=True)
            origins = None
            if patch_origins_path:
                try:
                    origins = read_ply(patch_origins_path)[0]
                except (OSError, ValueError):           # the reference falls back to the vertices when it cannot read the file (:281, 341)
                    origins = None
            placed, scale_ = distribute_instances_on_mesh(v_, f_, n_, uv_, patch_scale, origins, jitter_amount, seed)
            tr += list(placed)
            patch_scale = scale_
            distributed = True
            tex_mesh = (v_, f_, uv_)
            if mesh is None:
                mesh = (v_, f_)
        elif instancer_mesh is not None:
            tex_mesh = tuple(np.asarray(a) for a in instancer_mesh)
            distributed = True
            if mesh is None:
                mesh = (tex_mesh[0], tex_mesh[1])
        self._tr = np.ascontiguousarray(np.stack(tr) if tr else np.zeros((0, 4, 4), np.float32))
        # only DistributeInstancesOnMesh stores the scale (instancer.cpp:236); it widens nearest_blend's transition (:697)
        self.patch_scale = float(patch_scale) if distributed else 1.0
        self.n_parameters = n_par
        self._light_dir_idx, self._light_strength_idx = light_dir, light_strength
        self.device = int(device)
        self.seed = int(seed)
        self._calls = 0
        desc = _lib.InstancerDesc()
        desc.size = C.sizeof(_lib.InstancerDesc)
        desc.b_0 = (C.c_float * 3)(*[float(v) for v in b_0]); desc.b_1 = (C.c_float * 3)(*[float(v) for v in b_1])
        desc.n_parameters, desc.light_dir_parameter_idx, desc.light_strength_parameter_idx = n_par, light_dir, light_strength
        desc.instance_sample_method = SAMPLING_METHODS[instance_sampling_method]
        desc.use_mean_distance = int(bool(use_mean_distance)); desc.cast_shadow_rays = int(bool(cast_shadow_rays))
        desc.patch_scale = self.patch_scale
        desc.min_shadow_samples, desc.n_shadow_samples = int(min_shadow_samples), int(n_shadow_samples)
        self._h = C.c_void_p()
        _lib.check(_lib.lib.ntx_instancer_create(C.byref(desc), self._tr.ctypes.data_as(C.POINTER(C.c_float)), self._tr.shape[0],
                                                 self.device, C.byref(self._h)))
        base = mesh if mesh is not None else (read_ply(mesh_path) if mesh_path is not None else None)
        # AddMesh (instancer.cpp:393-417): (path | (V, F, N[, UV]), texture path | pixels | '')
        has_tex = lambda t: not isinstance(t, str) or t != ""
        aux = [(read_ply(path, normals=True, uv=has_tex(t)) if isinstance(path, str) else tuple(path)) for path, t in auxiliary_meshes]
        self.meshes = None; self.mesh_prim = None; self.mesh_uv = None; self.mesh_tex = None; self.aux_textures = []
        if aux:
            self.set_meshes(base, aux, [load_texture(t) if has_tex(t) else None for _, t in auxiliary_meshes])
        elif base is not None:
            self.set_mesh(*base)
        # parameter textures (getParameters, :640-667): only after DistributeInstancesOnMesh (:911)
        self.tex_idx, self.textures, self.inst_mesh = tex_idx, tex_mats, tex_mesh
        e = np.maximum(np.asarray(b_0, np.float32), np.asarray(b_1, np.float32))
        self.patch_max_extent = np.float32(np.sqrt(e[0] * e[0] + (e[1] * e[1] + e[2] * e[2])))          # :69 (Eigen pairs x0^2 + (x1^2 + x2^2))
        if distributed:
            self.patch_max_extent = np.float32(self.patch_max_extent * np.float32(self.patch_scale))    # :246
        self.min_texture_samples, self.n_texture_samples = int(min_texture_samples), int(n_texture_samples)
        if tex_idx and tex_mesh is not None:
            v = np.ascontiguousarray(np.asarray(tex_mesh[0], np.float32).reshape(-1, 3)); f = np.ascontiguousarray(np.asarray(tex_mesh[1], np.int32).reshape(-1, 3))
            uv = np.ascontiguousarray(np.asarray(tex_mesh[2], np.float32).reshape(-1, 2))
            used = [tex_mats[i] for i in range(len(tex_idx))]            # file i multiplies by entry i of the list of ALL channels (:656-662)
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
            _lib.check(_lib.lib.ntx_instancer_set_parameter_textures(
                self._h, fp(v), fp(uv), v.shape[0], f.ctypes.data_as(C.POINTER(C.c_int32)), f.shape[0], float(self.patch_max_extent), len(tex_idx),
                (C.c_int32 * len(tex_idx))(*tex_idx), _textures_struct(used), self.min_texture_samples, self.n_texture_samples))
        if transformation_export_path is not None:
            self.export_transformations(transformation_export_path)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None and getattr(_lib, "lib", None) is not None:   # (interpreter shutdown)
            _lib.lib.ntx_instancer_destroy(h)
            self._h = None

    def n_instances(self) -> int:                                                        # instancer.pyx:32-33
        return int(_lib.lib.ntx_instancer_count(self._h))

    def set_mesh(self, vertices, faces) -> None:
        """The instancer mesh (instancer.cpp:369-389): rays end at it with an opaque black sample."""
        import numpy as np
        v = np.ascontiguousarray(np.asarray(vertices, np.float32).reshape(-1, 3))
        f = np.ascontiguousarray(np.asarray(faces, np.int32).reshape(-1, 3))
        _lib.check(_lib.lib.ntx_instancer_set_mesh(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), v.shape[0],
                                                   f.ctypes.data_as(C.POINTER(C.c_int32)), f.shape[0]))
        self.meshes = (v, f, None, np.zeros(f.shape[0], np.uint8)); self.mesh_prim = np.arange(f.shape[0])

    def set_meshes(self, instancer_mesh, auxiliary, textures=None) -> None:
        """The instancer mesh (vertices, faces) or None, and auxiliary meshes [(vertices, faces, vertex normals[, uv]), ...] (AddMesh,
        instancer.cpp:393-417): all of them cull and cast shadows; a ray that ends on an auxiliary mesh gets a shaded closing sample
        (shadeMesh, :716-743), its albedo 0.8 or `textures[i]` (the channel matrices of load_texture) at the hit's texture coordinates."""
        import numpy as np
        textures = list(textures) if textures is not None else [None] * len(auxiliary)
        vs, ns, fs, ks, us, ps, ts, sets, base = [], [], [], [], [], [], [], [], 0
        meshes = ([(instancer_mesh[0], instancer_mesh[1], None, None, 0, None)] if instancer_mesh is not None else []) + \
                 [(m[0], m[1], m[2], m[3] if len(m) > 3 else None, 1, t) for m, t in zip(auxiliary, textures)]
        for v, f, n, uv, kind, tex in meshes:
            v = np.asarray(v, np.float32).reshape(-1, 3); f = np.asarray(f, np.int32).reshape(-1, 3)
            if kind and n is None:
                raise ValueError("an auxiliary mesh needs vertex normals (the reference shades with them, instancer.cpp:722-724)")
            if tex is not None and uv is None:
                raise ValueError("a textured auxiliary mesh needs texture coordinates (instancer.cpp:730)")
            prim = np.arange(f.shape[0])
            vs.append(v); fs.append(f + base); ps.append(prim)
            ks.append((kind | np.where(prim == 1, 2, 0)).astype(np.uint8))            # bit 1: primID 1 of its own mesh (shadow filter, :553)
            ns.append(np.asarray(n, np.float32).reshape(-1, 3) if n is not None else np.zeros_like(v))
            us.append(np.asarray(uv, np.float32).reshape(-1, 2) if uv is not None else np.zeros((v.shape[0], 2), np.float32))
            ts.append(np.full(f.shape[0], len(sets) if tex is not None else -1, np.int32))
            if tex is not None:
                sets.append([np.ascontiguousarray(m, dtype=np.float32) for m in tex])
            base += v.shape[0]
        v, n, f, k, uv, prim, ft = (np.ascontiguousarray(np.concatenate(x)) for x in (vs, ns, fs, ks, us, ps, ts))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.lib.ntx_instancer_set_meshes(self._h, fp(v), fp(n), v.shape[0], f.ctypes.data_as(C.POINTER(C.c_int32)),
                                                     k.ctypes.data_as(C.POINTER(C.c_uint8)), f.shape[0]))
        if sets:                                      # three channel matrices per set; not exactly three channels: the first for all (:732)
            flat = [m for s_ in sets for m in (s_ if len(s_) == 3 else [s_[0]] * 3)]
            _lib.check(_lib.lib.ntx_instancer_set_mesh_textures(self._h, fp(uv), v.shape[0], ft.ctypes.data_as(C.POINTER(C.c_int32)), f.shape[0],
                                                                len(sets), _textures_struct(flat)))
        # as the library holds them (tests hand them to the oracle): kind bit 0 only, primIDs, uv, texture set per face, the sets
        self.meshes = (v, f, n, (k & 1).astype(np.uint8))
        self.mesh_prim, self.mesh_uv, self.mesh_tex, self.aux_textures = prim, uv, ft, sets

    def reserve(self, max_rays: int) -> None:
        _lib.check(_lib.lib.ntx_instancer_reserve(self._h, int(max_rays)))

    def matrices(self):
        """(world -> patch [K,4,4], direction maps [K,3,3], origins [K,3]) as the library holds them."""
        import numpy as np
        k = self.n_instances()
        w2p = np.zeros((k, 4, 4), np.float32); dirs = np.zeros((k, 3, 3), np.float32); org = np.zeros((k, 3), np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.lib.ntx_instancer_matrices(self._h, fp(w2p), fp(dirs), fp(org)))
        return w2p, dirs, org

    def export_transformations(self, file_path: str) -> None:
        """ExportTransformations (instancer.cpp:1040-1061): the patch -> world matrices as a JSON list of 4x4 lists."""
        with open(file_path, "w") as f:
            json.dump([[[float(v) for v in row] for row in m] for m in self._tr], f, indent=4)

    def get_model_input(self, rays_o, rays_d, parameters, n_samples: int, step_size: float, seed: Optional[int] = None,
                        ray_index=None, sparse: bool = False, fill: Optional[float] = None):
        """instancer.pyx:38-54: rays_o [n,3], rays_d [n,3], parameters [n,P] -> (rays_d_map [n,S,3], pts [n,S,3], t [n,S],
        dists [n,S], color [n,1,3], density [n,1], density_weight [n,S], instance_id [n,S] int32, idxs [k,1] = where(hit),
        params_map [n,S,P]) as torch tensors on the instancer's GPU (numpy or CPU inputs are uploaded).  `seed` fixes the call's
        draws (default: the constructor's seed and a call counter), `ray_index` = (index0, run_length, run_stride) of the rays
        in a larger image (include/nerftex.h: ntx_render_opts).  `sparse` (NTX_OPT_INSTANCER_SPARSE): the rows behind a ray's last step
        (dists == 0) stay unwritten in the six [n,S,...] buffers beside dists -- for `InstanceRenderer`, which reads rows with dists > 0
        only; `fill`: what the buffers hold before the call (tests)."""
        import numpy as np
        import torch
        dev = torch.device("cuda", self.device)
        to = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).to(device=dev, dtype=torch.float32).contiguous()
        rays_o, rays_d = to(rays_o).reshape(-1, 3), to(rays_d).reshape(-1, 3)
        n, S, P = rays_o.shape[0], int(n_samples), self.n_parameters
        par = to(parameters).reshape(n, -1) if P > 0 else None
        if P > 0 and par.shape[1] != P:
            raise ValueError(f"parameters must be [n,{P}] (the textures list has {P} entries' worth), got {tuple(parameters.shape)}")
        e = lambda *shape, dt=torch.float32: (torch.empty(shape, device=dev, dtype=dt) if fill is None else
                                              torch.full(shape, fill if dt == torch.float32 else (int(fill) if fill == fill else (255 if dt == torch.uint8 else -2 ** 31)), device=dev, dtype=dt))
        rays_d_map, pts, t, dists = e(n, S, 3), e(n, S, 3), e(n, S), e(n, S)
        color, density, weight = e(n, 1, 3), e(n, 1), e(n, S)
        instance_id, hit = e(n, S, dt=torch.int32), e(n, dt=torch.uint8)
        params_map = e(n, S, P)
        self._status = torch.zeros(1, device=dev, dtype=torch.int32)
        if seed is None:
            seed = (self.seed << 32) + self._calls
            self._calls += 1
        self.last_seed = int(seed)
        opts = _lib.render_opts(ray_index=ray_index, flags=_lib.OPT_INSTANCER_SPARSE if sparse else 0) if (ray_index is not None or sparse) else None
        ptr = lambda x: x.data_ptr() if x is not None and x.numel() else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_instancer_model_input(
                self._h, ptr(rays_o), ptr(rays_d), ptr(par), n, S, float(step_size), int(seed) & (2 ** 64 - 1), opts,
                ptr(rays_d_map), ptr(pts), ptr(t), ptr(dists), ptr(color), ptr(density), ptr(weight), ptr(instance_id), ptr(hit),
                ptr(params_map), self._status.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        self.last_hit = hit
        idxs = hit.nonzero(as_tuple=False)                                               # tf.where(hit), instancer.pyx:54
        return rays_d_map, pts, t, dists, color, density, weight, instance_id, idxs, params_map

    def status(self) -> int:
        """Flags of the last call (synchronises): 1 = a ray crossed more than 200 faces (MAX_TOTAL_HITS, instancer.cpp:22)."""
        st = getattr(self, "_status", None)
        return 0 if st is None else int(st.item())


def read_ply(path: str, normals: bool = False, uv: bool = False):
    """Vertices [nv,3] float32 and triangles [nf,3] int32 of a PLY file (`normals`: and the vertex normals nx / ny / nz [nv,3], for an
    auxiliary mesh; `uv`: and the texture coordinates s / t (or u / v, texture_u / texture_v) [nv,2], for DistributeInstancesOnMesh) (ascii or binary_little_endian; x/y/z of the vertex
    element, the list property of the face element; polygons are fanned) -- what igl::readPLY hands AddMesh /
    DistributeInstancesOnMesh for the culling mesh (instancer.cpp:241, 400)."""
    import numpy as np
    types = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
             "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
             "double": "f8", "float64": "f8"}
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                elements[-1][2].append(tuple(tok[1:]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: PLY format {fmt!r} is not supported")
        verts, faces, vnorm, vuv = None, [], None, None
        for name, count, props in elements:
            is_list = any(p[0] == "list" for p in props)
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                if name == "vertex":
                    cols = [p[-1] for p in props]
                    ix = [cols.index(c) for c in ("x", "y", "z")]
                    verts = np.asarray([[float(r[i]) for i in ix] for r in rows], np.float32).reshape(-1, 3)
                    if normals and all(c in cols for c in ("nx", "ny", "nz")):
                        vnorm = np.asarray([[float(r[cols.index(c)]) for c in ("nx", "ny", "nz")] for r in rows], np.float32).reshape(-1, 3)
                    for a_, b_ in _UV_NAMES:
                        if a_ in cols and b_ in cols:
                            vuv = np.asarray([[float(r[cols.index(a_)]), float(r[cols.index(b_)])] for r in rows], np.float32).reshape(-1, 2)
                            break
                elif name == "face":
                    for r in rows:
                        k = int(r[0]); idx = [int(v) for v in r[1:1 + k]]
                        faces += [[idx[0], idx[i], idx[i + 1]] for i in range(1, k - 1)]
            elif not is_list:
                dt = np.dtype([(p[-1], "<" + types[p[0]]) for p in props])
                data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                if name == "vertex":
                    verts = np.stack([data["x"], data["y"], data["z"]], -1).astype(np.float32)
                    if normals and all(c in data.dtype.names for c in ("nx", "ny", "nz")):
                        vnorm = np.stack([data["nx"], data["ny"], data["nz"]], -1).astype(np.float32)
                    for a_, b_ in _UV_NAMES:
                        if a_ in data.dtype.names and b_ in data.dtype.names:
                            vuv = np.stack([data[a_], data[b_]], -1).astype(np.float32)
                            break
            else:
                for _ in range(count):
                    idx = None
                    for p in props:
                        if p[0] == "list":
                            k = int(np.frombuffer(f.read(np.dtype(types[p[1]]).itemsize), "<" + types[p[1]])[0])
                            it = np.dtype("<" + types[p[2]])
                            vals = np.frombuffer(f.read(it.itemsize * k), it)
                            if idx is None and name == "face":
                                idx = [int(v) for v in vals]
                        else:
                            f.read(np.dtype(types[p[0]]).itemsize)
                    if idx is not None:
                        faces += [[idx[0], idx[i], idx[i + 1]] for i in range(1, len(idx) - 1)]
    if verts is None:
        raise ValueError(f"{path}: no vertex element")
    tri = np.asarray(faces, np.int32).reshape(-1, 3)
    if normals and vnorm is None:
        raise ValueError(f"{path}: no vertex normals (nx, ny, nz)")
    if uv and vuv is None:
        raise ValueError(f"{path}: no texture coordinates (s, t)")
    return (verts, tri) + ((vnorm,) if normals else ()) + ((vuv,) if uv else ())


_UV_NAMES = (("s", "t"), ("u", "v"), ("texture_u", "texture_v"))


def closest_point_triangle(p, a, b, c):
    """closest_point_triangle (instancer.cpp:154-198; Ericson's regions): the closest point of triangle abc to p and its
    barycentrics, float32."""
    import numpy as np
    F = np.float32
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = F(ab @ ap), F(ac @ ap)
    if d1 <= 0 and d2 <= 0:
        return a, np.asarray([1, 0, 0], F)
    bp = p - b
    d3, d4 = F(ab @ bp), F(ac @ bp)
    if d3 >= 0 and d4 <= d3:
        return b, np.asarray([0, 1, 0], F)
    cp = p - c
    d5, d6 = F(ab @ cp), F(ac @ cp)
    if d6 >= 0 and d5 <= d6:
        return c, np.asarray([0, 0, 1], F)
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0:
        v = d1 / (d1 - d3)
        return a + v * ab, np.asarray([1 - v, v, 0], F)
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0:
        v = d2 / (d2 - d6)
        return a + v * ac, np.asarray([1 - v, 0, v], F)
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
        v = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        return b + v * (c - b), np.asarray([0, 1 - v, v], F)
    denom = F(1.0) / (va + vb + vc)
    v, w = vb * denom, vc * denom
    return a + v * ab + w * ac, np.asarray([1 - v - w, v, w], F)


def distribute_instances_on_mesh(vertices, faces, normals, uv, scale: float, patch_origins=None, jitter_amount: float = 0.0, seed: int = 0):
    """DistributeInstancesOnMesh (instancer.cpp:233-390) on the host: the patch -> world transformations [K,4,4] it hands AddInstance,
    and the patch scale (`scale <= 0`: the mesh's average edge length, :244-245).  Per vertex a tangent frame from the texture
    coordinates (:249-276: the u direction, made orthogonal to the vertex normal; bitangent = n x t); per patch origin the frame of
    the closest point of the mesh within one average edge length (:300-322), or, without origins, one patch per distinct vertex
    (:343-366); columns (tangent, bitangent, normal) * scale, translation = the origin.  `jitter_amount` > 0 turns the frame about its
    normal by jitter_amount * U(0, pi) (:326-329, 351-355), the draws taken like the reference's: std::mt19937(seed) through
    std::uniform_real_distribution<float> -- numpy's legacy MT19937 seeding gives the same words, and libstdc++'s
    generate_canonical<float, 24> is float(word) / 2^32."""
    import numpy as np
    F = np.float32
    V = np.asarray(vertices, F).reshape(-1, 3); Fa = np.asarray(faces, np.int64).reshape(-1, 3)
    N = np.asarray(normals, F).reshape(-1, 3).copy(); UV = np.asarray(uv, F).reshape(-1, 2)
    edges = np.concatenate([np.linalg.norm(V[Fa[:, j]] - V[Fa[:, (j + 1) % 3]], axis=1).astype(F) for j in range(3)])
    avg_edge = float(edges.astype(np.float64).sum() / (3 * Fa.shape[0]))                     # igl::avg_edge_length
    scale = float(scale) if scale > 0 else avg_edge
    e0, e1 = V[Fa[:, 1]] - V[Fa[:, 0]], V[Fa[:, 2]] - V[Fa[:, 0]]
    uv0, uv1 = UV[Fa[:, 1]] - UV[Fa[:, 0]], UV[Fa[:, 2]] - UV[Fa[:, 0]]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = F(1.0) / (uv0[:, 0] * uv1[:, 1] - uv0[:, 1] * uv1[:, 0])
        t_face = ((e0 * uv1[:, 1:2] - e1 * uv0[:, 1:2]) * r[:, None]).astype(F)
    T = np.zeros_like(V)
    for j in range(3):
        np.add.at(T, Fa[:, j], t_face)                                                        # :258-264 (summation order: by face)
    def unit(x):                                                                              # Eigen's normalize(): only when the squared norm is positive
        n = np.linalg.norm(x, axis=-1, keepdims=True)
        return (x / np.where(n > 0, n, 1)).astype(F)
    N = unit(N)
    T = unit(T - N * np.sum(N * T, -1, keepdims=True))
    B = np.cross(N, T).astype(F)
    words = np.random.RandomState(int(seed))._bit_generator.random_raw
    pi_f = F(np.pi)

    def turn(b, n):                                                                           # Rodrigues about n, :328 / :353
        c = F(words(1)[0]) / F(4294967296.0)
        c = np.nextafter(F(1), F(0)) if c >= 1 else c
        angle = F(jitter_amount) * (pi_f * c)
        return (b * np.cos(angle) + np.cross(n, b) * np.sin(angle) + n * (n @ b) * (F(1) - np.cos(angle))).astype(F)

    out = []
    if patch_origins is not None:
        origins = np.asarray(patch_origins, F).reshape(-1, 3)
        from scipy.spatial import cKDTree
        cen = V[Fa].mean(1); reach = np.linalg.norm(V[Fa] - cen[:, None], axis=-1).max()
        tree = cKDTree(cen)
        for pt in origins:
            best = None
            for f in sorted(tree.query_ball_point(pt, avg_edge + reach + 1e-6)):               # (the triangles that can hold a point that close)
                q, w = closest_point_triangle(pt, V[Fa[f, 0]], V[Fa[f, 1]], V[Fa[f, 2]])
                d = F(np.linalg.norm(pt - q))
                if d < (best[0] if best else F(avg_edge)):                                     # radius = avg_edge_length, shrinking (:219-224)
                    best = (d, f, w)
            if best is None:
                raise ValueError(f"patch origin {pt.tolist()} lies further than one average edge length ({avg_edge:.4g}) from the mesh")
            _, f, w = best
            mix = lambda A: (A[Fa[f, 0]] * w[0] + A[Fa[f, 1]] * w[1] + A[Fa[f, 2]] * w[2]).astype(F)
            n, t = unit(mix(N)), unit(mix(T))
            b = np.cross(n, t).astype(F)
            if jitter_amount > 0:
                b = turn(b, n)
            t = np.cross(b, n).astype(F)
            m = np.eye(4, dtype=F)
            m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = t * F(scale), b * F(scale), n * F(scale), pt
            out.append(m)
    else:
        seen = set()
        for i in range(V.shape[0]):
            key = V[i].tobytes()
            if key in seen:
                continue
            seen.add(key)
            t, b, n = T[i], B[i], N[i]
            if jitter_amount > 0:
                b = turn(b, n)
                tc = np.cross(n, b).astype(F)
                t = (F(-1.0) if (t @ tc) < 0 else F(1.0)) * tc
            m = np.eye(4, dtype=F)
            m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = t * F(scale), b * F(scale), n * F(scale), V[i]
            out.append(m)
    return np.stack(out) if out else np.zeros((0, 4, 4), F), scale
