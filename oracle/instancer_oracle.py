"""CPU restatement of the reference's patch instancer, `C_Instancer::GetModelInput` (instancer/src/instancer.cpp:751-1037)
behind `Instancer.get_model_input` (instancer/instancer.pyx:38-54).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in nerf_tex_amd/); the product is the HIP kernel
`ntx_inst::inst_hits_kernel` / `inst_march_kernel` (nerf_tex_amd/csrc/ntx_instancer.hip) behind `ntx_instancer_model_input`.

PARITY UNPINNED.  The reference's instancer is C++ on Embree 3 (+ Eigen, libigl): none of them is in this image, its four
submodules are empty and its meshes are LFS pointers, so nothing of it can be built or run here and it ships no test vectors.
What is restated line by line is the reference's OWN code: the hit ordering (instancer.cpp:441-452), the segment sums
(:800-826), the step count / dists / offset (:840-859), the marching loop (:870-1010), nearest / random / blended patch choice
(:670-713), the point / direction / light maps (:556-589) and the closing sample (:1013-1027).  What Embree does underneath
(`rtcIntersect1` on instanced quads with an all-hits filter, :779) is restated from its published contract: every face of an
instanced box the ray crosses with tnear < t <= tfar is reported once, at the ray parameter of the WORLD ray; here that is the
slab test of the ray taken into patch coordinates (entry and exit parameter), evaluated in float32.  Rays through an edge or a
corner (two faces at one t), rays starting on a face, and more than MAX_TOTAL_HITS = 200 face crossings (Embree's traversal
order decides which are dropped, :539) are outside what this restatement can vouch for.

Random numbers: the reference draws from one std::mt19937 in ray order (:855, :675, :710), which a kernel that renders rays in
parallel cannot reproduce; the product keys a Philox4x32-10 block by (seed, global ray index, sample) instead and THIS file
restates those draws (`offset_uniforms`, `choice_uniforms`), like `nerftex_oracle.jitter_uniforms` does for the jitter.

Image textures (round 4): `load_texture` (:34-50), `interpolate2d` (:605-637), `get_parameters` (:640-667: the closest point of the
instancer mesh within patch_max_extent, `closest_point_triangle` :154-198, brute force over the triangles where the reference
walks Embree's BVH), the per-segment texture samples with their LINEAR interpolation (:910-927, 989-998) and the albedo of an
auxiliary mesh (:725-733).  Reference quirks kept: a texture file of c channels takes c parameters but only ONE of its channel
matrices is ever multiplied in -- `textures[i]` for the i-th FILE, where `textures` is the list of all channel matrices (:656-662);
an interpolated row is `s0 * (1 - w) + s1 * w` in EVERY column, also those no texture touches (:923).  Where the reference reads
outside a texture (u or v = 1 makes idx + 1 = rows, :612-613; its assert is compiled out) the index is clamped here: the weight of
such a texel is 0.  Which of several triangles at the same distance Embree reports first is not knowable; here the lowest primID.
Eigen's fixed-size reductions (dot, squaredNorm of a Vector3f, a coefficient of a 3x3 * 3 product) pair their terms as x0 + (x1 + x2)
(Redux.h's unroller): every expression Eigen evaluates in the reference (getPt, getDir, normalized, norm, dot, squaredNorm) is spelled
that way; what Embree does on its side (the ray taken into an instance, the triangle tests) sums left to right.

Everything is sequential Python over numpy float32 scalars: small cases only.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import nerftex_oracle as orc

F32 = np.float32
INVALID = 0xFFFFFFFF          # RTC_INVALID_GEOMETRY_ID: the instID of a hit on a mesh (instancer.cpp:444)
MAX_TOTAL_HITS = 200          # instancer.cpp:22
T_NEAR, T_FAR = F32(0.0), F32(100.0)      # init_ray(..., 0, 100, ...)  instancer.cpp:776
SAMPLE_RANDOM, SAMPLE_NEAREST, SAMPLE_NEAREST_BLEND = 0, 1, 2        # instancer.pyx:14


@dataclass
class InstancerSpec:
    """What the constructor + AddInstance leave behind (instancer.cpp:53-93, 124-141)."""
    b_0: np.ndarray                      # patch box, patch coordinates
    b_1: np.ndarray
    inv: np.ndarray                      # [K,4,4] world -> patch  (this->transformations)
    dir_t: np.ndarray                    # [K,3,3] (this->dir_transformations)
    origins: np.ndarray                  # [K,3]   (this->instance_origins)
    n_parameters: int = 0
    light_dir_idx: int = -1
    light_strength_idx: int = -1
    sample_method: int = SAMPLE_RANDOM
    use_mean_distance: bool = False
    patch_scale: float = 1.0             # only DistributeInstancesOnMesh sets it (instancer.cpp:236); 1 otherwise (:53)
    mesh_v: Optional[np.ndarray] = None  # the instancer mesh (culls, closes a ray with an opaque black sample)
    mesh_f: Optional[np.ndarray] = None
    mesh_n: Optional[np.ndarray] = None  # vertex normals and a kind per face (0 = instancer mesh, 1 = auxiliary mesh: shaded, :716-743)
    mesh_kind: Optional[np.ndarray] = None
    cast_shadow_rays: bool = False       # instancer.cpp:53
    min_shadow_samples: int = 4
    n_shadow_samples: int = 512
    mesh_prim: Optional[np.ndarray] = None      # per face: its primID inside its own mesh (the shadow filter's `primID == 1`, :553)
    mesh_uv: Optional[np.ndarray] = None        # per vertex of the mesh list: texture coordinates (auxiliary albedo, :730)
    mesh_tex: Optional[np.ndarray] = None       # per face: index into aux_textures, -1 = none (albedo 0.8, :728)
    aux_textures: Optional[list] = None         # per auxiliary mesh with a texture: its channel matrices (loadTexture)
    # parameter textures on the instancer mesh (DistributeInstancesOnMesh was called: instancer_geomID is valid, :911)
    tex_idx: Optional[list] = None              # texture_parameter_idxs: first parameter of every texture FILE (:87)
    textures: Optional[list] = None             # all channel matrices of all files, in order (:86)
    inst_v: Optional[np.ndarray] = None         # the instancer mesh: vertices, faces, texture coordinates (:236)
    inst_f: Optional[np.ndarray] = None
    inst_uv: Optional[np.ndarray] = None
    patch_max_extent: float = 0.0               # :69, scaled by :246
    min_texture_samples: int = 4
    n_texture_samples: int = 512


def load_texture(path) -> List[np.ndarray]:
    """loadTexture (instancer.cpp:34-50): the image's channels as matrices [width, height] of value / 255, indexed (x, y counted from
    the BOTTOM row): stb hands rows top-down as [height * width, channels]; a channel's column is mapped column-major to
    (width, height) -- element (x, y) = pixel (row y, column x) -- and `.rowwise().reverse()` turns the y axis over.  Decoded with PIL
    (the product has its own decoder: nerf_tex_amd/png.py); stb's channel count: grey 1, grey + alpha 2, RGB 3, RGBA 4, a palette
    expanded to RGB(A), 16-bit samples reduced to their high byte."""
    from PIL import Image
    im = Image.open(path)
    if im.mode == "P":
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    elif im.mode == "1":
        im = im.convert("L")
    elif im.mode.startswith("I"):
        im = Image.fromarray((np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8))
    a = np.asarray(im)
    if a.ndim == 2:
        a = a[:, :, None]
    return texture_from_pixels(a)


def texture_from_pixels(pixels) -> List[np.ndarray]:
    """The matrices loadTexture builds from stb's pixels [height, width, channels] uint8 (instancer.cpp:40-46)."""
    a = np.asarray(pixels, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    f = a.astype(F32) / F32(255.0)                                                       # cast<float>() / 255.f
    return [np.ascontiguousarray(f[::-1, :, c].T) for c in range(a.shape[2])]           # [x, y from the bottom]


def parse_textures(textures: Sequence[str], images=None):
    """(n_parameters, light_dir_parameter_idx, light_strength_parameter_idx, texture_parameter_idxs, channel matrices) from the
    constructor's `textures` list (instancer.cpp:74-92).  `images`: {path: channel matrices} for paths that are not files."""
    n, ld, ls = 0, -1, -1
    idx, mats = [], []
    for path in textures:
        if path == "light":
            ld = n; n += 3
        elif path == "point":
            ls = n; ld = n + 1; n += 4
        elif path != "":
            tex = images[path] if images is not None and path in images else load_texture(path)
            mats += list(tex)
            idx.append(n)
            n += len(tex)
        else:
            n += 1
    return n, ld, ls, idx, mats


def prepare_instances(transformations) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """AddInstance (instancer.cpp:124-141) for a list of patch -> world matrices [K,4,4]: (world -> patch, direction maps,
    origins).  The inverse is taken in float64 and rounded (Eigen's float 4x4 inverse differs in the last place)."""
    tr = np.asarray(transformations, dtype=F32).reshape(-1, 4, 4)
    inv = np.linalg.inv(tr.astype(np.float64)).astype(F32)
    cols = np.transpose(tr[:, :3, :3], (0, 2, 1)).astype(np.float64)           # rows of R^T
    dir_t = (cols / np.linalg.norm(cols, axis=-1, keepdims=True)).astype(F32)   # .rowwise().normalized()
    return inv, dir_t, tr[:, :3, 3].copy()


def make_spec(b_0, b_1, transformations, textures=(), instance_sampling_method="random", use_mean_distance=False,
              mesh=None, matrices=None, cast_shadow_rays=False, min_shadow_samples=4, n_shadow_samples=512,
              images=None, instancer_mesh=None, patch_scale=1.0, min_texture_samples=4, n_texture_samples=512,
              mesh_uv=None, mesh_tex=None, aux_textures=None, mesh_prim=None) -> InstancerSpec:
    """`mesh`: the list of culling meshes as (vertices, faces[, normals, kind per face]); `mesh_prim` the faces' primIDs inside their
    own meshes (default: one mesh per run of equal kind... i.e. the face index where only one mesh is given).
    `instancer_mesh` = (vertices, faces, uv): DistributeInstancesOnMesh was called with it (parameter textures apply, :911), and
    `patch_scale` is its scale (patch_max_extent *= scale, :246)."""
    n, ld, ls, tex_idx, tex = parse_textures(textures, images)
    inv, dir_t, org = prepare_instances(transformations) if matrices is None else matrices
    mv = mf = mn = mk = None
    if mesh is not None:
        mv = np.asarray(mesh[0], F32).reshape(-1, 3); mf = np.asarray(mesh[1], np.int32).reshape(-1, 3)
        if len(mesh) > 2 and mesh[2] is not None:                                       # (vertices, faces, normals, kind per face)
            mn = np.asarray(mesh[2], F32).reshape(-1, 3); mk = np.asarray(mesh[3], np.uint8).reshape(-1)
    spec = InstancerSpec(np.asarray(b_0, F32), np.asarray(b_1, F32), inv, dir_t, org, n, ld, ls,
                         {"random": 0, "nearest": 1, "nearest_blend": 2}[instance_sampling_method], bool(use_mean_distance),
                         1.0, mv, mf, mn, mk, bool(cast_shadow_rays), int(min_shadow_samples), int(n_shadow_samples))
    if mf is not None:
        spec.mesh_prim = np.arange(mf.shape[0], dtype=np.int64) if mesh_prim is None else np.asarray(mesh_prim, np.int64).reshape(-1)
    spec.mesh_uv = None if mesh_uv is None else np.asarray(mesh_uv, F32).reshape(-1, 2)
    spec.mesh_tex = None if mesh_tex is None else np.asarray(mesh_tex, np.int64).reshape(-1)
    spec.aux_textures = aux_textures
    spec.tex_idx, spec.textures = tex_idx, tex
    spec.min_texture_samples, spec.n_texture_samples = int(min_texture_samples), int(n_texture_samples)
    # patch_max_extent (:69): the norm of the elementwise maximum of b_0 and b_1; Eigen pairs x0^2 + (x1^2 + x2^2)
    e = np.maximum(spec.b_0, spec.b_1).astype(F32)
    spec.patch_max_extent = np.sqrt(e[0] * e[0] + (e[1] * e[1] + e[2] * e[2]))
    if instancer_mesh is not None:
        spec.inst_v = np.asarray(instancer_mesh[0], F32).reshape(-1, 3); spec.inst_f = np.asarray(instancer_mesh[1], np.int64).reshape(-1, 3)
        spec.inst_uv = np.asarray(instancer_mesh[2], F32).reshape(-1, 2)
        spec.patch_scale = float(patch_scale)
        spec.patch_max_extent = F32(spec.patch_max_extent * F32(patch_scale))           # :246
    return spec


# ------------------------------------------------------------------------------------------------------------------------------
# the product's random draws, restated
# ------------------------------------------------------------------------------------------------------------------------------

def offset_uniforms(n_rays: int, seed: int, ray_index=None) -> np.ndarray:
    """One U[0,1) per ray for t_offset (instancer.cpp:848, 855): Philox counter (0, ray lo, ray hi, 2), key = seed."""
    ray = orc.global_ray_index(n_rays, ray_index).astype(np.uint64)
    bits = orc.philox4x32_10(np.uint32(0), (ray & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ray >> np.uint64(32)).astype(np.uint32),
                             np.uint32(2), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return orc.uniform01_from_bits(bits)


def choice_uniforms(n_rays: int, n_pts: int, seed: int, ray_index=None) -> np.ndarray:
    """One U[0,1) per (ray, step) for the random / blended patch choice (instancer.cpp:675, 710): counter (step, ray lo,
    ray hi, 3)."""
    ray = orc.global_ray_index(n_rays, ray_index).astype(np.uint64)[:, None]
    i = np.arange(n_pts, dtype=np.uint32)[None, :]
    bits = orc.philox4x32_10(i, (ray & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ray >> np.uint64(32)).astype(np.uint32),
                             np.uint32(3), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return orc.uniform01_from_bits(bits)


# ------------------------------------------------------------------------------------------------------------------------------
# geometry
# ------------------------------------------------------------------------------------------------------------------------------

def _affine(m, p):
    """A point taken into patch coordinates the way the ray is on Embree's side (instance traversal; Embree's own arithmetic is not
    knowable here): products summed left to right, then the translation."""
    return np.asarray([((m[r, 0] * p[0] + m[r, 1] * p[1]) + m[r, 2] * p[2]) + m[r, 3] for r in range(3)], F32)


def _linear(m, p):
    return np.asarray([(m[r, 0] * p[0] + m[r, 1] * p[1]) + m[r, 2] * p[2] for r in range(3)], F32)


def _affine_e(m, p):
    """block<3,3>(0,0) * p + block<3,1>(0,3) as EIGEN evaluates it (getPt, instancer.cpp:556-558): a coefficient of the 3x3 * 3 product
    is the unrolled reduction x0 + (x1 + x2) (Redux.h: redux_novec_unroller splits 3 terms as 1 + 2), then the translation."""
    return np.asarray([(m[r, 0] * p[0] + (m[r, 1] * p[1] + m[r, 2] * p[2])) + m[r, 3] for r in range(3)], F32)


def _linear_e(m, p):
    """Matrix3f * Vector3f on Eigen's side (getDir, :561-563)."""
    return np.asarray([m[r, 0] * p[0] + (m[r, 1] * p[1] + m[r, 2] * p[2]) for r in range(3)], F32)


def _normalized(v):
    """Eigen's normalized(): v / sqrt(squaredNorm) when the squared norm is positive; squaredNorm = x0^2 + (x1^2 + x2^2)."""
    n2 = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2])
    return (v / np.sqrt(n2)).astype(F32) if n2 > 0 else v


def _norm(v):
    return np.sqrt(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]))


def box_hits(spec: InstancerSpec, o, d) -> List[Tuple[np.float32, int]]:
    """Face crossings of the world ray o + t d with every instanced box, as (t, instID): the ray goes into patch coordinates
    (Embree's instance traversal), the slab test gives the entry and the exit parameter, each is a hit when tnear < t <= tfar."""
    hits = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for k in range(spec.inv.shape[0]):
            ol = _affine(spec.inv[k], o); dl = _linear(spec.inv[k], d)
            t_in, t_out = F32(-np.inf), F32(np.inf)
            miss = False
            for a in range(3):
                if dl[a] == 0:
                    if ol[a] < spec.b_0[a] or ol[a] > spec.b_1[a]:
                        miss = True
                    continue
                inv_d = F32(1.0) / dl[a]
                t0 = (spec.b_0[a] - ol[a]) * inv_d; t1 = (spec.b_1[a] - ol[a]) * inv_d
                lo, hi = (t0, t1) if t0 < t1 else (t1, t0)
                t_in = lo if lo > t_in else t_in
                t_out = hi if hi < t_out else t_out
            if miss or not (t_in < t_out):
                continue
            for tt in (t_in, t_out):
                if T_NEAR < tt <= T_FAR:
                    hits.append((F32(tt), k))
    return hits


def mesh_hit(spec: InstancerSpec, o, d, full: bool = False):
    """Closest crossing of the meshes (Moeller-Trumbore in float32, no culling, tnear < t <= tfar) or None; ties go to the lower
    triangle.  `full`: (t, triangle, u, v) -- Embree's hit of :1020."""
    if spec.mesh_v is None:
        return None
    best = None
    cross = lambda a, b: np.asarray([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F32)
    dot = lambda a, b: (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
    with np.errstate(divide="ignore", invalid="ignore"):
        for k, f in enumerate(spec.mesh_f):
            v0, v1, v2 = spec.mesh_v[f[0]], spec.mesh_v[f[1]], spec.mesh_v[f[2]]
            e1 = v1 - v0; e2 = v2 - v0
            p = cross(d, e2); det = dot(e1, p)
            if det == 0:
                continue
            inv_det = F32(1.0) / det
            s = o - v0
            u = dot(s, p) * inv_det
            if u < 0 or u > 1:
                continue
            q = cross(s, e1)
            v = dot(d, q) * inv_det
            if v < 0 or u + v > 1:
                continue
            tt = dot(e2, q) * inv_det
            if T_NEAR < tt <= T_FAR and (best is None or tt < best[0]):
                best = (F32(tt), k, u, v)
    if best is None:
        return None
    return best if full else best[0]


def _dot3(a, b):
    """Eigen's dot of two Vector3f: the unrolled reduction pairs x0 + (x1 + x2) (Redux.h, redux_novec_unroller)."""
    return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2])


def closest_point_triangle(p, a, b, c):
    """closest_point_triangle (instancer.cpp:154-198): the closest point of triangle abc to p and its barycentrics, float32."""
    one, zero = F32(1.0), F32(0.0)
    ab = b - a; ac = c - a; ap = p - a
    d1 = _dot3(ab, ap); d2 = _dot3(ac, ap)
    if d1 <= 0 and d2 <= 0:
        return a, np.asarray([1, 0, 0], F32)
    bp = p - b
    d3 = _dot3(ab, bp); d4 = _dot3(ac, bp)
    if d3 >= 0 and d4 <= d3:
        return b, np.asarray([0, 1, 0], F32)
    cp = p - c
    d5 = _dot3(ab, cp); d6 = _dot3(ac, cp)
    if d6 >= 0 and d5 <= d6:
        return c, np.asarray([0, 0, 1], F32)
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0:
        v = d1 / (d1 - d3)
        return (a + v * ab).astype(F32), np.asarray([one - v, v, zero], F32)
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0:
        v = d2 / (d2 - d6)
        return (a + v * ac).astype(F32), np.asarray([one - v, zero, v], F32)
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
        v = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        return (b + v * (c - b)).astype(F32), np.asarray([zero, one - v, v], F32)
    denom = one / ((va + vb) + vc)
    v = vb * denom; w = vc * denom
    return ((a + v * ab) + w * ac).astype(F32), np.asarray([(one - v) - w, v, w], F32)


def closest_points_all(verts, faces, q):
    """closest_point_triangle for every triangle at once (numpy float32, operation for operation the scalar function above: its
    branches become masks taken in the same order): distances [F] and barycentrics [F,3]."""
    one, zero = F32(1.0), F32(0.0)
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    dot = lambda x, y: x[:, 0] * y[:, 0] + (x[:, 1] * y[:, 1] + x[:, 2] * y[:, 2])
    ab = b - a; ac = c - a; ap = q - a; bp = q - b; cp = q - c; bc = c - b
    d1 = dot(ab, ap); d2 = dot(ac, ap); d3 = dot(ab, bp); d4 = dot(ac, bp); d5 = dot(ab, cp); d6 = dot(ac, cp)
    vc = d1 * d4 - d3 * d2; vb = d5 * d2 - d1 * d6; va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        v_ab = d1 / (d1 - d3); v_ac = d2 / (d2 - d6); v_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        denom = one / ((va + vb) + vc)
    v_in = vb * denom; w_in = vc * denom
    conds = [(d1 <= 0) & (d2 <= 0), (d3 >= 0) & (d4 <= d3), (d6 >= 0) & (d5 <= d6), (vc <= 0) & (d1 >= 0) & (d3 <= 0),
             (vb <= 0) & (d2 >= 0) & (d6 <= 0), (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)]
    n = faces.shape[0]
    z = np.zeros(n, F32); o1 = np.ones(n, F32)
    pts = [a, b, c, a + v_ab[:, None] * ab, a + v_ac[:, None] * ac, b + v_bc[:, None] * bc, (a + v_in[:, None] * ab) + w_in[:, None] * ac]
    uvws = [np.stack([o1, z, z], -1), np.stack([z, o1, z], -1), np.stack([z, z, o1], -1), np.stack([one - v_ab, v_ab, z], -1),
            np.stack([one - v_ac, z, v_ac], -1), np.stack([z, one - v_bc, v_bc], -1), np.stack([(one - v_in) - w_in, v_in, w_in], -1)]
    region = np.full(n, 6)
    for k in range(5, -1, -1):
        region = np.where(conds[k], k, region)
    pt = np.choose(region[:, None], [p.astype(F32) for p in pts]); uvw = np.choose(region[:, None], [u.astype(F32) for u in uvws])
    e = q - pt
    with np.errstate(invalid="ignore"):
        d = np.sqrt(e[:, 0] * e[:, 0] + (e[:, 1] * e[:, 1] + e[:, 2] * e[:, 2]))
    return d.astype(F32), uvw


def closest_point_on_mesh(verts, faces, q, radius):
    """rtcPointQuery with closest_point_query_function (instancer.cpp:200-230, 644-654): the triangle whose closest point lies nearest
    to q, strictly within `radius` (the callback shrinks the radius: d < radius); (primID, barycentrics) or (None, None).  Embree
    calls back in the order of its BVH walk; here the triangles come in ascending order, so that of several at one distance the
    lowest primID stays."""
    d, uvw = closest_points_all(verts, faces, np.asarray(q, F32))
    d = np.where(np.isnan(d), F32(np.inf), d)
    k = int(np.argmin(d))                                                                # the first of the smallest
    if not d[k] < F32(radius):
        return None, None
    return k, uvw[k]


def interpolate2d(x, y_ref):
    """interpolate2d (instancer.cpp:605-625) of one channel matrix at x in [0,1]^2: bilinear between the four texels around
    x * (rows - 1, cols - 1); indices by truncation (cast<int>), weights x - floor(x).  Indices outside the matrix are clamped (the
    reference reads past it there; for u or v = 1 the weight of that texel is 0)."""
    rows, cols = y_ref.shape
    x0 = F32(x[0]) * (F32(rows) - F32(1.0)); x1 = F32(x[1]) * (F32(cols) - F32(1.0))
    i = int(np.trunc(x0)); j = int(np.trunc(x1))
    w0 = x0 - np.floor(x0); w1 = x1 - np.floor(x1)
    cl = lambda v, n: min(max(v, 0), n - 1)
    y00 = y_ref[cl(i, rows), cl(j, cols)]; y01 = y_ref[cl(i, rows), cl(j + 1, cols)]
    y10 = y_ref[cl(i + 1, rows), cl(j, cols)]; y11 = y_ref[cl(i + 1, rows), cl(j + 1, cols)]
    one = F32(1.0)
    return F32(((y00 * (one - w0) * (one - w1) + y01 * (one - w0) * w1) + y10 * w0 * (one - w1)) + y11 * w0 * w1)


def _bary_mix(rows, f, uvw):
    """A.row(f0) * uvw(0) + A.row(f1) * uvw(1) + A.row(f2) * uvw(2), elementwise, left to right (:661, 722, 730)."""
    return ((rows[f[0]] * uvw[0] + rows[f[1]] * uvw[1]) + rows[f[2]] * uvw[2]).astype(F32)


def get_parameters(spec: InstancerSpec, pt, parameters):
    """getParameters (instancer.cpp:640-667): the parameter row at world point pt -- every texture FILE i multiplies parameter
    texture_parameter_idxs[i] by the interpolated value of channel matrix textures[i] (sic) at the texture coordinates of the closest
    point of the instancer mesh within patch_max_extent; nothing within reach: the row as given."""
    out = np.asarray(parameters, F32).copy()
    f, uvw = closest_point_on_mesh(spec.inst_v, spec.inst_f, np.asarray(pt, F32), spec.patch_max_extent)
    if f is not None:
        uv = _bary_mix(spec.inst_uv, spec.inst_f[f], uvw)
        for i, idx in enumerate(spec.tex_idx):
            out[idx] = out[idx] * interpolate2d(uv, spec.textures[i])
    return out


def shade_mesh(spec: InstancerSpec, o, d, hit, light):
    """shadeMesh (instancer.cpp:716-743) for a hit (t, triangle, u, v) on an auxiliary mesh: albedo 0.8 or its texture at the hit's
    texture coordinates (one value for all three channels unless the image has exactly three), the interpolated vertex normal, diffuse = max(n . l, 0) unless the point just above the surface is shadowed, 0.2 ambient."""
    tt, k, u, v = hit
    f = spec.mesh_f[k]
    w0 = (F32(1.0) - u) - v                                                              # Vector3f(1 - u - v, u, v), :1020
    n = ((spec.mesh_n[f[0]] * w0 + spec.mesh_n[f[1]] * u) + spec.mesh_n[f[2]] * v).astype(F32)
    n = _normalized(n)
    pt = ((o + tt * d).astype(F32) + n * F32(1e-6)).astype(F32)                         # :736
    diffuse = F32(1.0)
    if not is_shadowed(spec, pt, light):
        nl = _normalized(np.asarray(light, F32).copy())
        nd = n[0] * nl[0] + (n[1] * nl[1] + n[2] * nl[2])                                # n.dot(dir.normalized())
        diffuse = diffuse * (nd if nd > 0 else F32(0.0))
    else:
        diffuse = F32(0.0)
    sm = diffuse + F32(0.2)
    shade = sm if sm < 1 else F32(1.0)
    albedo = np.full(3, F32(0.8), F32)                                                   # :728
    if spec.mesh_tex is not None and spec.mesh_tex[k] >= 0:                              # :730-732
        tex = spec.aux_textures[int(spec.mesh_tex[k])]
        uv = _bary_mix(spec.mesh_uv, f, np.asarray([w0, u, v], F32))
        val = [interpolate2d(uv, m) for m in tex]
        albedo = np.asarray(val, F32) if len(val) == 3 else np.full(3, val[0], F32)
    return (albedo * shade).astype(F32)


def is_shadowed(spec: InstancerSpec, pt, direction) -> bool:
    """isShadowed (instancer.cpp:591-602): an occlusion query from `pt` along `direction` (as given, not normalised; 0 < t <= 100)
    whose filter (:543-554) accepts a hit on the TOP face of a patch box from outside (primID 4 = the z = b_1 quad of createAABB,
    :113; dot(dir, Ng) < 0), any hit on the BOTTOM face (primID 1 = the z = b_0 quad, :110), a hit on a mesh from its front
    (Ng = cross(v1 - v0, v2 - v0)) or on the triangle with primID 1 of any mesh from either side (the filter's last clause does not look
    at the geometry); side faces are ignored.  Ray and normal are taken in patch coordinates, where Embree's
    instance traversal calls the filter: from outside through the top = the patch-space direction points down."""
    pt = np.asarray(pt, F32); direction = np.asarray(direction, F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        for k in range(spec.inv.shape[0]):
            ol = _affine(spec.inv[k], pt); dl = _linear(spec.inv[k], direction)
            if dl[2] == 0:
                continue
            inv_d = F32(1.0) / dl[2]
            for z, top in ((spec.b_1[2], True), (spec.b_0[2], False)):
                tt = (z - ol[2]) * inv_d
                if not (T_NEAR < tt <= T_FAR):
                    continue
                x = ol[0] + tt * dl[0]; y = ol[1] + tt * dl[1]
                if spec.b_0[0] <= x <= spec.b_1[0] and spec.b_0[1] <= y <= spec.b_1[1] and (not top or dl[2] < 0):
                    return True
        if spec.mesh_v is not None:
            cross = lambda a, b: np.asarray([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F32)
            dot = lambda a, b: (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
            o, d = pt, direction
            for fi, f in enumerate(spec.mesh_f):
                v0, v1, v2 = spec.mesh_v[f[0]], spec.mesh_v[f[1]], spec.mesh_v[f[2]]
                e1 = v1 - v0; e2 = v2 - v0
                p = cross(d, e2); det = dot(e1, p)
                if det == 0:
                    continue
                inv_det = F32(1.0) / det
                sv = o - v0
                u = dot(sv, p) * inv_det
                if u < 0 or u > 1:
                    continue
                q = cross(sv, e1)
                v = dot(d, q) * inv_det
                if v < 0 or u + v > 1:
                    continue
                tt = dot(e2, q) * inv_det
                # the filter's `primID == 1` holds for ANY geometry (:553): the second triangle of every mesh occludes from both sides
                if T_NEAR < tt <= T_FAR and (dot(d, cross(e1, e2)) < 0 or spec.mesh_prim[fi] == 1):
                    return True
    return False


def get_mean_distance(mu, hw):
    """instancer.cpp:746-748; std::pow(float, int) promotes to double, the result is returned as float."""
    mu = float(mu); hw = float(hw)
    return F32(mu + 2 * mu * hw ** 2 / (3 * mu ** 2 + hw ** 2))


# ------------------------------------------------------------------------------------------------------------------------------
# GetModelInput
# ------------------------------------------------------------------------------------------------------------------------------

def get_model_input(spec: InstancerSpec, rays_o, rays_d, parameters, n_samples: int, step_size: float,
                    u_offset: np.ndarray, u_choice: Optional[np.ndarray] = None):
    """instancer.pyx:38-54 + instancer.cpp:751-1037.  rays_o, rays_d [n,3], parameters [n,P]; u_offset [n] and u_choice
    [n, n_samples] are the uniform draws.  Returns the tuple of instancer.pyx:54 as numpy arrays, `hit` as a bool mask."""
    rays_o = np.asarray(rays_o, F32); rays_d = np.asarray(rays_d, F32)
    parameters = np.asarray(parameters, F32).reshape(rays_o.shape[0], -1)
    n, S, P = rays_o.shape[0], int(n_samples), parameters.shape[1]
    assert P == spec.n_parameters, "the textures list decides the width of a parameter row (instancer.cpp:755)"
    h = F32(step_size)
    # the buffers as instancer.pyx:41-50 allocates them
    rays_d_map = np.repeat(rays_d[:, None, :], S, axis=1).copy()
    t = np.zeros((n, S), F32); dists = np.zeros((n, S), F32); pts = np.zeros((n, S, 3), F32)
    color = np.zeros((n, 1, 3), F32); density = np.zeros((n, 1), F32)
    density_weight = np.ones((n, S), F32); instance_id = np.zeros((n, S), np.int32); hit = np.zeros(n, bool)
    params_map = np.repeat(parameters[:, None, :], S, axis=1).copy()

    def t_of(step, t_offset, segment_offset):
        t_mu = (F32(step) * h + t_offset) + segment_offset                              # :872, :985
        return t_mu, (get_mean_distance(t_mu, h) if spec.use_mean_distance else t_mu)

    for i in range(n):
        o, d = rays_o[i], rays_d[i]
        # at most MAX_TOTAL_HITS box crossings are kept (:539; here the first ones of the sorted list, in the reference the
        # first ones Embree's traversal meets -- a ray with more is flagged by the product and not comparable)
        hits = sorted([(tt, k, False) for tt, k in box_hits(spec, o, d)], key=lambda e: (e[0], e[1]))[:MAX_TOTAL_HITS]
        mh = mesh_hit(spec, o, d, full=True)
        tm = None if mh is None else mh[0]
        if tm is not None:
            hits.append((tm, INVALID, True))
        if not hits:                                                                    # :782
            continue
        hit[i] = True
        hits.sort(key=lambda e: (e[0], e[1]))                                           # :787, order of :441-452

        # segment lengths inside the union of the boxes (:800-826)
        active: set = set()
        total = F32(0.0); t_entry = F32(0.0)
        has_mesh = False
        segment_lengths = []
        for tt, k, is_mesh in hits:
            if is_mesh:
                if active:
                    total = total + (tt - t_entry)
                    segment_lengths.append(tt - t_entry)
                has_mesh = True
                break
            if k in active:
                active.discard(k)
                if not active:
                    total = total + (tt - t_entry)
                    segment_lengths.append(tt - t_entry)
            else:
                if not active:
                    t_entry = tt
                active.add(k)
        active = set()

        default_light = params_map[i, 0, spec.light_dir_idx:spec.light_dir_idx + 3].copy() if spec.light_dir_idx >= 0 else None
        default_str = params_map[i, 0, spec.light_strength_idx] if spec.light_strength_idx >= 0 else None

        if total > 0:
            necessary = int(np.uint32(total / h))                                       # :842
            n_steps = min(necessary, S)
            if n_steps == 0:
                dists[i, 0] = total
                t_offset = u_offset[i] * total
                n_steps = 1
            else:
                dists[i, :n_steps - 1] = h
                dists[i, n_steps - 1] = (h + total) - F32(n_steps) * h                  # :853
                t_offset = u_offset[i] * h
            # shadow samples of the ray (:861-862) and the interpolation state of :866-869
            interpolate = False
            if spec.cast_shadow_rays and spec.light_dir_idx >= 0:
                n_shadow = max(spec.min_shadow_samples, int(np.uint32(F32(spec.n_shadow_samples) * total)))
                interpolate = n_shadow < S
            seg_l = 0; k_shadow = 0; t_0_sh = t_1_sh = step_sh = F32(0.0); s_0 = s_1 = False
            # texture samples of the ray (:863) and their interpolation state (:867-870)
            tex_on = spec.inst_v is not None and bool(spec.tex_idx)                      # :911
            tex_interp = False
            if tex_on:
                n_tex = max(spec.min_texture_samples, int(np.uint32(F32(spec.n_texture_samples) * total)))
                tex_interp = n_tex < S
            default_parameters = parameters[i].copy()                                    # :870
            k_tex = 0; t_0_tx = t_1_tx = step_tx = F32(0.0); p_0 = p_1 = None
            segment_offset = F32(0.0); cleared = F32(0.0); t_entry = F32(0.0)
            step = 0
            for tt, k, is_mesh in hits:
                if step >= n_steps:
                    break
                t_mu, t_pt = t_of(step, t_offset, segment_offset)
                while active and t_pt < tt and step < n_steps:
                    t[i, step] = t_mu
                    pt = (o + t_pt * d).astype(F32)                                     # getPtOnRay :566-568
                    ids = sorted(active)                                                # std::set iterates in ascending order
                    if len(ids) == 1:
                        inst = ids[0]; density_weight[i, step] = 1.0
                    elif spec.sample_method == SAMPLE_RANDOM:                           # :672-677
                        inst = ids[min(int(u_choice[i, step] * F32(len(ids))), len(ids) - 1)]
                        density_weight[i, step] = F32(len(ids))
                    elif spec.sample_method == SAMPLE_NEAREST:                          # :681-692
                        best = F32(np.inf); inst = ids[0]
                        for c in ids:
                            dd = _norm(pt - spec.origins[c])
                            if dd < best:
                                inst = c; best = dd
                        density_weight[i, step] = 1.0
                    else:                                                               # :696-713
                        tr = F32(0.2) * F32(spec.patch_scale)
                        ds = [_norm(pt - spec.origins[c]) for c in ids]
                        mn = min(ds)
                        ws = [max((tr + mn) - w, F32(0.0)) for w in ds]
                        tot = F32(0.0)
                        for w in ws:
                            tot = tot + w
                        # std::discrete_distribution: the first index whose cumulative probability exceeds the draw
                        target = u_choice[i, step] * tot
                        acc = F32(0.0); pick = len(ids) - 1
                        for q, w in enumerate(ws):
                            acc = acc + w
                            if target < acc:
                                pick = q
                                break
                        inst = ids[pick]
                        density_weight[i, step] = tot / ws[pick]                        # 1 / probability
                    instance_id[i, step] = inst
                    if tex_on and tex_interp:                                           # :911-923: between the two texture samples around t_pt
                        while t_pt > t_1_tx:
                            t_0_tx = t_1_tx
                            k_tex += 1
                            t_1_tx = t_entry + F32(k_tex) * step_tx
                            p_0 = p_1
                            p_1 = get_parameters(spec, (o + t_1_tx * d).astype(F32), default_parameters)
                        w = (t_pt - t_0_tx) / step_tx
                        params_map[i, step] = p_0 * (F32(1.0) - w) + p_1 * w
                    elif tex_on:                                                        # :924-927: a query per step
                        params_map[i, step] = get_parameters(spec, pt, default_parameters)
                    if spec.light_dir_idx >= 0:                                         # :945-967
                        shadowed = False
                        if spec.cast_shadow_rays and interpolate:                       # :946-958
                            while t_pt > t_1_sh:
                                t_0_sh = t_1_sh
                                k_shadow += 1
                                t_1_sh = t_entry + F32(k_shadow) * step_sh
                                s_0 = s_1
                                s_1 = is_shadowed(spec, (o + t_1_sh * d).astype(F32), default_light)
                            w = (t_pt - t_0_sh) / step_sh >= F32(0.5)                    # nearest of the two shadow samples
                            shadowed = (not w and s_0) or (w and s_1)
                        elif spec.cast_shadow_rays:                                     # :959-961
                            shadowed = is_shadowed(spec, pt, default_light)
                        if shadowed:                                                    # getShadowedLightDir, :571-581
                            params_map[i, step, spec.light_dir_idx:spec.light_dir_idx + 3] = F32([0, 0, -1])
                        else:
                            src = (default_light - pt).astype(F32) if spec.light_strength_idx >= 0 else default_light
                            params_map[i, step, spec.light_dir_idx:spec.light_dir_idx + 3] = _linear_e(spec.dir_t[inst], _normalized(src))
                    if spec.light_strength_idx >= 0:                                    # :970-972, :583-588 (double)
                        dv = (default_light - pt).astype(F32)
                        d2 = dv[0] * dv[0] + (dv[1] * dv[1] + dv[2] * dv[2])               # squaredNorm
                        params_map[i, step, spec.light_strength_idx] = F32(float(default_str) / (4 * np.pi * float(d2) + float(F32(1e-6))))
                    pts[i, step] = _affine_e(spec.inv[inst], pt)                        # :975
                    rays_d_map[i, step] = _linear_e(spec.dir_t[inst], _normalized(d))   # :976, :561-563
                    step += 1
                    t_mu, t_pt = t_of(step, t_offset, segment_offset)
                if is_mesh:                                                             # :988
                    break
                if k in active:
                    active.discard(k)
                    if not active:
                        cleared = cleared + (tt - t_entry)                              # :996
                else:
                    if not active:
                        segment_offset = tt - cleared                                   # :1001
                        t_entry = tt
                        if tex_on and tex_interp:                                       # :989-998
                            seg_len = segment_lengths[seg_l]
                            n_seg = max(spec.min_texture_samples, int(np.uint32(F32(n_tex) * seg_len / total)))
                            step_tx = seg_len / F32(n_seg - 1)
                            k_tex = 1
                            t_0_tx = t_entry; t_1_tx = t_entry + step_tx
                            p_0 = get_parameters(spec, (o + t_0_tx * d).astype(F32), default_parameters)
                            p_1 = get_parameters(spec, (o + t_1_tx * d).astype(F32), default_parameters)
                        if spec.cast_shadow_rays and spec.light_dir_idx >= 0 and interpolate:      # :1018-1027
                            seg_len = segment_lengths[seg_l]
                            n_seg = max(spec.min_shadow_samples, int(np.uint32(F32(n_shadow) * seg_len / total)))
                            step_sh = seg_len / F32(n_seg - 1)
                            k_shadow = 1
                            t_0_sh = t_entry; t_1_sh = t_entry + step_sh
                            s_0 = is_shadowed(spec, (o + t_0_sh * d).astype(F32), default_light)
                            s_1 = is_shadowed(spec, (o + t_1_sh * d).astype(F32), default_light)
                        seg_l += 1                                                      # :1029
                    active.add(k)
        if has_mesh:                                                                    # :1013-1027: the instancer mesh is black, an auxiliary one shaded
            density[i, 0] = 1.0
            if spec.mesh_kind is not None and spec.mesh_kind[mh[1]] != 0:
                color[i, 0] = shade_mesh(spec, o, d, mh, default_light)
    return rays_d_map, pts, t, dists, color, density, density_weight, instance_id, hit, params_map


# ------------------------------------------------------------------------------------------------------------------------------
# DistributeInstancesOnMesh (instancer.cpp:233-390), restated sequentially: the twin of nerf_tex_amd.instancer.distribute_instances_on_mesh
# (product code, vectorised numpy), which tests compare with this one
# ------------------------------------------------------------------------------------------------------------------------------

def _cross(a, b):
    return np.asarray([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F32)


def mt19937_canonical_floats(seed: int):
    """The floats std::uniform_real_distribution<float> draws from std::mt19937(seed) (libstdc++: generate_canonical<float, 24> takes one
    32-bit word, word / 2^32 in float, pulled below 1 when it rounds up to it).  numpy's legacy MT19937 seeding is the standard's."""
    words = np.random.RandomState(int(seed))._bit_generator.random_raw
    while True:
        c = F32(words(1)[0]) / F32(4294967296.0)
        yield np.nextafter(F32(1), F32(0)) if c >= 1 else c


def distribute_instances_on_mesh(vertices, faces, normals, uv, scale, patch_origins=None, jitter_amount=0.0, seed=0):
    """The patch -> world matrices AddInstance is handed, in order, and the scale in force (scale <= 0: the average edge length, :243-245).
    Line by line: the tangent of a face from its texture coordinates added onto its three vertices (:250-264); per vertex the normal
    normalised, the tangent made orthogonal to it and normalised, the bitangent n x t (:265-275); with anchor points the frame at the
    closest point of the mesh within one average edge length, interpolated, re-orthogonalised, turned about its normal by
    jitter_amount * U(0, pi) (:300-342); without, one patch per DISTINCT vertex position (:346-367).  The average edge length is summed
    in double (igl's float reduction order is not knowable)."""
    V = np.asarray(vertices, F32).reshape(-1, 3); Fa = np.asarray(faces, np.int64).reshape(-1, 3)
    N = np.asarray(normals, F32).reshape(-1, 3).copy(); UV = np.asarray(uv, F32).reshape(-1, 2)
    total = 0.0
    for f in Fa:
        for j in range(3):
            e = V[f[j]] - V[f[(j + 1) % 3]]
            total += float(np.sqrt(e[0] * e[0] + (e[1] * e[1] + e[2] * e[2])))
    avg_edge = F32(total / (3 * len(Fa)))
    scale = F32(scale) if scale > 0 else avg_edge
    T = np.zeros_like(V); B = np.zeros_like(V)
    with np.errstate(divide="ignore", invalid="ignore"):
        for f in Fa:
            e0 = V[f[1]] - V[f[0]]; e1 = V[f[2]] - V[f[0]]
            uv0 = UV[f[1]] - UV[f[0]]; uv1 = UV[f[2]] - UV[f[0]]
            r = F32(1.0) / (uv0[0] * uv1[1] - uv0[1] * uv1[0])
            t = ((e0 * uv1[1] - e1 * uv0[1]) * r).astype(F32)
            for j in range(3):
                T[f[j]] = T[f[j]] + t
        for i in range(len(V)):
            N[i] = _normalized(N[i])
            n, t = N[i], T[i]
            t = (t - n * _dot3(n, t)).astype(F32)
            t = _normalized(t)
            T[i] = t
            B[i] = _cross(n, t)
    draws = mt19937_canonical_floats(seed)
    pi_f = F32(np.pi)

    def turn(b, n):                                                                       # Rodrigues about n, :330 / :355
        angle = F32(jitter_amount) * (next(draws) * (pi_f - F32(0)) + F32(0))
        c, s_ = F32(np.cos(angle)), F32(np.sin(angle))
        return ((b * c + _cross(n, b) * s_) + (n * _dot3(n, b)) * (F32(1) - c)).astype(F32)

    out = []
    if patch_origins is not None:
        for pt in np.asarray(patch_origins, F32).reshape(-1, 3):
            f, w = closest_point_on_mesh(V, Fa, pt, avg_edge)
            if f is None:
                raise ValueError("a patch origin lies further than one average edge length from the mesh (the reference then indexes with an invalid primID)")
            f = Fa[f]
            n = _normalized(_bary_mix(N, f, w)); t = _normalized(_bary_mix(T, f, w))
            b = _cross(n, t)
            if jitter_amount > 0:
                b = turn(b, n)
            t = _cross(b, n)
            m = np.eye(4, dtype=F32)
            m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = t * scale, b * scale, n * scale, pt
            out.append(m)
    else:
        seen = []
        for i in range(len(V)):
            if any(np.array_equal(V[i], s_) for s_ in seen):
                continue
            t, b, n = T[i], B[i], N[i]
            if jitter_amount > 0:
                b = turn(b, n)
                tc = _cross(n, b)
                t = ((F32(-1.0) if _dot3(t, tc) < 0 else F32(1.0)) * tc).astype(F32)
            m = np.eye(4, dtype=F32)
            m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = t * scale, b * scale, n * scale, V[i]
            out.append(m)
            seen.append(V[i].copy())
    return (np.stack(out) if out else np.zeros((0, 4, 4), F32)), float(scale)
