"""CPU oracle for the NeRF-Tex volumetric render path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm for the hot path named in
BASELINE.json.  It is the *checker* for the HIP product code: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The
product package (`nerf_tex_amd/`) never imports, links or executes anything in `oracle/`.

PARITY UNPINNED (arithmetic).  The reference (`/root/reference`, hbaatz/nerf-tex) ships no
tests, golden vectors or fixtures for this path, and every file on the path imports
`tensorflow` (pinned `tensorflow-gpu=2.4.1`, reference `environment.yml:12`), which is not
installed here and cannot be installed (no network).  So the arithmetic below is restated from
reading the reference source, not checked against outputs of the reference.  What IS pinned:
camera poses / material parameters, which come from the reference's TF-free modules
(`data/distribution.py`, `data/sampler.py`, `configs/*.py`) run in this container by
`oracle/gen_golden.py` and committed under `tests/golden/`.

TensorFlow semantics taken on trust (cannot be executed here):
  * `tf.linspace(0., 1., S)` (TF 2.4 `math_ops.linspace_nd`): interior points are
    `start + ((stop-start)/(S-1)) * k` evaluated in the tensor dtype, end points exact.
  * Keras `Dense` on rank-2 input: `matmul(x, kernel) + bias`, then the activation;
    `kernel` is `[in, out]` row-major; accumulation order unspecified.
  * `tf.math.cumprod(x, exclusive=True)`: sequential left-to-right product with leading 1.
  * `tf.where(c, a, b)`, `tf.gather_nd`, `tf.scatter_nd` (zeros where not scattered).

Every function takes `dtype` (np.float64 = the truth the parity gate is measured against,
np.float32 = what a float32 TF-CPU run would compute, up to summation order) and cites the
reference file:line it follows.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from math import tan
from typing import List, Optional, Sequence, Tuple

import numpy as np

F64 = np.float64
F32 = np.float32


# --------------------------------------------------------------------------------------
# Model description (what `model_config` of a reference config file holds)
# --------------------------------------------------------------------------------------
@dataclass
class ModelSpec:
    """Mirror of the kwargs of `network.model.ParamNerf` (model.py:58) / `Nerf` (model.py:9)."""

    kind: str = "ParamNerf"              # "ParamNerf" | "Nerf"
    n_parameters: Tuple[int, int] = (1, 6)  # [geometry, appearance]  (model.py:63-64)
    n_pos: int = 3
    pos_freq: int = 10                   # pos_embedding.n_freq_bands
    dir_freq: int = 4                    # dir_embedding.n_freq_bands
    param_freq: int = 4                  # param_embedding.n_freq_bands
    depth: int = 8
    width: int = 256
    skips: Tuple[int, ...] = (4,)
    color_depth: int = 1                 # ParamNerf only (model.py:118); Nerf has none
    pos_encoding: str = "fourier"        # "fourier" = FourierFeatures; "ipe" = IntegratedPositionalEncoding (n_pos = 6)
    param_depth: int = 0                 # ParamNerf only (model.py:88-101): Dense(param_width, relu) layers on the parameter features
    param_width: int = 128

    @property
    def n_geo(self) -> int:
        return 0 if self.kind == "Nerf" else int(self.n_parameters[0])

    @property
    def n_app(self) -> int:
        return 0 if self.kind == "Nerf" else int(self.n_parameters[1])

    @property
    def n_params(self) -> int:
        return self.n_geo + self.n_app

    @property
    def param_layers(self) -> int:
        return 0 if self.kind == "Nerf" else int(self.param_depth)

    def branch_dim(self, n: int) -> int:
        """width of a parameter branch as it is concatenated to pos_map / dir_map (model.py:89-93, 97-101): FF(params) itself,
        or the output of the last of `param_depth` Dense(param_width) layers"""
        if n == 0:
            return 0
        return self.param_width if self.param_layers > 0 else n * (1 + 2 * self.param_freq)

    @property
    def pos_map_dim(self) -> int:
        emb = 6 * self.pos_freq if self.pos_encoding == "ipe" else self.n_pos * (1 + 2 * self.pos_freq)
        return emb + self.branch_dim(self.n_geo)

    @property
    def dir_map_dim(self) -> int:
        return 3 * (1 + 2 * self.dir_freq) + self.branch_dim(self.n_app)


def layer_table(spec: ModelSpec) -> List[Tuple[str, int, int]]:
    """(name, in, out) of every Dense layer in `tf.keras.Model.get_weights()` order (each contributing
    kernel[in,out] then bias[out]) for the functional models of model.py:125 (ParamNerf) / :45 (Nerf).
    Keras sorts `model.layers` by graph depth, ties by traversal order from `outputs=[color, alpha]`
    (keras/engine/functional.py `_map_graph_network`, TF 2.4): trunk, feature, colour layers, color,
    and `alpha` LAST -- not the creation order of model.py:104-123, where alpha (:111) precedes feature (:114)."""
    # every Dense layer in the order a depth-first traversal from outputs=[color, alpha] FIRST meets it (a layer before its inputs,
    # inputs in the order of the call: concat([dir_map, feature]) visits the appearance branch before the feature layer), with its
    # graph depth = longest path to an output (concat nodes are layers too and take a level); then by decreasing depth, ties in
    # traversal order.  For param_depth = 0 this is trunk.., feature, colour layers, color_half, color, alpha.
    pd, pw, w = spec.param_layers, spec.param_width, spec.width
    ff = lambda n: n * (1 + 2 * spec.param_freq)
    cd = spec.color_depth if spec.kind == "ParamNerf" else 0
    seq = [("color", w // 2, 3, 0), ("color_half", w if cd > 0 else w + spec.dir_map_dim, w // 2, 1)]
    for i in reversed(range(cd)):
        seq.append((f"color_hidden{i}", w + spec.dir_map_dim if i == 0 else w, w, 1 + cd - i))
    d_dircat = cd + 2
    if pd > 0 and spec.n_app > 0:               # dir_map = concat([FF(dir), app branch]) is a node of its own at d_dircat + 1
        for i in reversed(range(pd)):
            seq.append((f"param_app{i}", ff(spec.n_app) if i == 0 else pw, pw, d_dircat + 2 + (pd - 1 - i)))
    d = d_dircat + 1                            # feature
    k_in = lambda i: spec.pos_map_dim if i == 0 else w + (spec.pos_map_dim if (i - 1) in spec.skips else 0)
    k_head = w + (spec.pos_map_dim if (spec.depth - 1) in spec.skips else 0)
    seq.append(("feature", k_head, w, d))
    d_alpha_path = 1                            # (alpha's own path to the trunk is shorter than the feature path: never the longest)
    for i in reversed(range(spec.depth)):
        d += 1 + (1 if i in spec.skips else 0)  # a skip behind layer i puts a concat node between it and its consumer
        seq.append((f"trunk{i}", k_in(i), w, d))
    if pd > 0 and spec.n_geo > 0:               # pos_map = concat([FF(pos), geometry branch]): a node at depth(trunk0) + 1
        for i in reversed(range(pd)):
            seq.append((f"param_geo{i}", ff(spec.n_geo) if i == 0 else pw, pw, d + 2 + (pd - 1 - i)))
    seq.append(("alpha", k_head, 1, 0))
    order = sorted(range(len(seq)), key=lambda j: (-seq[j][3], j))
    return [seq[j][:3] for j in order]


def macs_per_sample(spec: ModelSpec) -> int:
    return sum(i * o for _, i, o in layer_table(spec))


def n_weight_floats(spec: ModelSpec) -> int:
    return sum(i * o + o for _, i, o in layer_table(spec))


def split_blob(spec: ModelSpec, blob: np.ndarray) -> List[np.ndarray]:
    """Flat float32 blob (kernel, bias, kernel, bias, ... in `layer_table` order) -> list of arrays."""
    out, p = [], 0
    for _, i, o in layer_table(spec):
        out.append(blob[p:p + i * o].reshape(i, o)); p += i * o
        out.append(blob[p:p + o]); p += o
    assert p == blob.size, (p, blob.size)
    return out


# --------------------------------------------------------------------------------------
# Camera / ray generation
# --------------------------------------------------------------------------------------
def full_pixels(height: int, width: int) -> np.ndarray:
    """pixel_sampler.Full.__call__  (pixel_sampler.py:14-15): all pixels, row-major (row i, col j)."""
    k = np.arange(height * width, dtype=np.int32)
    return np.stack([k // width, k % width], -1)


def focal_from_angle(width: int, angle: float) -> float:
    """dataset.GenerateData return value (dataset.py:229): width / tan(angle / 2) / 2 (python float)."""
    return width / tan(angle / 2) / 2


def _normalize(v):
    return v / np.sqrt(np.sum(v * v))


def look_at(pos, to=(0., 0., 0.), offset=(0., 0., 0.), eps=1e-6, dtype=F32) -> np.ndarray:
    """dataset.look_at (dataset.py:231-238).  `pos` arrives as float64 numpy from the pose
    distribution; `pos - to + eps` with `to` a float32 tf.constant makes TF convert `pos` to
    float32, so the whole function runs in float32 in the reference; dtype=F64 gives the truth."""
    pos = np.asarray(pos, dtype=dtype)
    to = np.asarray(to, dtype=dtype)
    offset = np.asarray(offset, dtype=dtype)
    e = dtype(eps)
    v_forward = _normalize(pos - to + e)
    v_right = _normalize(np.cross(np.asarray([0, 0, 1.], dtype=dtype), v_forward) + e)
    v_up = _normalize(np.cross(v_forward, v_right) + e)
    top = np.stack([v_right, v_up, v_forward, pos + offset], axis=1)          # [3,4]
    return np.concatenate([top, np.asarray([[0, 0, 0, 1.]], dtype=dtype)], axis=0)


def rays_from_camera(image_plane_loc, height: int, width: int, focal: float, c2w, dtype=F32):
    """ray_sampler.rays_from_camera (ray_sampler.py:39-48)."""
    loc = np.asarray(image_plane_loc).astype(dtype)
    c2w = np.asarray(c2w, dtype=dtype)
    f = dtype(focal)
    half_w = dtype(.5 * width)          # `.5 * width` is python arithmetic, then cast
    half_h = dtype(.5 * height)
    dirs = np.stack([(loc[:, 1] + dtype(.5) - half_w) / f,
                     -(loc[:, 0] + dtype(.5) - half_h) / f,
                     -np.ones(loc.shape[0], dtype=dtype)], -1)
    rays_d = np.sum(dirs[:, None, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).copy()
    norm_xy = np.sqrt(np.sum(dirs[:, :2] * dirs[:, :2], -1))
    norm = np.sqrt(np.sum(dirs * dirs, -1))
    cone_scale = np.cos(np.arctan(norm_xy)) / norm / f
    return rays_o, rays_d, cone_scale[:, None]


def aabb(rays_o, rays_d, b_0, b_1, dtype=F32):
    """proxy.AABB.__call__ (proxy.py:13-35): slab test, [inf, inf] on a miss.  1/0 -> +-inf and
    0*inf -> NaN follow IEEE exactly as in the reference (no special-casing)."""
    o = np.asarray(rays_o, dtype=dtype)
    d = np.asarray(rays_d, dtype=dtype)
    b_0 = np.asarray(b_0, dtype=dtype)
    b_1 = np.asarray(b_1, dtype=dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = dtype(1.) / d
        t_0 = (b_0 - o) * inv
        t_1 = (b_1 - o) * inv
        t_0_tmp = t_0
        t_0 = np.where(t_0 < t_1, t_0, t_1)
        t_1 = np.where(t_0_tmp > t_1, t_0_tmp, t_1)
        # tf.reduce_max / reduce_min propagate NaN like numpy's max/min
        t_0 = np.max(t_0, axis=1)
        t_1 = np.min(t_1, axis=1)
        t_0_tmp = t_0
        inf = dtype(np.inf)
        t_0 = np.where(t_0 < t_1, t_0, inf)
        t_1 = np.where(t_0_tmp < t_1, t_1, inf)
    return np.stack([t_0, t_1], -1).astype(dtype)


def proxy_rays(image_plane_loc, height, width, focal, c2w, b_0, b_1, dtype=F32):
    """ray_sampler.Proxy.__call__ (ray_sampler.py:32-37): normalise rays_d, then AABB t-range."""
    rays_o, rays_d, cone_scale = rays_from_camera(image_plane_loc, height, width, focal, c2w, dtype)
    rays_d = rays_d / np.sqrt(np.sum(rays_d * rays_d, -1, keepdims=True))
    t = aabb(rays_o, rays_d, b_0, b_1, dtype)
    return rays_o, rays_d, t, cone_scale


def frustum_rays(image_plane_loc, height, width, focal, c2w, near, far, dtype=F32):
    """ray_sampler.Frustum.__call__ (ray_sampler.py:15-21): un-normalised rays_d, constant t."""
    rays_o, rays_d, cone_scale = rays_from_camera(image_plane_loc, height, width, focal, c2w, dtype)
    n = rays_o.shape[0]
    t = np.stack([np.full(n, near, dtype=dtype), np.full(n, far, dtype=dtype)], -1)
    return rays_o, rays_d, t, cone_scale


# --------------------------------------------------------------------------------------
# Model
# --------------------------------------------------------------------------------------
def fourier_features(x, n_freq_bands: int, dtype=F32):
    """layer.FourierFeatures.call (layer.py:8-23): [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...],
    every block the full D-vector; `freq * x` is rounded in the tensor dtype before sin/cos."""
    x = np.asarray(x, dtype=dtype)
    out = [x]
    for k in range(n_freq_bands):
        freq = dtype(2.0 ** k)
        out.append(np.sin(freq * x))
        out.append(np.cos(freq * x))
    return np.concatenate(out, -1)


def integrated_positional_encoding(x, n_freq_bands: int, dtype=F32):
    """layer.IntegratedPositionalEncoding.call (layer.py:25-41): x = [mean(3) | diagonal covariance(3)] ->
    [sin(y) e^(-y_var/2) | sin(y + pi/2) e^(-y_var/2)], y and y_var laid out frequency-major (index f*3+c)."""
    x = np.asarray(x, dtype=dtype)
    freq = (2.0 ** np.arange(n_freq_bands)).astype(dtype)
    y = (x[..., None, :3] * freq[:, None]).reshape(-1, 3 * n_freq_bands)             # :33
    y_var = (x[..., None, 3:] * freq[:, None] ** 2).reshape(-1, 3 * n_freq_bands)    # :34
    xx = np.concatenate([y, y + dtype(.5 * np.pi)], axis=-1)                           # :36
    vv = np.concatenate([y_var, y_var], axis=-1)
    return (np.sin(xx) * np.exp(dtype(-.5) * vv)).astype(dtype)                        # :38-41


def _dense(x, kernel, bias, dtype, relu):
    y = x @ np.asarray(kernel, dtype=dtype) + np.asarray(bias, dtype=dtype)
    return np.maximum(y, dtype(0)) if relu else y


def model_forward(weights: Sequence[np.ndarray], spec: ModelSpec, pos, dirs, params, dtype=F32,
                  return_intermediates: bool = False):
    """network.model.ParamNerf (model.py:58-125) / Nerf (model.py:9-45) forward on rank-2 inputs.
    weights = [k0, b0, k1, b1, ...] in `layer_table(spec)` order.  Returns (color[M,3], alpha[M,1])."""
    pos = np.asarray(pos, dtype=dtype)
    dirs = np.asarray(dirs, dtype=dtype)
    params = np.asarray(params, dtype=dtype).reshape(pos.shape[0], -1)
    g, a = spec.n_geo, spec.n_app
    inter = {}

    if spec.pos_encoding == "ipe":
        pos_map = integrated_positional_encoding(pos, spec.pos_freq, dtype)
    else:
        pos_map = fourier_features(pos, spec.pos_freq, dtype)                   # model.py:77
    dir_map = fourier_features(dirs, spec.dir_freq, dtype)                      # model.py:78
    if g > 0:                                                                  # model.py:88-93
        pos_map = np.concatenate([pos_map, fourier_features(params[:, :g], spec.param_freq, dtype)], -1)
    if a > 0:                                                                  # model.py:96-101
        dir_map = np.concatenate([dir_map, fourier_features(params[:, g:g + a], spec.param_freq, dtype)], -1)
    inter["pos_map"], inter["dir_map"] = pos_map, dir_map

    names = [n for n, _, _ in layer_table(spec)]
    assert len(weights) == 2 * len(names), (len(weights), len(names))
    W = {n: (weights[2 * j], weights[2 * j + 1]) for j, n in enumerate(names)}
    dense = lambda x, name, relu: _dense(x, W[name][0], W[name][1], dtype, relu)
    if spec.param_layers > 0:                                                  # model.py:88-101 with param_depth > 0
        if g > 0:
            pg = fourier_features(params[:, :g], spec.param_freq, dtype)
            for i in range(spec.param_layers):
                pg = dense(pg, f"param_geo{i}", True)
            pos_map = np.concatenate([pos_map[:, :pos_map.shape[1] - g * (1 + 2 * spec.param_freq)], pg], -1)
        if a > 0:
            pa = fourier_features(params[:, g:g + a], spec.param_freq, dtype)
            for i in range(spec.param_layers):
                pa = dense(pa, f"param_app{i}", True)
            dir_map = np.concatenate([dir_map[:, :dir_map.shape[1] - a * (1 + 2 * spec.param_freq)], pa], -1)
        inter["pos_map"], inter["dir_map"] = pos_map, dir_map
    h = pos_map
    for i in range(spec.depth):                                                # model.py:104-108
        h = dense(h, f"trunk{i}", True)
        inter[f"trunk{i}"] = h
        if i in spec.skips:
            h = np.concatenate([pos_map, h], -1)
    alpha = dense(h, "alpha", False)                                           # model.py:111 (last in get_weights())
    h = dense(h, "feature", False)                                             # model.py:114
    inter["feature"] = h
    h = np.concatenate([dir_map, h], -1)                                       # model.py:115
    if spec.kind == "ParamNerf":
        for i in range(spec.color_depth):                                      # model.py:118-119
            h = dense(h, f"color_hidden{i}", True)
            inter[f"color_hidden{i}"] = h
    h = dense(h, "color_half", True)                                           # model.py:122
    inter["color_half"] = h
    color = dense(h, "color", False)                                           # model.py:123
    if return_intermediates:
        return color, alpha, inter
    return color, alpha


# --------------------------------------------------------------------------------------
# Renderer
# --------------------------------------------------------------------------------------
def linspace_tf(n: int, dtype=F32):
    """tf.linspace(0., 1., n) as TF 2.4 evaluates it (see module docstring)."""
    if n == 1:
        return np.zeros(1, dtype=dtype)
    delta = (dtype(1.) - dtype(0.)) / dtype(n - 1)
    inner = dtype(0.) + delta * np.arange(1, n - 1).astype(dtype)
    return np.concatenate([np.zeros(1, dtype), inner.astype(dtype), np.ones(1, dtype)])


def z_values(t, n_samples: int, dtype=F32):
    """renderer.py:101-103 (no jitter)."""
    t = np.asarray(t, dtype=dtype)
    t_vals = linspace_tf(n_samples, dtype)
    return t[:, None, 0] * (dtype(1) - t_vals) + t[:, None, 1] * t_vals


def jitter_bounds(z_vals):
    """renderer.py:107-109: the [lower, upper] interval a perturbed sample is drawn from;
    z = lower + (upper - lower) * U[0,1) (renderer.py:110-111)."""
    mids = z_vals.dtype.type(.5) * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = np.concatenate([mids, z_vals[..., -1:]], -1)
    lower = np.concatenate([z_vals[..., :1], mids], -1)
    return lower, upper


def philox4x32_10(c0, c1, c2, c3, k0, k1, words: int = 1):
    """Word 0 (words=1) or words (0, 1) (words=2) of the Philox4x32-10 block (Salmon et al., SC'11; the generator behind
    tf.random.uniform / tf.random.normal) at counter (c0, c1, c2, c3) under key (k0, k1); numpy uint32 arrays, element-wise.
    Restated from the published algorithm."""
    u32, u64 = np.uint32, np.uint64
    c0, c1, c2, c3 = (np.asarray(c, dtype=u32) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = u32(k0); k1 = u32(k1)
    for _ in range(10):
        p0 = u64(0xD2511F53) * c0.astype(u64)
        p1 = u64(0xCD9E8D57) * c2.astype(u64)
        hi0, lo0 = (p0 >> u64(32)).astype(u32), p0.astype(u32)
        hi1, lo1 = (p1 >> u64(32)).astype(u32), p1.astype(u32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = u32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF); k1 = u32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return c0 if words == 1 else (c0, c1)


def uniform01_from_bits(x):
    """float32 in [0,1) from the low 23 bits, TensorFlow's Uint32ToFloat (random_distributions.h)."""
    bits = (np.asarray(x, dtype=np.uint32) & np.uint32(0x7FFFFF)) | np.uint32(0x3F800000)
    return bits.view(np.float32) - np.float32(1.0)


def global_ray_index(n_rays: int, ray_index=None):
    """The ray index that keys the product's generators (include/nerftex.h: ntx_render_opts): local ray k counts as
    index0 + (k // run_length) * run_stride + k % run_length; None = k itself."""
    k = np.arange(n_rays, dtype=np.int64)
    if ray_index is None:
        return k
    i0, run, stride = (int(v) for v in ray_index)
    return i0 + (k // run) * stride + k % run


def jitter_uniforms(n_rays: int, n_points: int, seed: int, ray_index=None):
    """The draws the product uses for the stratified jitter (include/nerftex.h: ntx_sample_depths): Philox counter
    (sample index, ray index lo, ray index hi, 0), key (seed lo, seed hi).  TensorFlow's own stream cannot be
    reproduced; this pins OUR stream so the jittered render can be compared sample for sample."""
    ray = global_ray_index(n_rays, ray_index).astype(np.uint64)[:, None]
    i = np.arange(n_points, dtype=np.uint32)[None, :]
    bits = philox4x32_10(i, (ray & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ray >> np.uint64(32)).astype(np.uint32),
                         np.uint32(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return uniform01_from_bits(bits)


def noise_normals(n_rays: int, n_points: int, seed: int, ray_index=None, dtype=F32):
    """The N(0,1) draws the product uses for raw_noise_std (renderer.py:190-192 / 335-337; include/nerftex.h:
    ntx_render_opts): words 0 and 1 of the Philox block at counter (sample index, ray index lo, ray index hi, 1) through the
    first output of tf.random.normal's Box-Muller transform (random_distributions.h BoxMullerFloat: u1 = max(U(x0), 1e-7),
    sqrt(-2 ln u1) * sin(2 pi U(x1))).  The uniforms are float32 as in TensorFlow; the transform is evaluated in `dtype`."""
    ray = global_ray_index(n_rays, ray_index).astype(np.uint64)[:, None]
    i = np.arange(n_points, dtype=np.uint32)[None, :]
    x0, x1 = philox4x32_10(i, (ray & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ray >> np.uint64(32)).astype(np.uint32),
                           np.uint32(1), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, words=2)
    u1 = np.maximum(uniform01_from_bits(x0), np.float32(1.0e-7)).astype(dtype)
    v1 = dtype(2.0 * np.pi) * uniform01_from_bits(x1).astype(dtype) if dtype is F64 else np.float32(6.2831855) * uniform01_from_bits(x1)
    return (np.sin(v1) * np.sqrt(dtype(-2.0) * np.log(u1))).astype(dtype)


def z_values_perturbed(t, n_points: int, seed: int, dtype=F32, ray_index=None):
    """renderer.py:101-111 with z_rand = jitter_uniforms (float32 draws, arithmetic in `dtype`)."""
    z = z_values(t, n_points, dtype)
    lower, upper = jitter_bounds(z)
    return lower + (upper - lower) * jitter_uniforms(z.shape[0], n_points, seed, ray_index).astype(dtype)


def evaluate_model(weights, spec, pts, dirs, parameters, cone_scale, z_vals, blur_idx, net_chunk, dtype=F32):
    """Renderer.evaluate_model (renderer.py:145-168).  The blur product of :155-158 is evaluated in the dtype its operands
    arrive in (render_rays' `points_dtype`), the model in `dtype`."""
    n, S = pts.shape[0], pts.shape[1]
    pos_flat = pts.reshape(-1, pts.shape[-1])
    dirs_flat = np.repeat(dirs, S, axis=0)
    params_flat = np.repeat(parameters, S, axis=0)
    if blur_idx is not None:                                                   # renderer.py:155-158
        blur_scale = cone_scale[..., None, :] * z_vals[..., :, None]
        blur_scale_flat = blur_scale.reshape(-1, 1)
        params_flat = np.concatenate([params_flat[:, :blur_idx],
                                      params_flat[:, blur_idx, None] * blur_scale_flat,
                                      params_flat[:, blur_idx + 1:]], -1)
    pos_flat, dirs_flat, params_flat = (np.asarray(a, dtype=dtype) for a in (pos_flat, dirs_flat, params_flat))
    color, alpha = [], []
    for i in range(0, pos_flat.shape[0], net_chunk):                           # renderer.py:160-163
        c, a = model_forward(weights, spec, pos_flat[i:i + net_chunk], dirs_flat[i:i + net_chunk],
                             params_flat[i:i + net_chunk], dtype)
        color.append(c); alpha.append(a)
    color = np.concatenate(color, 0); alpha = np.concatenate(alpha, 0)
    return color.reshape(n, S, 3), alpha.reshape(n, S)


def map_model_output(color, alpha, z_vals, rays_d, composite_bkgd, bkgd_color, map_exr=False,
                     noise=None, dtype=F32):
    """Renderer.map_model_output (renderer.py:170-213).  `noise` = the N(0, raw_noise_std) draw
    (renderer.py:190-192) supplied by the caller, or None."""
    color = np.asarray(color, dtype=dtype); alpha = np.asarray(alpha, dtype=dtype)
    z_vals = np.asarray(z_vals, dtype=dtype); rays_d = np.asarray(rays_d, dtype=dtype)
    dists = z_vals[..., 1:] - z_vals[..., :-1]                                  # :174
    dists = np.concatenate([dists, dists[..., -1:]], -1)                        # :177 (copy, not 1e10)
    dists = dists * np.sqrt(np.sum(rays_d[..., None, :] ** 2, -1))              # :180
    if map_exr:                                                                 # :182-187
        with np.errstate(over="ignore"):
            color_map = np.where(color > 0, color, np.exp(np.minimum(color, dtype(0))) - dtype(1)) + dtype(1)
    else:
        with np.errstate(over="ignore"):
            color_map = dtype(1) / (dtype(1) + np.exp(-color))
    if noise is not None:
        alpha = alpha + np.asarray(noise, dtype=dtype)
    alpha_map = dtype(1) - np.exp(-np.maximum(alpha, dtype(0)) * dists)         # :195
    trans = (dtype(1.) - alpha_map) + dtype(1e-10)                              # `1.-alpha_map + 1e-10`
    cum = np.cumprod(trans, axis=-1, dtype=dtype)
    excl = np.concatenate([np.ones_like(cum[..., :1]), cum[..., :-1]], -1)      # exclusive=True
    weights = alpha_map * excl                                                  # :198
    color_out = np.sum(weights[..., None] * color_map, axis=-2, dtype=dtype)    # :201
    depth = np.sum(weights * z_vals, axis=-1, dtype=dtype)                      # :204 (not returned)
    alpha_out = np.sum(weights, -1, dtype=dtype)                                # :207
    if composite_bkgd:                                                          # :210-211
        color_out = color_out + (dtype(1.) - alpha_out[..., None]) * np.asarray(bkgd_color, dtype=dtype)
    return color_out.astype(dtype), alpha_out.astype(dtype), weights.astype(dtype), depth


def render_rays(weights, spec, rays_o, rays_d, t, parameters, cone_scale, n_samples, composite_bkgd,
                bkgd_color, blur_idx=None, map_exr=False, net_chunk=65536, z_override=None, dtype=F32,
                return_aux=False, noise=None, points_dtype=None):
    """Renderer.render_rays (renderer.py:92-143) with perturb=False, n_importance=0 (TF's RNG stream cannot be
    reproduced).  `z_override` [n,S] replaces z_vals, standing in for the jittered samples of renderer.py:106-111 when the
    caller draws them itself; `noise` [n,S] = raw_noise_std * N(0,1), standing in for tf.random.normal of :190-192.

    `points_dtype` (default = `dtype`): the precision of the ELEMENTWISE ray arithmetic in front of the network -- rays_d_n (:98),
    z_vals (:101-102), pts (:114), the blur product (:155-158).  These float32 operations have no summation order: any float32
    run of the reference, whatever its BLAS, feeds the network exactly these bits.  `dtype=float64, points_dtype=float32` is
    therefore "the exact network and composite on the sample points a float32 run evaluates": it separates the rounding of the
    inputs (sample positions rounded to float32 in front of sin(2^9 x), which no float32 implementation can avoid) from the
    arithmetic of the network and the composite (which an implementation answers for)."""
    pd = dtype if points_dtype is None else points_dtype
    rays_o = np.asarray(rays_o, dtype=pd); rays_d = np.asarray(rays_d, dtype=pd)
    t = np.asarray(t, dtype=pd); parameters = np.asarray(parameters, dtype=pd)
    cone_scale = np.asarray(cone_scale, dtype=pd)
    rays_d_n = rays_d / np.sqrt(np.sum(rays_d * rays_d, -1, keepdims=True))     # :98
    z_vals = z_values(t, n_samples, pd) if z_override is None else np.asarray(z_override, dtype=pd)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]    # :114 (un-normalised d)
    color, alpha = evaluate_model(weights, spec, pts, rays_d_n, parameters, cone_scale, z_vals,
                                  blur_idx, net_chunk, dtype)
    z_vals = np.asarray(z_vals, dtype=dtype); rays_d = np.asarray(rays_d, dtype=dtype)
    color_map, alpha_map, w, _ = map_model_output(color, alpha, z_vals, rays_d, composite_bkgd,
                                                  bkgd_color, map_exr, noise, dtype)
    out = {"color_pred": color_map, "alpha_pred": alpha_map}
    if return_aux:
        out.update({"z_vals": z_vals, "pts": pts, "raw_color": color, "raw_alpha": alpha, "weights": w})
    return out


def sample_pdf(bins, weights, n_samples: int, det: bool = False, u=None, dtype=F32, return_conditioning: bool = False):
    """renderer.sample_pdf (renderer.py:589-617): inverse-CDF sampling of the piecewise-constant pdf given by
    `weights` over `bins`.  det=True uses u = tf.linspace(0, 1, n); otherwise the caller supplies the
    uniform draws `u` [n_rays, n_samples] (TF's RNG stream cannot be reproduced).

    return_conditioning: also `allowed` [n, n_samples], how far a FLOAT32 evaluation of this very function may place each sample
    from the value returned here (call it with dtype=float64) -- an error model of the reference's own arithmetic, not of any
    implementation.  t = (u - cdf[below]) / denom with cdf a float32 running sum of n_bins terms <= 1: each entry carries up to
    eps_c = n_bins * 2^-24, so t moves by up to 2 eps_c / denom and the sample by that times the bin width -- negligible in a bin
    that holds real weight (denom ~ 1e-2: 1e-5 of a bin) and the WHOLE bin where the weight is 1e-5-ish, i.e. in the empty bins of a
    dense medium, where the pdf is 1e-5 / sum(w) by construction (:593).  There the reference is also discontinuous: `denom < 1e-5
    -> 1` (:613-614) sits exactly on those bins and snaps every u to the lower edge, so whether a u within eps_c of a cdf entry --
    tf.linspace hits the flat stretches exactly -- lands in this bin or the next (`searchsorted`, :606), i.e. one bin width lower or
    higher, is decided by the last bit.  Two correct float32 implementations (TensorFlow on two BLAS builds) differ by this much."""
    bins = np.asarray(bins, dtype=dtype)
    w = np.asarray(weights, dtype=dtype) + dtype(1e-5)                           # :593
    pdf = w / np.sum(w, -1, keepdims=True, dtype=dtype)
    cdf = np.cumsum(pdf, -1, dtype=dtype)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)                 # :596
    if det:                                                                      # :599-601
        u = np.broadcast_to(linspace_tf(n_samples, dtype), cdf.shape[:-1] + (n_samples,))
    else:
        u = np.asarray(u, dtype=dtype)
    inds = np.stack([np.searchsorted(c, uu, side="right") for c, uu in zip(cdf, u)])   # :606
    below = np.maximum(0, inds - 1)
    above = np.minimum(cdf.shape[-1] - 1, inds)
    cdf_g0 = np.take_along_axis(cdf, below, -1); cdf_g1 = np.take_along_axis(cdf, above, -1)
    bins_g0 = np.take_along_axis(bins, below, -1); bins_g1 = np.take_along_axis(bins, above, -1)
    denom_raw = cdf_g1 - cdf_g0
    denom = np.where(denom_raw < dtype(1e-5), np.ones_like(denom_raw), denom_raw)   # :614
    t = (u - cdf_g0) / denom
    samples = (bins_g0 + t * (bins_g1 - bins_g0)).astype(dtype)                  # :616
    if not return_conditioning:
        return samples
    n_bins = cdf.shape[-1]
    eps_c = n_bins * 2.0 ** -24
    width = np.abs(bins_g1 - bins_g0)
    nb = bins.shape[-1]
    w_lo = np.abs(np.take_along_axis(bins, np.maximum(below - 1, 0), -1) - bins_g0)          # the neighbouring bins' widths
    w_hi = np.abs(np.take_along_axis(bins, np.minimum(above + 1, nb - 1), -1) - bins_g1)
    allowed = width * np.minimum(1.0, 2 * eps_c / np.maximum(denom_raw, 1e-30))
    near_switch = np.abs(denom_raw / 1e-5 - 1) <= 2 * eps_c / 1e-5                # either side of :613 -> anywhere in the bin
    allowed = np.where(near_switch, width, allowed)
    allowed = allowed + np.where(np.abs(u - cdf_g0) <= 2 * eps_c, w_lo, 0) + np.where(np.abs(cdf_g1 - u) <= 2 * eps_c, w_hi + width, 0)
    allowed = allowed + 4 * 2.0 ** -24 * np.abs(samples)                          # rounding of the result itself
    return samples, allowed.astype(np.float64)


def render_rays_hierarchical(weights_coarse, weights_fine, spec, rays_o, rays_d, t, parameters, cone_scale, n_samples,
                             n_importance, composite_bkgd, bkgd_color, perturb=True, u=None, blur_idx=None,
                             map_exr=False, net_chunk=65536, dtype=F32):
    """Renderer.render_rays with n_importance > 0 (renderer.py:92-143), jitter of the coarse samples left out.
    Note the reference quirk `det=self.perturb` (:128): perturb=True gives the DETERMINISTIC u.
    "z_allowed" [n, n_importance]: sample_pdf's float32 conditioning of every importance depth (see there)."""
    coarse = render_rays(weights_coarse, spec, rays_o, rays_d, t, parameters, cone_scale, n_samples, composite_bkgd,
                         bkgd_color, blur_idx, map_exr, net_chunk, None, dtype, return_aux=True)
    z_vals, w = coarse["z_vals"], coarse["weights"]
    z_mid = dtype(.5) * (z_vals[..., 1:] + z_vals[..., :-1])                     # :127
    z_samples, allowed = sample_pdf(z_mid, w[..., 1:-1], n_importance, det=perturb, u=u, dtype=dtype, return_conditioning=True)   # :128
    z_all = np.sort(np.concatenate([z_vals, z_samples], -1), -1)                 # :130
    fine = render_rays(weights_fine if weights_fine is not None else weights_coarse, spec, rays_o, rays_d, t,
                       parameters, cone_scale, n_samples + n_importance, composite_bkgd, bkgd_color, blur_idx, map_exr,
                       net_chunk, z_all, dtype)                                  # :131-136
    return {"color_pred": fine["color_pred"], "alpha_pred": fine["alpha_pred"],
            "color_pred_coarse": coarse["color_pred"], "alpha_pred_coarse": coarse["alpha_pred"],
            "z_vals": z_all, "z_samples": z_samples, "z_allowed": allowed}


def cone_segment_gaussians(rays_o, rays_d, t_vals, radii, dtype=F32):
    """MipRenderer.get_cone_segment_gaussians (renderer.py:411-437): t_vals [n,S+1] edges, radii [n,1]."""
    t0 = t_vals[..., :-1]; t1 = t_vals[..., 1:]
    mu = (t0 + t1) / dtype(2); hw = (t1 - t0) / dtype(2)
    t_mean = mu + (dtype(2) * mu * hw ** 2) / (dtype(3) * mu ** 2 + hw ** 2)
    t_var = (hw ** 2) / dtype(3) - dtype(4 / 15) * ((hw ** 4 * (dtype(12) * mu ** 2 - hw ** 2)) / (dtype(3) * mu ** 2 + hw ** 2) ** 2)
    r_var = radii ** 2 * ((mu ** 2) / dtype(4) + dtype(5 / 12) * hw ** 2 - dtype(4 / 15) * (hw ** 4) / (dtype(3) * mu ** 2 + hw ** 2))
    mean = rays_o[..., None, :] + rays_d[..., None, :] * t_mean[..., None]
    d_mag_sq = np.maximum(dtype(1e-10), np.sum(rays_d ** 2, axis=-1, keepdims=True))
    d_outer_diag = rays_d ** 2
    null_outer_diag = dtype(1) - d_outer_diag / d_mag_sq
    cov_diag = t_var[..., None] * d_outer_diag[..., None, :] + r_var[..., None] * null_outer_diag[..., None, :]
    return mean.astype(dtype), cov_diag.astype(dtype)


def mip_render_rays(weights, spec, rays_o, rays_d, t, parameters, cone_scale, n_samples, blur_idx, composite_bkgd,
                    bkgd_color, map_exr=False, net_chunk=65536, z_override=None, dtype=F32):
    """MipRenderer.render_rays + map_model_output (renderer.py:365-473), perturb=False.  `parameters` [n, P+1]
    still holds the blur parameter at `blur_idx`; the model (`spec`, an IPE ParamNerf) sees the other P."""
    rays_o = np.asarray(rays_o, dtype=dtype); rays_d = np.asarray(rays_d, dtype=dtype)
    t = np.asarray(t, dtype=dtype); parameters = np.asarray(parameters, dtype=dtype)
    cone_scale = np.asarray(cone_scale, dtype=dtype).reshape(rays_o.shape[0], 1)
    rays_d_n = rays_d / np.sqrt(np.sum(rays_d * rays_d, -1, keepdims=True))
    z_vals = z_values(t, n_samples + 1, dtype) if z_override is None else np.asarray(z_override, dtype=dtype)   # :374-376
    blur = parameters[..., blur_idx, None] * cone_scale                                                       # :385
    params = np.concatenate([parameters[..., :blur_idx], parameters[..., blur_idx + 1:]], axis=-1)          # :386
    mean, cov = cone_segment_gaussians(rays_o, rays_d, z_vals, blur, dtype)
    pts = np.concatenate([mean, cov], axis=-1)                                                                # :390
    color, alpha = evaluate_model(weights, spec, pts, rays_d_n, params, None, None, None, net_chunk, dtype)
    dists = (z_vals[..., 1:] - z_vals[..., :-1]) * np.sqrt(np.sum(rays_d[..., None, :] ** 2, -1))            # :441-444
    with np.errstate(over="ignore"):
        cm = (np.where(color > 0, color, np.exp(np.minimum(color, dtype(0))) - dtype(1)) + dtype(1)) if map_exr \
            else dtype(1) / (dtype(1) + np.exp(-color))
    am = dtype(1) - np.exp(-np.maximum(alpha, dtype(0)) * dists)                                              # :459
    trans = (dtype(1.) - am) + dtype(1e-10)
    cum = np.cumprod(trans, axis=-1, dtype=dtype)
    w = am * np.concatenate([np.ones_like(cum[..., :1]), cum[..., :-1]], -1)                                  # :462
    c = np.sum(w[..., None] * cm, axis=-2, dtype=dtype); a = np.sum(w, -1, dtype=dtype)
    if composite_bkgd:
        c = c + (dtype(1.) - a[..., None]) * np.asarray(bkgd_color, dtype=dtype)
    return {"color_pred": c.astype(dtype), "alpha_pred": a.astype(dtype), "z_vals": z_vals}


def mip_instance_cov(rays_d, t_vals, radii, dists, dtype=F32):
    """MipInstanceRenderer.get_cone_segment_gaussians (renderer.py:570-587) on flat samples: mu = t, hw = dists."""
    mu, hw = t_vals, dists
    t_var = (hw ** 2) / dtype(3) - dtype(4 / 15) * ((hw ** 4 * (dtype(12) * mu ** 2 - hw ** 2)) / (dtype(3) * mu ** 2 + hw ** 2) ** 2)
    r_var = radii ** 2 * ((mu ** 2) / dtype(4) + dtype(5 / 12) * hw ** 2 - dtype(4 / 15) * (hw ** 4) / (dtype(3) * mu ** 2 + hw ** 2))
    d_mag_sq = np.maximum(dtype(1e-10), np.sum(rays_d ** 2, axis=-1, keepdims=True))
    d_outer_diag = rays_d ** 2
    return (t_var[:, None] * d_outer_diag + r_var[:, None] * (dtype(1) - d_outer_diag / d_mag_sq)).astype(dtype)


def mip_instance_evaluate_model(weights, spec, rays_d_map, pts, t, dists, color_last, alpha_last, alpha_weight, hit,
                                params_map, cone_scale, blur_idx, patch_scale=1.0, density_scale=1.0,
                                density_reweighting=True, map_exr=False, composite_bkgd=False, bkgd_color=(1., 1., 1.),
                                dtype=F32):
    """MipInstanceRenderer.evaluate_model (renderer.py:485-568): the InstanceRenderer tail with the model fed
    (sample point, cone covariance) and the blur parameter spliced out."""
    n_rays, S = dists.shape
    f = lambda a: np.asarray(a, dtype=dtype)
    rays_d_map, pts, t, dists, params_map = f(rays_d_map), f(pts), f(t), f(dists), f(params_map)
    cone_scale = f(cone_scale).reshape(n_rays, 1)
    idxs = np.nonzero(np.asarray(hit))[0]
    if idxs.shape[0] == 0:
        return np.zeros((n_rays, 3), dtype), np.zeros((n_rays,), dtype)
    g = lambda a: np.asarray(a)[idxs]
    rays_d_map, pts, t, dists, params_map, cone = g(rays_d_map), g(pts), g(t), g(dists), g(params_map), g(cone_scale)
    color_last, alpha_last, alpha_weight = g(f(color_last)), g(f(alpha_last)), g(f(alpha_weight))
    blur = params_map[..., blur_idx] * cone[..., None, 0] / dtype(patch_scale)                               # :511
    params_map = np.concatenate([params_map[..., :blur_idx], params_map[..., blur_idx + 1:]], axis=-1)      # :512
    pos_flat = pts.reshape(-1, 3); dirs_flat = rays_d_map.reshape(-1, 3)
    params_flat = params_map.reshape(pos_flat.shape[0], params_map.shape[-1])
    sel = np.nonzero(dists.reshape(-1) > 0)[0]                                                               # :526
    color = np.zeros((pos_flat.shape[0], 3), dtype); alpha = np.zeros((pos_flat.shape[0], 1), dtype)
    if sel.shape[0] > 0:
        cov = mip_instance_cov(dirs_flat[sel], t.reshape(-1)[sel], blur.reshape(-1)[sel], dists.reshape(-1)[sel], dtype)
        c, a = model_forward(weights, spec, np.concatenate([pos_flat[sel], cov], -1), dirs_flat[sel], params_flat[sel], dtype)
        color[sel] = c; alpha[sel] = a
    alpha = alpha.reshape(pts.shape[:-1]) * (alpha_weight if density_reweighting else dtype(1)) * dtype(density_scale)
    cm, am = instance_map_model_output(color.reshape(pts.shape), color_last, alpha, alpha_last, dists, patch_scale,
                                       composite_bkgd, bkgd_color, map_exr, False, dtype)
    color_map = np.zeros((n_rays, 3), dtype); alpha_map = np.zeros((n_rays,), dtype)
    color_map[idxs] = cm; alpha_map[idxs] = am
    return color_map, alpha_map


def renderer_call(weights, spec, rays_o, rays_d, t, parameters, cone_scale, n_samples=64,
                  composite_bkgd=False, bkgd_color=(1., 1., 1.), blur_idx=None, map_exr=False,
                  render_chunk=32768, net_chunk=65536, dtype=F32, points_dtype=None):
    """Renderer.__call__ (renderer.py:47-90): inputs are batched [B, HW, ...], `parameters` [B, P]."""
    rays_o = np.asarray(rays_o, dtype=dtype); rays_d = np.asarray(rays_d, dtype=dtype)
    t = np.asarray(t, dtype=dtype); parameters = np.asarray(parameters, dtype=dtype)
    cone_scale = np.asarray(cone_scale, dtype=dtype)
    B, HW = rays_o.shape[0], rays_o.shape[1]
    o_f = rays_o.reshape(-1, 3); d_f = rays_d.reshape(-1, 3); t_f = t.reshape(-1, 2)
    p_f = np.repeat(parameters, HW, axis=0)                                     # :54
    c_f = cone_scale.reshape(-1, cone_scale.shape[-1])
    hit = t_f[:, 0] != np.inf                                                   # :58 (NaN counts as hit)
    idxs = np.nonzero(hit)[0]
    outs = {}
    for i in range(0, idxs.shape[0], render_chunk):                             # :72-77
        sl = idxs[i:i + render_chunk]
        o = render_rays(weights, spec, o_f[sl], d_f[sl], t_f[sl], p_f[sl], c_f[sl], n_samples,
                        composite_bkgd, bkgd_color, blur_idx, map_exr, net_chunk, None, dtype, points_dtype=points_dtype)
        for k, v in o.items():
            outs.setdefault(k, []).append(v)
    result = {}
    keys = outs.keys() if outs else ("color_pred", "alpha_pred")
    for k in keys:
        tail = (3,) if "color" in k else ()
        full = np.zeros((B * HW,) + tail, dtype=dtype)                          # scatter_nd zeros :83
        if outs:
            full[idxs] = np.concatenate(outs[k], 0)
        if composite_bkgd and "color" in k:                                     # :85-86
            full[~hit] += np.asarray(bkgd_color, dtype=dtype)
        result[k] = full.reshape((B, HW) + tail)                                # :87
    return result


def instance_map_model_output(color, color_last, alpha, alpha_last, dists, patch_scale, composite_bkgd, bkgd_color,
                              map_exr=False, false_color=False, dtype=F32, noise=None):
    """InstanceRenderer.map_model_output (renderer.py:318-354): S marched samples plus ONE appended
    sample (color_last [n,1,3] taken as is, alpha_last [n,1] taken as an alpha, not a density)."""
    color = np.asarray(color, dtype=dtype); alpha = np.asarray(alpha, dtype=dtype)
    dists = np.asarray(dists, dtype=dtype)
    color_last = np.asarray(color_last, dtype=dtype); alpha_last = np.asarray(alpha_last, dtype=dtype)
    if false_color:                                                              # :322-323
        color_map = np.concatenate([color, color_last], axis=1)
    else:
        with np.errstate(over="ignore"):
            if map_exr:                                                          # :325-327
                cm = np.where(color > 0, color, np.exp(np.minimum(color, dtype(0))) - dtype(1)) + dtype(1)
            else:                                                                # :329-330
                cm = dtype(1) / (dtype(1) + np.exp(-color))
        color_map = np.concatenate([cm, color_last], axis=1)                     # :331
    if noise is not None:                                                        # :335-337: raw_noise_std * N(0,1), [n,S]
        alpha = alpha + np.asarray(noise, dtype=dtype)
    alpha_map = np.concatenate([dtype(1) - np.exp(-np.maximum(alpha, dtype(0)) * dists / dtype(patch_scale)),
                                alpha_last], axis=1)                             # :339
    trans = (dtype(1.) - alpha_map) + dtype(1e-10)
    cum = np.cumprod(trans, axis=-1, dtype=dtype)
    excl = np.concatenate([np.ones_like(cum[..., :1]), cum[..., :-1]], -1)
    weights = alpha_map * excl                                                   # :342
    color_out = np.sum(weights[..., None] * color_map, axis=-2, dtype=dtype)     # :345
    alpha_out = np.sum(weights, -1, dtype=dtype)                                 # :348
    if composite_bkgd:                                                           # :351-352
        color_out = color_out + (dtype(1.) - alpha_out[..., None]) * np.asarray(bkgd_color, dtype=dtype)
    return color_out.astype(dtype), alpha_out.astype(dtype)


def instance_evaluate_model(weights, spec, rays_d_map, pts, t, dists, color_last, alpha_last, alpha_weight, instance_id,
                            hit, params_map, cone_scale, blur_idx=None, patch_scale=1.0, density_scale=1.0,
                            density_reweighting=True, map_exr=False, composite_bkgd=False, bkgd_color=(1., 1., 1.),
                            instance_color=None, net_chunk=65536, dtype=F32, noise=None):
    """InstanceRenderer.evaluate_model (renderer.py:247-316) downstream of `instancer.get_model_input`
    (instancer.pyx:38-54), whose buffers are the arguments: rays_d_map/pts [n,S,3], t/dists/alpha_weight [n,S],
    color_last [n,1,3], alpha_last [n,1], instance_id [n,S] int32, hit [n] bool, params_map [n,S,P].
    `instance_color` [n_instances,3] switches on the false-colour mode (renderer.py:226-227, 309-310)."""
    n_rays, S = dists.shape
    rays_d_map = np.asarray(rays_d_map, dtype=dtype); pts = np.asarray(pts, dtype=dtype)
    t = np.asarray(t, dtype=dtype); dists = np.asarray(dists, dtype=dtype)
    params_map = np.asarray(params_map, dtype=dtype); cone_scale = np.asarray(cone_scale, dtype=dtype).reshape(n_rays, 1)
    idxs = np.nonzero(np.asarray(hit))[0]
    if idxs.shape[0] == 0:                                                       # :255-256
        return np.zeros((n_rays, 3), dtype), np.zeros((n_rays,), dtype)
    if blur_idx is not None:                                                     # :259-262
        blur_scale = cone_scale[..., None, :] * t[..., :, None] / dtype(patch_scale)
        params_map = np.concatenate([params_map[..., :blur_idx], params_map[..., blur_idx, None] * blur_scale,
                                     params_map[..., blur_idx + 1:]], axis=-1)
    g = lambda a: np.asarray(a)[idxs]                                            # :265-272
    rays_d_map, pts, dists, params_map = g(rays_d_map), g(pts), g(dists), g(params_map)
    color_last, alpha_last = g(np.asarray(color_last, dtype=dtype)), g(np.asarray(alpha_last, dtype=dtype))
    alpha_weight, instance_id = g(np.asarray(alpha_weight, dtype=dtype)), g(instance_id)
    pos_flat = pts.reshape(-1, 3); dirs_flat = rays_d_map.reshape(-1, 3)
    params_flat = params_map.reshape(pos_flat.shape[0], params_map.shape[-1])
    n_pts = pos_flat.shape[0]
    idxs_pts = np.nonzero(dists.reshape(-1) > 0)[0]                              # :284
    color = np.zeros((n_pts, 3), dtype); alpha = np.zeros((n_pts, 1), dtype)     # scatter_nd zeros :296-298 / :302-303
    if idxs_pts.shape[0] > 0:
        cs, as_ = [], []
        for i in range(0, idxs_pts.shape[0], net_chunk):                         # :290-293
            sl = idxs_pts[i:i + net_chunk]
            c, a = model_forward(weights, spec, pos_flat[sl], dirs_flat[sl], params_flat[sl], dtype)
            cs.append(c); as_.append(a)
        color[idxs_pts] = np.concatenate(cs, 0); alpha[idxs_pts] = np.concatenate(as_, 0)
        alpha = alpha.reshape(pts.shape[:-1])
        alpha = alpha * (alpha_weight if density_reweighting else dtype(1)) * dtype(density_scale)   # :300
    else:
        alpha = alpha.reshape(pts.shape[:-1])
    color = color.reshape(pts.shape)
    if instance_color is not None:                                               # :306-307
        color = np.asarray(instance_color, dtype=dtype)[instance_id.reshape(-1)].reshape(color.shape)
    cm, am = instance_map_model_output(color, color_last, alpha, alpha_last, dists, patch_scale, composite_bkgd,
                                       bkgd_color, map_exr, instance_color is not None, dtype,
                                       None if noise is None else np.asarray(noise)[idxs])   # `noise` [n_rays,S] like dists
    color_map = np.zeros((n_rays, 3), dtype); alpha_map = np.zeros((n_rays,), dtype)   # :313-314: culled rays stay 0,
    color_map[idxs] = cm; alpha_map[idxs] = am                                         # even with composite_bkgd
    return color_map, alpha_map


def render_image_rgba(pred, height, width):
    """logger.Logger.render_image packing only (logger.py:126): [H, W, 4] premultiplied RGBA."""
    return np.concatenate([pred["color_pred"].reshape(-1, 3), pred["alpha_pred"].reshape(-1, 1)], -1) \
             .reshape(height, width, 4)


def gaussian_kernel_1d(size: int, std: float, dtype=F32):
    """util.interpolate.gaussian_kernel (interpolate.py:68-76), 1-D factor before normalisation.  Note the
    reference shifts the taps by +0.5 when `size` is even (line 71)."""
    x = np.linspace(-(size - 1) / 2, (size - 1) / 2, size).astype(dtype) + dtype(.5 if size % 2 == 0 else 0)
    return np.exp(dtype(-.5) * (x / dtype(std)) ** 2).astype(dtype)


def filtered_downsample(img, downsampling_factor: int, std: float = .5, dtype=F32):
    """util.interpolate.filtered_downsample (interpolate.py:78-82): depthwise 2-D cross-correlation with the
    normalised gaussian, stride = factor, TensorFlow 'SAME' zero padding.  img [H,W,C] -> [ceil(H/f),ceil(W/f),C]."""
    img = np.asarray(img, dtype=dtype)
    f = int(downsampling_factor)
    K = int(f * std * 6)
    k1 = gaussian_kernel_1d(K, f * std, dtype)
    k2 = np.tensordot(k1, k1, axes=0)
    k2 = (k2 / np.sum(k2, dtype=dtype)).astype(dtype)
    H, W, C = img.shape
    oh, ow = -(-H // f), -(-W // f)
    ph = max((oh - 1) * f + K - H, 0); pw = max((ow - 1) * f + K - W, 0)
    pad = np.zeros((H + ph, W + pw, C), dtype)
    pad[ph // 2:ph // 2 + H, pw // 2:pw // 2 + W] = img
    out = np.zeros((oh, ow, C), dtype)
    for i in range(K):
        for j in range(K):
            out += k2[i, j] * pad[i:i + (oh - 1) * f + 1:f, j:j + (ow - 1) * f + 1:f]
    return out


def image_epilogue(rgba, downsampling_factor: int = 1, write_exr: bool = False, dtype=F32):
    """logger.Logger.render_image after the renderer (logger.py:128-137): optional filtered downsample, then
    (unless EXR output) premultiplied -> straight colour, rgb / (a + 1e-5)."""
    img = np.asarray(rgba, dtype=dtype)
    if downsampling_factor > 1:                                                  # :129-130
        img = filtered_downsample(img, downsampling_factor, dtype=dtype)
    if not write_exr:                                                            # :133-135
        img = np.concatenate([img[..., :3] / (img[..., 3:] + dtype(1e-5)), img[..., 3:]], axis=-1)
    return img


def to_uint8(img):
    """tf.image.convert_image_dtype(float32 -> uint8) as logger.write_image uses it (logger.py:144):
    saturating cast of x * 255.5 (truncation)."""
    x = np.asarray(img, dtype=F32) * F32(255.5)
    return np.clip(np.nan_to_num(x, nan=0.0), 0, 255).astype(np.uint8)


def rel_linf(out, ref) -> float:
    """The parity metric of BASELINE.json: max|out - ref| / max|ref|."""
    out = np.asarray(out, dtype=F64); ref = np.asarray(ref, dtype=F64)
    denom = float(np.max(np.abs(ref)))
    return float(np.max(np.abs(out - ref))) / (denom if denom > 0 else 1.0)
