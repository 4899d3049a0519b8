"""Seeded synthetic weights and rays.  The reference distributes no trained
models or datasets, so benches and parity tests run on these."""

from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

# (n_parameters, AABB b_0, b_1, camera position, camera angle, material parameters) per BASELINE config family
FAMILIES = {
    # configs/config_carpet_render.py:21-22,35-40,51-52,72 (pose #0 of the 5 on latitude u=.3)
    "carpet": dict(n_parameters=(1, 6), b_0=(-1.5, -1.5, -1.5), b_1=(1.5, 1.5, 1.5),
                   cam=(0.9165151389911681 * 6, 0.0, 0.4 * 6), angle=0.55, params=(1, 1, 1, .1, 0, 0, 1), blur_idx=None),
    # configs/config_grass_render.py:20-24,58-59,79
    "grass": dict(n_parameters=(1, 4), b_0=(-1.2, -1.2, -.1), b_1=(1.2, 1.2, 1.),
                  cam=(0.30614675 * 6, -0.73910363 * 6, 0.6 * 6), angle=0.5, params=(0, .33, 0, -.707, .707), blur_idx=None),
    # configs/config_fur_train.py:29-30,39-48,81 (no fur render config exists)
    "fur": dict(n_parameters=(1, 4), b_0=(-1.4, -1.1, -.3), b_1=(1.4, 1.5, 3.),
                cam=(.47 * 5, -.65 * 5, .6 * 5), angle=0.63, params=(1, .5, 0, -.707, .707), blur_idx=None),
    # configs/config_grass_filtered_train.py:29-30,39-48,81,102
    "grass_filtered": dict(n_parameters=(2, 3), b_0=(-2.5, -2.5, -1.), b_1=(2.5, 2.5, 2.5),
                           cam=(.47 * 5, -.65 * 5, .6 * 5), angle=0.63, params=(10, 0, 0, -.707, .707), blur_idx=0),
}


def synthetic_weights(layer_table: Sequence[Tuple[str, int, int]], seed: int = 0, dense_media: bool = False) -> np.ndarray:
    """Keras-style glorot-uniform kernels, biases U(-0.1, 0.1) (non-zero on purpose), as one flat
    float32 blob in `layer_table` order (= Keras `get_weights()` order).  The layers are DRAWN in the order
    model.py:104-123 creates them (alpha before feature), whatever the table's order, so a layer's values
    depend only on the seed and its name.  `dense_media` scales the alpha head x32 and shifts its bias by
    +0.5 so alpha spans (0, 1] and the composite scan saturates."""
    rng = np.random.default_rng(seed)
    table = list(layer_table)
    creation = [r for r in table if r[0].startswith("trunk")] + [r for r in table if r[0] == "alpha"] + \
               [r for r in table if not r[0].startswith("trunk") and r[0] != "alpha"]
    drawn = {}
    for name, i, o in creation:
        lim = np.sqrt(6.0 / (i + o))
        k = rng.uniform(-lim, lim, size=(i, o)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, size=o).astype(np.float32)
        if dense_media and name == "alpha":
            k = k * np.float32(32.0)
            b = b + np.float32(0.5)
        drawn[name] = [k.ravel(), b]
    return np.concatenate([a for name, _, _ in table for a in drawn[name]]).astype(np.float32)


def all_hit_rays(n_rays: int, b_0, b_1, cam, seed: int = 1):
    """Rays from `cam` aimed at uniform points inside the AABB: every ray hits, so N*S is exact
    (throughput batches).  Returns float32 rays_o [n,3], unit rays_d [n,3], t [n,2], cone_scale [n,1]."""
    rng = np.random.default_rng(seed)
    b_0 = np.asarray(b_0, dtype=np.float64); b_1 = np.asarray(b_1, dtype=np.float64)
    o = np.asarray(cam, dtype=np.float64)
    target = rng.uniform(b_0 + 0.05 * (b_1 - b_0), b_1 - 0.05 * (b_1 - b_0), size=(n_rays, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    ta, tb = (b_0 - o) * inv, (b_1 - o) * inv
    t0 = np.max(np.minimum(ta, tb), axis=-1); t1 = np.min(np.maximum(ta, tb), axis=-1)
    assert np.all(t0 < t1)
    rays_o = np.broadcast_to(o, d.shape).astype(np.float32)
    cone = rng.uniform(1e-4, 1e-3, size=(n_rays, 1)).astype(np.float32)
    return rays_o.copy(), d.astype(np.float32), np.stack([t0, t1], -1).astype(np.float32), cone


PATCH_BOX = ([-1.4, -1.2, -.1], [1.2, 1.2, 1.8])     # configs/config_carpet_render.py:83-84


def patch_sheet(grid: int, extent: float = 1.5, scale: float = 0.09):
    """A synthetic scene for the patch instancer in the shape of the shipped render configs (their meshes are LFS pointers):
    grid x grid patches on the sheet z = 0.08 sin(2x) cos(2y), |x|, |y| <= extent.  Returns the patch -> world transformations
    [grid^2,4,4] -- (tangent, bitangent, normal) * scale at the vertex, what DistributeInstancesOnMesh builds
    (instancer.cpp:360-366) -- and the sheet itself as a triangle mesh (vertices [grid^2,3], faces [2 (grid-1)^2, 3])."""
    xs = np.linspace(-extent, extent, grid)
    x, y = np.meshgrid(xs, xs, indexing="ij")
    z = 0.08 * np.sin(2 * x) * np.cos(2 * y)
    dzdx = 0.16 * np.cos(2 * x) * np.cos(2 * y); dzdy = -0.16 * np.sin(2 * x) * np.sin(2 * y)
    tr = np.zeros((grid, grid, 4, 4), np.float32)
    n = np.stack([-dzdx, -dzdy, np.ones_like(z)], -1); n /= np.linalg.norm(n, axis=-1, keepdims=True)
    tx = np.stack([np.ones_like(z), np.zeros_like(z), dzdx], -1); tx /= np.linalg.norm(tx, axis=-1, keepdims=True)
    bt = np.cross(n, tx)
    tr[..., :3, 0] = tx * scale; tr[..., :3, 1] = bt * scale; tr[..., :3, 2] = n * scale
    tr[..., :3, 3] = np.stack([x, y, z], -1); tr[..., 3, 3] = 1
    v = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange(grid * grid).reshape(grid, grid)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, d], -1)]).astype(np.int32)
    return tr.reshape(-1, 4, 4), v, f
