"""Where the cameras and the material parameters of a generated dataset come from (reference: data/sampler.py -- points of the unit cube --
and data/distribution.py -- their images on a sphere, in a box, or a list of constants; `network/dataset.py:198-229` draws a pose and a
parameter vector per view from them).  Every shipped render config and every training config's validation set is described this way:
`Sphere` over `Concat(Constant, Grid)` for a turntable on one latitude, `Constant` for fixed views, `Concat(Constant, Sphere ...)` for
parameter sweeps.

Pinned: `tests/golden/cameras_*.json` hold what the reference's own two modules (TensorFlow-free, run in the build container by
`oracle/gen_golden.py`) return for the shipped configs; `tests/test_data.py` asks this module for the same sequences, bit for bit.

One module for both halves.  The names configs use map like this (`util.remap_reference_config`):
    data.sampler.{Sampler, Independent, Constant, Grid, Stratified, Concat} -> {Counter, UniformPoints, FixedPoint, GridPoints, JitteredGridPoints, JoinedPoints}
    data.distribution.{Sphere, Hemisphere, AABB, Constant, Range, Concat}   -> {Sphere, Hemisphere, Box, Constants, Range, Joined}
Random draws come from numpy's GLOBAL stream, which `main.py:30` seeds from the config: a config that mixes random poses and random
parameters consumes it in the reference's order (pose first, then radius, then parameters: dataset.py:218-219).
"""

from __future__ import annotations

from math import ceil
from typing import Sequence, Union

import numpy as np

from . import util


# ---- points of the unit cube [0, 1)^d ------------------------------------------------------------------------------------
class Counter:
    """data.sampler.Sampler (sampler.py:7-21): how many points have been handed out (`idx`) of how many there are (`n`, negative: no end)
    in how many dimensions (`d`).  A call only counts."""

    def __init__(self, d: int = 1, n: int = -1, idx: int = 0) -> None:
        self.d, self.n, self.idx = d, n, idx

    def __call__(self):
        self.idx += 1

    def done(self) -> bool:
        return 0 <= self.n <= self.idx


class UniformPoints(Counter):
    """data.sampler.Independent (sampler.py:23-27): iid uniform."""

    def __call__(self) -> np.ndarray:
        self.idx += 1
        return np.random.rand(self.d)


class FixedPoint(Counter):
    """data.sampler.Constant (sampler.py:29-39): always `c` (a float fills all d coordinates).  Note its default n = 0, not -1."""

    def __init__(self, d: int = 1, n: int = 0, c: Union[float, Sequence[float]] = 0., idx: int = 0) -> None:
        super().__init__(d, n, idx)
        self.c = np.array([c] * d if isinstance(c, float) else c, dtype=float)

    def __call__(self) -> np.ndarray:
        self.idx += 1
        return self.c


class GridPoints(Counter):
    """data.sampler.Grid (sampler.py:41-60): point idx of a grid with ceil(n^(1/d)) cells a side, the first coordinate running fastest; the
    cells' lower corners, or their centres with `sample_center`.  (n < 0 has no grid: the reference raises on the complex root, so does this.)"""

    def __init__(self, d: int = 1, n: int = -1, idx: int = 0, sample_center: bool = False) -> None:
        super().__init__(d, n, idx)
        self.cells_per_d = ceil(self.n ** (1 / self.d))
        self.cell_size = 1 / self.cells_per_d
        self.sample_center = sample_center

    def corner(self) -> np.ndarray:
        side = self.cells_per_d
        cell = np.array([(self.idx // side ** axis) % side for axis in range(self.d)], dtype=float)
        return cell / side

    def __call__(self) -> np.ndarray:
        x = self.corner()
        if self.sample_center:
            x += self.cell_size / 2
        self.idx += 1
        return x


class JitteredGridPoints(GridPoints):
    """data.sampler.Stratified (sampler.py:62-65): a uniform point of cell idx.  (The reference's calls a method `sample` its parent does
    not have and cannot run; this is what it describes.)"""

    def __call__(self) -> np.ndarray:
        x = self.corner() + np.random.rand(self.d) * self.cell_size
        self.idx += 1
        return x


class JoinedPoints(Counter):
    """data.sampler.Concat (sampler.py:67-78): the coordinates of two samplers side by side, both told this one's n and idx."""

    def __init__(self, sampler_config_0: dict, sampler_config_1: dict, n: int = -1, idx: int = 0) -> None:
        self.sampler_0 = util.instantiate(dict(sampler_config_0, n=n, idx=idx))
        self.sampler_1 = util.instantiate(dict(sampler_config_1, n=n, idx=idx))
        super().__init__(self.sampler_0.d + self.sampler_1.d, n, idx)

    def __call__(self) -> np.ndarray:
        self.idx += 1
        return np.concatenate([self.sampler_0(), self.sampler_1()])


# ---- their images ----------------------------------------------------------------------------------------------------------
def _blend(x, lo, hi):
    return (1 - x) * lo + x * hi


class Sphere:
    """data.distribution.Sphere (distribution.py:11-21): the unit sphere by height and azimuth -- x[0] in [0, 1] runs over
    z = 1 - 2u from `u_range`, x[1] over the azimuth 2 pi v from `v_range` -- which is uniform in area for uniform x."""

    def __init__(self, sampler_config: dict = None, u_range=(0, 1.), v_range=(0, 1.)) -> None:
        self.sampler = util.instantiate(dict(sampler_config) if sampler_config is not None else {"module": "nerf_tex_amd.distributions.UniformPoints", "d": 2})
        u, v = np.array(u_range), np.array(v_range)
        self.lo = np.array([(1 - 2 * u)[0], (2 * np.pi * v)[0]])
        self.hi = np.array([(1 - 2 * u)[1], (2 * np.pi * v)[1]])

    def __call__(self) -> np.ndarray:
        z, phi = _blend(self.sampler(), self.lo, self.hi)
        return np.array([np.cos(phi) * np.sqrt(1 - z ** 2), np.sin(phi) * np.sqrt(1 - z ** 2), z])


def Hemisphere(axis: int = 2, **kwargs) -> Sphere:
    """data.distribution.Hemisphere (distribution.py:23-34): the half with a non-negative coordinate `axis`."""
    halves = {0: dict(v_range=[-.25, .25]), 1: dict(v_range=[0, .5]), 2: dict(u_range=[0, .5])}
    return Sphere(**{"u_range": [0, 1.], "v_range": [0, 1.], **halves.get(axis, {}), **kwargs})


class Box:
    """data.distribution.AABB (distribution.py:36-45): the box from b_0 to b_1 (floats or lists), linear in each coordinate."""

    def __init__(self, sampler_config: dict = None, b_0: Union[float, Sequence[float]] = 0., b_1: Union[float, Sequence[float]] = 1.) -> None:
        self.sampler = util.instantiate(dict(sampler_config) if sampler_config is not None else {"module": "nerf_tex_amd.distributions.UniformPoints", "d": 3})
        self.lo, self.hi = np.stack([b_0, b_1])

    def __call__(self) -> np.ndarray:
        return _blend(self.sampler(), self.lo, self.hi)


class Constants:
    """data.distribution.Constant (distribution.py:47-57): the rows of `constants` in turn, round and round."""

    def __init__(self, constants: list = [[0]]) -> None:
        self.sampler = Counter(n=len(constants))
        self.constants = np.array(constants)

    def __call__(self) -> np.ndarray:
        row = self.constants[self.sampler.idx % self.sampler.n]
        self.sampler()
        return row


def Range(n: int = 128, b_0: Union[float, Sequence[float]] = 0., b_1: Union[float, Sequence[float]] = 1.) -> Box:
    """data.distribution.Range (distribution.py:59-61): n equal steps from b_0 to b_1 -- a ONE-dimensional grid, so with list bounds every
    coordinate takes the same step."""
    return Box({"module": "nerf_tex_amd.distributions.GridPoints", "n": n}, b_0, b_1)


class Joined:
    """data.distribution.Concat (distribution.py:63-77): the vectors of two distributions end to end; as long as the longer of the two
    (-1 if either has no end)."""

    def __init__(self, distribution_config_0: dict, distribution_config_1: dict) -> None:
        self.distribution_0 = util.instantiate(distribution_config_0)
        self.distribution_1 = util.instantiate(distribution_config_1)
        n0, n1 = self.distribution_0.sampler.n, self.distribution_1.sampler.n
        self.sampler = Counter(n=-1 if -1 in (n0, n1) else max(n0, n1))

    def __call__(self) -> np.ndarray:
        self.sampler()
        return np.concatenate([self.distribution_0(), self.distribution_1()])
