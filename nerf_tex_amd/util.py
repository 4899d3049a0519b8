"""Config / plugin mechanism of the render path.

Mirrors the reference's `util/util.py:8-54` convention: a config is a plain dict whose `'module'`
key is a dotted path to a callable and whose other keys are its kwargs; consumers instantiate their
sub-configs recursively.  A reference config file runs on this package by replacing the
`network.` prefix of the hot-path modules with `nerf_tex_amd.` (see `remap_reference_config`).
"""

from __future__ import annotations

import copy
import importlib
from typing import Any


class EasyDict(dict):
    """dict with attribute access; nested dicts are converted on construction (util.py:8-28)."""

    def __init__(self, other: dict = None, **kw) -> None:
        super().__init__()
        src = dict(other or {}, **kw)
        for key, value in src.items():
            if isinstance(value, dict) and not isinstance(value, EasyDict):
                value = EasyDict(value)
            self[key] = value

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    def __delattr__(self, key: str) -> None:
        del self[key]


def get_attr_from_path(path: str) -> Any:
    module_name, _, attr_name = path.rpartition(".")
    return getattr(importlib.import_module(module_name), attr_name)


def instantiate(config: dict) -> Any:
    """`{'module': 'pkg.mod.Attr', **kwargs}` -> `Attr(**kwargs)`; None -> None (util.py:44-54).  A reference module name that has a
    drop-in here (`_REMAP` below) resolves to it, so nested blocks need no remapping of their own."""
    if config is None:
        return None
    args = EasyDict(config)
    module = _REMAP.get(args.module, args.module)
    del args["module"]
    return get_attr_from_path(module)(**args)


# Reference module paths on the hot path -> their MI355X drop-ins in this package.
_REMAP = {
    "network.renderer.Renderer": "nerf_tex_amd.renderer.Renderer",
    "network.model.ParamNerf": "nerf_tex_amd.model.ParamNerf",
    "network.model.Nerf": "nerf_tex_amd.model.Nerf",
    "network.model.CoarseFine": "nerf_tex_amd.model.CoarseFine",
    "network.model.FourierFeatures": "nerf_tex_amd.layer.FourierFeatures",
    "network.layer.FourierFeatures": "nerf_tex_amd.layer.FourierFeatures",
    "network.model.IntegratedPositionalEncoding": "nerf_tex_amd.layer.IntegratedPositionalEncoding",
    "network.layer.IntegratedPositionalEncoding": "nerf_tex_amd.layer.IntegratedPositionalEncoding",
    "network.renderer.InstanceRenderer": "nerf_tex_amd.renderer.InstanceRenderer",
    "network.renderer.MipRenderer": "nerf_tex_amd.renderer.MipRenderer",
    "network.renderer.MipInstanceRenderer": "nerf_tex_amd.renderer.MipInstanceRenderer",
    "network.ray_sampler.Proxy": "nerf_tex_amd.ray_sampler.Proxy",
    "network.ray_sampler.Frustum": "nerf_tex_amd.ray_sampler.Frustum",
    "network.proxy.AABB": "nerf_tex_amd.proxy.AABB",
    "network.pixel_sampler.Full": "nerf_tex_amd.pixel_sampler.Full",
    "network.pixel_sampler.Independent": "nerf_tex_amd.pixel_sampler.Independent",
    "network.pixel_sampler.Proxy": "nerf_tex_amd.pixel_sampler.Proxy",
    "network.train.Train": "nerf_tex_amd.train.Train",
    "network.dataset.TFRecord": "nerf_tex_amd.dataset.TFRecord",
    "network.dataset.FileFolder": "nerf_tex_amd.dataset.FileFolder",
    "network.render.Render": "nerf_tex_amd.render.Render",
    "network.dataset.Dataset": "nerf_tex_amd.dataset.Dataset",
    "network.dataset.GenerateData": "nerf_tex_amd.dataset.GenerateData",
    "instancer.instancer.Instancer": "nerf_tex_amd.instancer.Instancer",
    "data.sampler.Sampler": "nerf_tex_amd.distributions.Counter",
    "data.sampler.Independent": "nerf_tex_amd.distributions.UniformPoints",
    "data.sampler.Constant": "nerf_tex_amd.distributions.FixedPoint",
    "data.sampler.Grid": "nerf_tex_amd.distributions.GridPoints",
    "data.sampler.Stratified": "nerf_tex_amd.distributions.JitteredGridPoints",
    "data.sampler.Concat": "nerf_tex_amd.distributions.JoinedPoints",
    "data.distribution.Sphere": "nerf_tex_amd.distributions.Sphere",
    "data.distribution.Hemisphere": "nerf_tex_amd.distributions.Hemisphere",
    "data.distribution.AABB": "nerf_tex_amd.distributions.Box",
    "data.distribution.Constant": "nerf_tex_amd.distributions.Constants",
    "data.distribution.Range": "nerf_tex_amd.distributions.Range",
    "data.distribution.Concat": "nerf_tex_amd.distributions.Joined",
    "network.loss.NerfLoss": "nerf_tex_amd.loss.NerfLoss",
    "network.loss.AlphaLoss": "nerf_tex_amd.loss.AlphaLoss",
}


def remap_reference_config(config: dict) -> EasyDict:
    """Deep-copy a reference config dict, pointing every hot-path `'module'` at this package.
    Modules without a drop-in (Logger, losses ...) are left untouched."""

    def walk(node):
        if isinstance(node, dict):
            out = {k: walk(v) for k, v in node.items()}
            if isinstance(out.get("module"), str):
                out["module"] = _REMAP.get(out["module"], out["module"])
            return out
        if isinstance(node, list):
            return [walk(v) for v in node]
        return copy.deepcopy(node)

    return EasyDict(walk(config))
