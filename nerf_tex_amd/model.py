"""Model containers (reference: network/model.py).

`ParamNerf` / `Nerf` / `CoarseFine` keep the reference's signatures and return `{name: model}`
exactly like model.py:45,56,125, so `renderer_config.update(model)` (render.py:24) works unchanged.
The model object holds the weights in the reference's own layout (`tf.keras.Model.get_weights()`:
kernel[in,out], bias[out] per Dense layer in the order of `model.layers`, which a functional model sorts
by graph depth -- the alpha head is LAST, see `layer_table`) and, per device, a context of the HIP
library holding the packed image.  Calling it runs the fused MLP kernel (`ntx_mlp_forward`).
"""

from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np

from .layer import embedding_kind, n_freq_bands_of

KIND_PARAMNERF, KIND_NERF = 0, 1


class NerfModel:
    """Stand-in for the `tf.keras.Model` the reference builds (model.py:125)."""

    def __init__(self, kind: int, n_parameters: Sequence[int], n_pos: int, pos_freq: int, dir_freq: int,
                 param_freq: int, depth: int, width: int, skips: Sequence[int], color_depth: int, name: str,
                 pos_encoding: str = "fourier", param_depth: int = 0, param_width: int = 128) -> None:
        self.kind = kind
        self.param_depth = 0 if kind == KIND_NERF else int(param_depth)   # model.py:88-101: Dense(param_width, relu) on FF(params)
        self.param_width = int(param_width)
        self.pos_encoding = pos_encoding          # "fourier" | "ipe" (IntegratedPositionalEncoding on n_pos = 6)
        self.n_geo, self.n_app = (0, 0) if kind == KIND_NERF else (int(n_parameters[0]), int(n_parameters[1]))
        self.n_pos, self.pos_freq, self.dir_freq, self.param_freq = n_pos, pos_freq, dir_freq, param_freq
        self.depth, self.width, self.skips, self.color_depth = depth, width, tuple(skips), color_depth
        self.name = name
        self._ctx: Dict[int, int] = {}
        self._cap: Dict[int, int] = {}            # rays reserved per context beyond the library default
        self._blob = np.zeros(self.n_weight_floats(), dtype=np.float32)
        self.initialize()

    # ---- architecture ------------------------------------------------------------------
    @property
    def n_params(self) -> int:
        return self.n_geo + self.n_app

    def branch_dim(self, n: int) -> int:
        """width a parameter branch adds to pos_map / dir_map (model.py:89-93, 97-101): FF(params) itself, or the output of the
        last of the `param_depth` Dense(param_width) layers"""
        if n == 0:
            return 0
        return self.param_width if self.param_depth > 0 else n * (1 + 2 * self.param_freq)

    @property
    def pos_map_dim(self) -> int:
        emb = 6 * self.pos_freq if self.pos_encoding == "ipe" else self.n_pos * (1 + 2 * self.pos_freq)
        return emb + self.branch_dim(self.n_geo)

    @property
    def dir_map_dim(self) -> int:
        return 3 * (1 + 2 * self.dir_freq) + self.branch_dim(self.n_app)

    def layer_table(self) -> List[Tuple[str, int, int]]:
        """(name, in, out) per Dense layer in the order of `tf.keras.Model.layers` / `get_weights()` for the model
        model.py:125 / 45 builds.  A functional model sorts layers by graph depth (distance to an output), ties by the
        order a traversal from `outputs=[color_outputs, alpha_outputs]` meets them: the trunk, the feature layer, the
        colour layers, `color`, and only then `alpha` -- although `alpha` is CREATED before the feature layer
        (model.py:111 vs 114).  The checkpoint keys `layer_with_weights-k` (logger.py:30-39) count in the same order."""
        # With param_depth > 0 the rule also places the branch layers: every Dense layer in the order the traversal first meets it
        # (a layer before its inputs; concat([dir_map, feature]) reaches the appearance branch before the feature layer) with its
        # graph depth (concat nodes take a level), then by decreasing depth, ties in traversal order: the geometry branch sits above
        # the whole trunk, the appearance branch interleaves with the trunk layers of equal depth and comes first there.
        pd, pw, w = self.param_depth, self.param_width, self.width
        ff = lambda n: n * (1 + 2 * self.param_freq)
        cd = self.color_depth if self.kind == KIND_PARAMNERF else 0
        seq = [("color", w // 2, 3, 0), ("color_half", w if cd > 0 else w + self.dir_map_dim, w // 2, 1)]
        for i in reversed(range(cd)):
            seq.append((f"color_hidden{i}", w + self.dir_map_dim if i == 0 else w, w, 1 + cd - i))
        d = cd + 2                                          # concat([dir_map, feature])
        if pd > 0 and self.n_app > 0:                       # dir_map = concat([FF(dir), branch]) is a node at d + 1
            for i in reversed(range(pd)):
                seq.append((f"param_app{i}", ff(self.n_app) if i == 0 else pw, pw, d + 2 + (pd - 1 - i)))
        k_head = w + (self.pos_map_dim if (self.depth - 1) in self.skips else 0)
        d += 1
        seq.append(("feature", k_head, w, d))
        for i in reversed(range(self.depth)):
            d += 1 + (1 if i in self.skips else 0)          # the concat a skip puts behind layer i
            seq.append((f"trunk{i}", self.pos_map_dim if i == 0 else w + (self.pos_map_dim if (i - 1) in self.skips else 0), w, d))
        if pd > 0 and self.n_geo > 0:                       # pos_map = concat([FF(pos), branch]) at depth(trunk0) + 1
            for i in reversed(range(pd)):
                seq.append((f"param_geo{i}", ff(self.n_geo) if i == 0 else pw, pw, d + 2 + (pd - 1 - i)))
        seq.append(("alpha", k_head, 1, 0))
        order = sorted(range(len(seq)), key=lambda j: (-seq[j][3], j))
        return [seq[j][:3] for j in order]

    def summary(self, print_fn=print) -> None:
        """`tf.keras.Model.summary()` as far as it can be said without Keras (train.py:36): the Dense layers in `get_weights()` order with
        their kernel shapes and parameter counts, and the total."""
        rows = [(name, f"({i}, {o})", i * o + o) for name, i, o in self.layer_table()]
        print_fn(f'Model: "{self.name}"')
        print_fn(f"{'Layer (Dense)':<20}{'kernel':<16}{'Param #':>10}")
        for name, shape, n in rows:
            print_fn(f"{name:<20}{shape:<16}{n:>10,}")
        print_fn(f"Total params: {sum(r[2] for r in rows):,}")

    def n_weight_floats(self) -> int:
        return sum(i * o + o for _, i, o in self.layer_table())

    def macs_per_sample(self) -> int:
        return sum(i * o for _, i, o in self.layer_table())

    def desc(self):
        from . import _lib
        # `skips` (model.py:107: `if i in skips` for i in range(depth)): one index as it is, none = -1, several = NTX_SKIP_MASK | bits
        live = sorted({int(i) for i in self.skips if 0 <= int(i) < self.depth})
        if any(i >= 30 for i in live):
            raise _lib.NtxError(_lib.NTX_E_UNSUPPORTED, f"skips={self.skips}: layer indices above 29 have no encoding in ntx_model_desc")
        skip = -1 if not live else live[0] if len(live) == 1 else _lib.SKIP_MASK | sum(1 << i for i in live)
        kind = _lib.KIND_PARAMNERF_EX if self.kind == KIND_PARAMNERF and self.param_depth != 0 else self.kind
        return _lib.ModelDesc(kind, self.n_geo, self.n_app, self.n_pos, self.pos_freq, self.dir_freq,
                              self.param_freq, self.depth, self.width, skip, self.color_depth,
                              1 if self.pos_encoding == "ipe" else 0, self.param_depth, self.param_width)

    # ---- weights -----------------------------------------------------------------------
    def initialize(self) -> None:
        """Keras defaults (model.py uses none other): glorot_uniform kernels, zero biases.  Draws
        from numpy's global RNG, which `main.py:30` seeds from the config."""
        parts = []
        for _, i, o in self.layer_table():
            lim = np.sqrt(6.0 / (i + o))
            parts.append(np.random.uniform(-lim, lim, size=i * o).astype(np.float32))
            parts.append(np.zeros(o, dtype=np.float32))
        self.set_blob(np.concatenate(parts))

    def get_weights(self) -> List[np.ndarray]:
        out, p = [], 0
        for _, i, o in self.layer_table():
            out.append(self._blob[p:p + i * o].reshape(i, o).copy()); p += i * o
            out.append(self._blob[p:p + o].copy()); p += o
        return out

    def set_weights(self, weights: Sequence[np.ndarray]) -> None:
        table = self.layer_table()
        if len(weights) != 2 * len(table):
            raise ValueError(f"expected {2 * len(table)} arrays, got {len(weights)}")
        flat = []
        for (name, i, o), k, b in zip(table, weights[0::2], weights[1::2]):
            k = np.asarray(k, dtype=np.float32); b = np.asarray(b, dtype=np.float32)
            if k.shape != (i, o) or b.shape != (o,):
                raise ValueError(f"layer {name}: expected kernel {(i, o)} bias {(o,)}, got {k.shape} {b.shape}")
            flat += [k.ravel(), b.ravel()]
        self.set_blob(np.concatenate(flat))

    def get_blob(self) -> np.ndarray:
        return self._blob.copy()

    def set_blob(self, blob: np.ndarray) -> None:
        blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
        if blob.size != self.n_weight_floats():
            raise ValueError(f"blob has {blob.size} floats, model needs {self.n_weight_floats()}")
        self._blob = blob.copy()
        if self._ctx:
            from . import _lib
            for ctx in self._ctx.values():
                _lib.check(_lib.lib.ntx_set_weights(ctx, self._blob.ctypes.data_as(C.POINTER(C.c_float)), self._blob.size))

    def set_weights_from_trainer(self, trainer, sync_host: bool = False) -> None:
        """The weights a `nerf_tex_amd.train.Trainer` holds become this model's, WITHOUT leaving the GPU: the render context of the
        trainer's device remakes its weight image from the trainer's device memory in the current stream's order (`ntx_set_weights_device`) --
        the validation render inside the reference's training loop (logger.py:76-81).  The model's host copy (`get_weights`, contexts
        of other devices, the fp16x3 images) is NOT updated unless `sync_host`, which is `set_blob(trainer.weights())`."""
        import torch
        from . import _lib
        if sync_host or getattr(trainer, "_pad", None) is not None:               # (a narrow network trained inside the 256-wide one: its own weights through the host)
            self.set_blob(trainer.weights())
            return
        if trainer.n_weights != self.n_weight_floats():
            raise ValueError(f"trainer has {trainer.n_weights} weights, model needs {self.n_weight_floats()}")
        dev = torch.device("cuda", trainer.device)
        ptr = C.c_void_p()
        _lib.check(_lib.lib.ntx_trainer_device_weights(trainer._h, C.byref(ptr)))
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_set_weights_device(self.ctx(trainer.device), ptr, trainer.n_weights, torch.cuda.current_stream(dev).cuda_stream))

    # ---- device ------------------------------------------------------------------------
    def ctx(self, device_index: int) -> int:
        """Context of the HIP library for `device_index`, created on first use."""
        from . import _lib
        if device_index not in self._ctx:
            handle = C.c_void_p()
            desc = self.desc()
            _lib.check(_lib.lib.ntx_create(C.byref(desc), self._blob.ctypes.data_as(C.POINTER(C.c_float)),
                                           self._blob.size, device_index, C.byref(handle)))
            self._ctx[device_index] = handle.value
        return self._ctx[device_index]

    def reserve(self, device_index: int, n_rays: int) -> None:
        """Make the context of `device_index` accept `n_rays` rays per `ntx_render_rays` call (`ntx_reserve`: setup-time,
        synchronising; grows only)."""
        from . import _lib
        cap = self._cap.get(device_index, _lib.DEFAULT_MAX_RAYS)
        if n_rays > cap:
            _lib.check(_lib.lib.ntx_reserve(self.ctx(device_index), int(n_rays)))
            self._cap[device_index] = int(n_rays)

    def close(self) -> None:
        if self._ctx:
            from . import _lib
            for ctx in self._ctx.values():
                _lib.lib.ntx_destroy(ctx)
            self._ctx = {}
            self._cap = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, inputs, training: bool = False):
        """model((pos[M,n_pos], dirs[M,3], params[M,P]), training) -> (color[M,3], alpha[M,1])
        (renderer.py:161).  Inputs are CUDA(ROCm) float32 tensors."""
        import torch
        from . import _lib
        pos, dirs, params = inputs
        pos = pos.contiguous().float(); dirs = dirs.contiguous().float()
        m = pos.shape[0]
        if pos.shape[1] != self.n_pos:
            raise ValueError(f"pos must be [{m},{self.n_pos}], got {tuple(pos.shape)}")
        if self.n_params > 0:
            params = params.contiguous().float()
            if params.shape != (m, self.n_params):
                raise ValueError(f"params must be [{m},{self.n_params}], got {tuple(params.shape)}")
        dev = pos.device
        color = torch.empty((m, 3), device=dev, dtype=torch.float32)
        alpha = torch.empty((m, 1), device=dev, dtype=torch.float32)
        # `self.precision` ("float32" default, or "fp16x3"): arithmetic of the Dense layers for direct calls of the model
        flags = _lib.PRECISIONS[getattr(self, "precision", "float32")]
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_mlp_forward(self.ctx(dev.index or 0), pos.data_ptr(), dirs.data_ptr(),
                                                params.data_ptr() if self.n_params > 0 else None, m, flags,
                                                color.data_ptr(), alpha.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream))
        return color, alpha


def Nerf(pos_embedding, dir_embedding, depth: int = 8, width: int = 256, skips: list = [4], name: str = "model",
         **kwargs) -> dict:
    """network.model.Nerf (model.py:9-45)."""
    return {name: NerfModel(KIND_NERF, (0, 0), 3, n_freq_bands_of(pos_embedding), n_freq_bands_of(dir_embedding),
                            0, depth, width, skips, 0, name)}


def ParamNerf(pos_embedding, dir_embedding, param_embedding, n_parameters: Union[int, list], n_pos: int = 3,
              param_depth: int = 0, param_width: int = 128, depth: int = 8, width: int = 256, skips: list = [4],
              color_depth: int = 1, embedding_config=None, include_param_dims: bool = False,
              name: str = "model") -> dict:
    """network.model.ParamNerf (model.py:58-125)."""
    if isinstance(n_parameters, int):
        n_parameters = [n_parameters, 0]                                    # model.py:63-64
    if embedding_config is not None:
        # (the reference cannot build such a model either: network/layer.py has no embedding layer for util.instantiate to find)
        raise NotImplementedError("embedding_config: the reference ships no embedding layer (network/layer.py) and there is no HIP kernel for one")
    return {name: NerfModel(KIND_PARAMNERF, n_parameters, n_pos, n_freq_bands_of(pos_embedding),
                            n_freq_bands_of(dir_embedding), n_freq_bands_of(param_embedding), depth, width, skips,
                            color_depth, name, embedding_kind(pos_embedding), param_depth=param_depth, param_width=param_width)}


def CoarseFine(model_config, **kwargs) -> dict:
    """network.model.CoarseFine (model.py:47-56)."""
    from . import util
    for key, value in kwargs.items():
        model_config.setdefault(key, value)
    model_coarse = util.instantiate(model_config)
    model_config["name"] = next(iter(model_coarse)) + "_fine"
    model_fine = util.instantiate(model_config)
    return dict(model_coarse, **model_fine)
