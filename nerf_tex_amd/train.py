"""The training step of the reference on the GPU (network/train.py:49-70): `Trainer` owns the weights, Adam's state and every layer's
activations behind the C ABI (`ntx_trainer_*`, include/nerftex.h); `step(...)` is the body of the reference's loop -- forward under a
tape, loss, gradients, `optimizer.apply_gradients`.

    trainer = Trainer(model, max_rays=1024, n_samples=256, lrate=5e-4, lrate_decay=500)       # config_carpet_train.py:100-109
    loss = trainer.step(rays_o, rays_d, t, parameters, cone_scale, color, alpha, loss_fn)      # one iteration of train.py:61-67

`Train(...)` is the reference's function of that name (train.py:7-70): the datasets (`nerf_tex_amd.dataset`; or any iterable of batch
dicts), model, loss, schedule and renderer from the config blocks, the loop, and what its Logger does every i_img / i_checkpoint steps (logger.py:57-86) -- the validation
views rendered through the inference path with the trainer's weights handed over on the device, checkpoints in TensorFlow's bundle format
with model + step + optimizer (train.py:55-57) that `Trainer.restore` resumes from.

Built for the ParamNerf architecture of the shipped training configs (8 x 256, skips [4], color_depth 1; narrower widths inside it)."""

from __future__ import annotations

import ctypes as C
from typing import Optional

from . import _lib


class Trainer:
    def __init__(self, model, max_rays: int, n_samples: int, lrate: float = 5e-4, lrate_decay: float = 0, perturb: bool = True, blur_idx: Optional[int] = None,
                 map_exr: bool = False, raw_noise_std: float = 0.0, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7, device: int = 0) -> None:
        """`model`: a nerf_tex_amd.model.ParamNerf container (its blob gives the initial weights); `lrate`, `lrate_decay` as train.py:49-50
        (ExponentialDecay(lrate, decay_steps=lrate_decay * 1e3, decay_rate=0.1) when lrate_decay > 0); `perturb`, `blur_idx`, `map_exr`: the
        renderer's (renderer.py:34); `raw_noise_std`: the density regulariser of map_model_output (renderer.py:190-192; config_grass_filtered_train.py:99
        trains with 0.1), drawn per (seed, ray, sample) like the jitter."""
        import numpy as np
        self.model = model
        self.device = int(device)
        self.n_samples, self.max_rays = int(n_samples), int(max_rays)
        self.lrate, self.lrate_decay = float(lrate), float(lrate_decay)
        self.perturb, self.blur_idx, self.map_exr = bool(perturb), blur_idx, bool(map_exr)
        self.raw_noise_std = float(raw_noise_std)
        if self.raw_noise_std < 0:
            raise ValueError("raw_noise_std must be >= 0")
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        blob = np.ascontiguousarray(model.get_blob(), dtype=np.float32)
        self._h = C.c_void_p()
        self._pad = None
        wide = _widened(model)
        if wide is not None:                                                    # a narrower network trains inside the 256-wide one (see _widened)
            self._pad = _narrow_in_wide(model, wide)
            full = np.zeros(wide.n_weight_floats(), np.float32)
            full[self._pad] = blob
            blob = full
        desc = (wide or model).desc()
        _lib.check(_lib.lib.ntx_trainer_create(C.byref(desc), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, self.device, self.max_rays,
                                               self.n_samples, C.byref(self._h)))
        self._n_native = int(_lib.lib.ntx_trainer_weight_count(self._h))
        self.n_weights = self._n_native if self._pad is None else int(self._pad.size)
        self._calls = 0
        self._last_rays = 0

    @classmethod
    def from_config(cls, config: dict, max_rays: Optional[int] = None, device: int = 0, weights=None):
        """The trainer and the loss a reference TRAINING config asks for, from its own blocks as written (train.py:20-52): `model_config`
        (network.model.ParamNerf ...), `loss_config`, `lrate`, `lrate_decay`, `renderer_config` (n_samples, perturb, raw_noise_std, blur_idx,
        map_exr; render_chunk / net_chunk / downsampling_factor have no meaning here: a step is one batch).  `max_rays` defaults to the config's batch --
        `train_dataset_config.batchsize` images x `pixel_sampler_config.n_samples` rays -- when the config says it.  Returns (trainer, loss).
        The data side (`nerf_tex_amd.dataset`) and the loop around the step are `Train`'s."""
        from . import util
        cfg = util.remap_reference_config(config)
        models = util.instantiate(dict(cfg["model_config"]))                     # {'model': ...} or CoarseFine's {'model': ..., 'model_fine': ...}
        model = models["model"] if "model" in models else next(iter(models.values()))
        if weights is not None:
            model.set_weights(weights) if isinstance(weights, (list, tuple)) else model.set_blob(weights)
        loss = util.instantiate(dict(cfg["loss_config"]))
        r = dict(cfg["renderer_config"])
        for k in ("module", "render_chunk", "net_chunk", "downsampling_factor"):
            r.pop(k, None)
        n_samples = int(r.pop("n_samples", 64))                                   # renderer.py:34 default
        known = {k: r.pop(k) for k in ("perturb", "raw_noise_std", "blur_idx", "map_exr") if k in r}
        n_importance = int(r.pop("n_importance", 0))
        if r:
            raise TypeError(f"renderer_config keys without a meaning in a training step: {sorted(r)}")
        if max_rays is None:
            ds = cfg.get("train_dataset_config") or {}
            max_rays = int(cfg.get("batchsize", ds.get("batchsize", 1))) * int(cfg.get("rays_per_image", (ds.get("pixel_sampler_config") or {}).get("n_samples", 1024)))
        kw = dict(max_rays=max_rays, n_samples=n_samples, lrate=cfg.get("lrate", 5e-4), lrate_decay=cfg.get("lrate_decay", 0), device=device, **known)
        if n_importance > 0:                                                    # renderer.py:125-138: a coarse and a fine pass
            return CoarseFineTrainer(model, models.get(model.name + "_fine"), n_importance=n_importance, **kw), loss
        return cls(model, **kw), loss

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None and getattr(_lib, "lib", None) is not None:
            _lib.lib.ntx_trainer_destroy(h)
            self._h = None

    def _vector(self, what: int):
        import numpy as np
        out = np.empty(self._n_native, np.float32)
        _lib.check(_lib.lib.ntx_trainer_get(self._h, what, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out if self._pad is None else np.ascontiguousarray(out[self._pad])

    def weights(self):
        """The weights as one float32 vector in `get_weights()` order (hand it to `model.set_blob` to render with them)."""
        return self._vector(_lib.TRAINER_WEIGHTS)

    def gradients(self):
        return self._vector(_lib.TRAINER_GRADIENTS)

    def adam_state(self):
        return self._vector(_lib.TRAINER_ADAM_M), self._vector(_lib.TRAINER_ADAM_V)

    def activation(self, layer: int, n_samples_total: int):
        """What the last step kept of layer 0-7 (trunk), 8 / 9 (colour layers), 10 (raw density), 11 (raw colour), or -- 20-27, 28, 29 -- the
        gradient at the outputs of the trunk layers, the first colour layer, the feature layer, 30 the composite's adjoint (dL/d raw colour,
        dL/d raw density): [n_samples_total, width] float32 (tests)."""
        import numpy as np
        width = {9: 128, 10: 1, 11: 3, 30: 4}.get(int(layer), 256)
        out = np.empty((int(n_samples_total), width), np.float32)
        _lib.check(_lib.lib.ntx_trainer_activation(self._h, int(layer), int(n_samples_total), out.ctypes.data_as(C.POINTER(C.c_float))))
        if self._pad is not None and width > 4:                                 # the narrow network's own columns
            out = np.ascontiguousarray(out[:, :self.model.width // (2 if layer == 9 else 1)])
        return out

    def set_weights(self, blob) -> None:
        """New weights (waits for what is in flight on the device); Adam's moments and iteration count stay as they are."""
        self._set(_lib.TRAINER_WEIGHTS, blob)

    def _set(self, what: int, values) -> None:
        import numpy as np
        b = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        if self._pad is not None:
            if b.size != self._pad.size:
                raise ValueError(f"{b.size} floats given, the model has {self._pad.size}")
            full = np.zeros(self._n_native, np.float32)
            full[self._pad] = b
            b = full
        _lib.check(_lib.lib.ntx_trainer_set(self._h, what, b.ctypes.data_as(C.POINTER(C.c_float)), b.size))

    @property
    def iterations(self) -> int:
        return int(_lib.lib.ntx_trainer_iterations(self._h))

    # ---- resuming (train.py:55-60, logger.py:30-39, 84-86) -------------------------------------------------------------------------
    def state_dict(self) -> dict:
        """Everything a resumed run needs to continue bit for bit: weights, Adam's moments and iteration count (its bias correction and
        the ExponentialDecay schedule run on it), and the counter the default jitter / noise seeds come from."""
        m, v = self.adam_state()
        return {"weights": self.weights(), "adam_m": m, "adam_v": v, "iterations": self.iterations, "calls": self._calls}

    def load_state_dict(self, state: dict) -> None:
        self._set(_lib.TRAINER_WEIGHTS, state["weights"])
        self._set(_lib.TRAINER_ADAM_M, state["adam_m"]); self._set(_lib.TRAINER_ADAM_V, state["adam_v"])
        _lib.check(_lib.lib.ntx_trainer_set_iterations(self._h, int(state["iterations"])))
        self._calls = int(state.get("calls", state["iterations"]))

    def save(self, prefix: str, step: int = None, root: str = None) -> str:
        """`checkpoint_manager.save(checkpoint_number=step)` (logger.py:84-86): a TensorBundle `<prefix>.index` / `.data-00000-of-00001` with
        the keys of `tf.train.Checkpoint(**{model.name: model}, step=step, optimizer=optimizer)` (train.py:55-57; nerf_tex_amd/checkpoint.py)."""
        import numpy as np
        from . import checkpoint
        table = self.model.layer_table()
        split = lambda blob: _split_blob(table, blob)
        st = self.state_dict()
        hyper = {"beta_1": self.beta_1, "beta_2": self.beta_2, "decay": 0.0}
        if not self.lrate_decay > 0:
            hyper["learning_rate"] = self.lrate                  # (under a schedule Keras has no such variable)
        return checkpoint.write_checkpoint(prefix, table, split(st["weights"]), split(st["adam_m"]), split(st["adam_v"]), iterations=st["iterations"],
                                           step=st["iterations"] if step is None else int(step), hyper=hyper, root=root or self.model.name)

    def restore(self, path: str, root: str = None, verify: bool = True) -> dict:
        """`checkpoint.restore(manager.latest_checkpoint)` (logger.py:39): `path` is a checkpoint prefix or a directory of `ckpt-<n>`.  Weights,
        Adam's slots and iteration count when the bundle holds them (a bundle with weights alone restarts the optimiser, as TensorFlow's
        `expect_partial()` restore would).  Returns {'prefix', 'step', 'iterations'}."""
        import os
        import numpy as np
        from . import checkpoint
        prefix = checkpoint.latest_checkpoint(path) if os.path.isdir(path) else path
        st = checkpoint.training_state_from_bundle(checkpoint.read_bundle(prefix, verify), self.model.layer_table(), root or self.model.name)
        flat = lambda arrs: np.concatenate([np.asarray(a, np.float32).ravel() for a in arrs])
        self._set(_lib.TRAINER_WEIGHTS, flat(st["weights"]))
        zeros = np.zeros(self.n_weights, np.float32)
        self._set(_lib.TRAINER_ADAM_M, zeros if st["m"] is None else flat(st["m"])); self._set(_lib.TRAINER_ADAM_V, zeros if st["v"] is None else flat(st["v"]))
        it = st["iterations"] if st["iterations"] is not None and st["m"] is not None else 0
        _lib.check(_lib.lib.ntx_trainer_set_iterations(self._h, int(it)))
        self._calls = int(it)
        return {"prefix": prefix, "step": st["step"], "iterations": it}

    def gradients_step(self, rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true, loss, composite_bkgd: bool = False, bkgd_color=(1., 1., 1.),
                       seed: Optional[int] = None, z_vals=None, rays_per_param_row: int = 1, n_samples: Optional[int] = None):
        """Forward + loss + gradients (train.py:61-66): rays_o / rays_d [N,3], t [N,2] (inf for a ray that misses the proxy: it predicts 0 / the
        background and counts in the loss, renderer.py:58-86), parameters [rows,P],
        cone_scale [N] or [N,1], color_true [N,3], alpha_true [N]; `loss`: a nerf_tex_amd.loss object.  Returns (loss [1], color_pred [N,3],
        alpha_pred [N]) as GPU tensors; the gradients stay in the trainer.  `z_vals` [N, n_samples]: given sample depths (`n_samples` of them,
        default the trainer's) instead of the ones placed between t."""
        import torch
        dev = torch.device("cuda", self.device)
        to = lambda a: None if a is None else (a if isinstance(a, torch.Tensor) else torch.as_tensor(a)).to(device=dev, dtype=torch.float32).contiguous()
        rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true, z_vals = (to(a) for a in (rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true, z_vals))
        n = rays_o.reshape(-1, 3).shape[0]
        self._last_rays = n
        flags = (_lib.FLAG_PERTURB if self.perturb else 0) | (_lib.FLAG_MAP_EXR if self.map_exr else 0) | (_lib.FLAG_COMPOSITE_BKGD if composite_bkgd else 0)
        opts = None
        if self.raw_noise_std > 0:
            flags |= _lib.FLAG_RAW_NOISE
            opts = C.byref(_lib.render_opts(raw_noise_std=self.raw_noise_std))
        if seed is None:
            seed = self._calls
        self._calls += 1
        color = torch.empty((n, 3), device=dev); alpha = torch.empty((n,), device=dev); val = torch.empty((1,), device=dev)
        desc = loss.desc()
        ptr = lambda x: x.data_ptr() if x is not None and x.numel() else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_train_step_gradients(
                self._h, ptr(rays_o), ptr(rays_d), ptr(t), ptr(parameters), int(rays_per_param_row), ptr(cone_scale), n, int(n_samples or self.n_samples),
                -1 if self.blur_idx is None else int(self.blur_idx), flags, _lib.f3(bkgd_color), int(seed) & (2 ** 64 - 1), opts, ptr(z_vals), ptr(color_true), ptr(alpha_true),
                C.byref(desc), ptr(color), ptr(alpha), ptr(val), torch.cuda.current_stream(dev).cuda_stream))
        return val, color, alpha

    def apply_gradients(self) -> None:
        """optimizer.apply_gradients (train.py:67): Adam under the learning-rate schedule of train.py:49-52."""
        import torch
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.ntx_trainer_adam_step(self._h, self.lrate, self.lrate_decay * 1e3 if self.lrate_decay > 0 else 0.0, 0.1, self.beta_1, self.beta_2,
                                                      self.epsilon, torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream))

    def train_step(self, data: dict, loss, composite_bkgd: bool = False, bkgd_color=(1., 1., 1.), comm=None, group=None, seed: Optional[int] = None) -> dict:
        """The body of the reference's loop on ITS batch dict (train.py:61-67: `pred = renderer(**data, composite_bkgd=..., bkgd_color=...)`,
        `loss_fn(color_true=data['color'], alpha_true=data['alpha'], **pred)`, gradients, `apply_gradients`): `data` as network/dataset.py maps it
        -- rays_o / rays_d [B,R,3], t [B,R,2], cone_scale [B,R,1], parameters [B,P] (one row per image: renderer.py:54), color [B,R,3],
        alpha [B,R].  Returns {'loss', 'color_pred' [B,R,3], 'alpha_pred' [B,R]} (GPU tensors)."""
        import torch
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
        ro = as_t(data["rays_o"]); B, R = int(ro.shape[0]), int(ro.shape[1])
        flat = lambda k, w: as_t(data[k]).reshape(B * R, w) if w else as_t(data[k]).reshape(B * R)
        alpha = flat("alpha", 0) if data.get("alpha") is not None else None
        val, c, a = self.gradients_step(flat("rays_o", 3), flat("rays_d", 3), flat("t", 2), as_t(data["parameters"]).reshape(B, -1), flat("cone_scale", 0),
                                        flat("color", 3), alpha, loss, composite_bkgd=composite_bkgd, bkgd_color=bkgd_color, seed=seed, rays_per_param_row=R)
        if comm is not None or (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1):
            self.sync_gradients(comm, group)
        self.apply_gradients()
        return {"loss": val, "color_pred": c.reshape(B, R, 3), "alpha_pred": a.reshape(B, R)}

    def sync_gradients(self, comm=None, group=None) -> None:
        """Data-parallel training (one process per GPU, every rank on its own rays -- equally many): the gradient becomes the mean over the
        ranks, which is the gradient of the loss over all their rays (the losses of loss.py are means over rays).  `comm`: a
        nerf_tex_amd.dist.Comm -- ONE ncclAllReduce of the 2.7 MB over xGMI behind the C ABI (`ntx_trainer_allreduce_gradients`), on the current
        stream.  Without one, the same mean through torch.distributed on host memory (gloo: the CPU-side tests, ranks sharing a GPU)."""
        import torch
        import torch.distributed as dist
        on_device = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 and not (on_device and getattr(self, "_shards_agreed", False)):
            # the mean of the ranks' means is the batch's gradient only when every rank brought as many rays (the losses are means over rays).
            # Over gloo (host tensors) this is asked every step; over nccl it costs a device sync, so it is asked at the first step only.
            self._shards_agreed = True
            mine = torch.tensor([self._last_rays], dtype=torch.int64, device=torch.device("cuda", self.device) if on_device else "cpu")
            counts = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
            dist.all_gather(counts, mine, group=group)
            if len({int(c.item()) for c in counts}) != 1:
                raise ValueError(f"data-parallel step with unequal shards ({[int(c.item()) for c in counts]} rays): the mean of per-rank gradients would be wrong")
        if comm is not None:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib.ntx_trainer_allreduce_gradients(self._h, comm.handle, torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream))
            return
        g = allreduce_mean_host(self.gradients(), group)
        _lib.check(_lib.lib.ntx_trainer_set(self._h, _lib.TRAINER_GRADIENTS, g.ctypes.data_as(C.POINTER(C.c_float)), g.size))

    def step(self, *args, comm=None, group=None, **kwargs):
        """One iteration of the reference's loop (train.py:61-67); returns the loss (a GPU tensor; with several ranks: this rank's).
        `comm` / `group`: data-parallel (see `sync_gradients`)."""
        import torch.distributed as dist
        val, _, _ = self.gradients_step(*args, **kwargs)
        if comm is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            self.sync_gradients(comm, group)
        self.apply_gradients()
        return val


class CoarseFineTrainer:
    """Training with `n_importance > 0` (renderer.py:125-138, loss.py:15-16, 41-47, model.py:47-56): a coarse pass on `n_samples` depths, the
    importance sampler on its composite weights (sample_pdf, no gradient through it: `tf.stop_gradient`, :129), a fine pass on the
    n_samples + n_importance merged depths by `model_fine` -- or by the same network when there is none (:132) -- and the loss of both
    passes added up.  Two networks take one step each, on their own gradients; one network takes one step on the sum of the two passes'.
    Every piece is the C ABI's: two `ntx_train_step_gradients` (the coarse one also leaves the weights: `ntx_trainer_composite_weights`),
    `ntx_sample_pdf` between them, `ntx_trainer_stash_gradients` for the shared network."""

    def __init__(self, model, model_fine=None, max_rays: int = 1024, n_samples: int = 64, n_importance: int = 64, **kw) -> None:
        self.n_samples, self.n_importance, self.shared = int(n_samples), int(n_importance), model_fine is None
        if self.n_importance < 1:
            raise ValueError("n_importance must be >= 1 (a plain Trainer otherwise)")
        total = self.n_samples + self.n_importance
        self.fine = Trainer(model if self.shared else model_fine, max_rays=max_rays, n_samples=total, **kw)
        self.coarse = self.fine if self.shared else Trainer(model, max_rays=max_rays, n_samples=self.n_samples, **kw)
        self.model, self.model_fine, self.device, self.perturb = model, model_fine, self.fine.device, self.fine.perturb
        self._calls = 0

    @property
    def trainers(self):
        return (self.fine,) if self.shared else (self.coarse, self.fine)

    @property
    def iterations(self) -> int:
        return self.fine.iterations

    def gradients_step(self, rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true, loss, composite_bkgd: bool = False, bkgd_color=(1., 1., 1.),
                       seed: Optional[int] = None, rays_per_param_row: int = 1, u=None, on_coarse=None):
        """Both passes and their gradients.  Returns (loss [1] = fine + coarse, color_pred, alpha_pred, color_pred_coarse, alpha_pred_coarse).
        `u` [N, n_importance]: the sampler's uniform draws when perturb is off (renderer.py:128 `det=self.perturb`: with perturb they are
        tf.linspace); default: torch's generator under `seed`.  `on_coarse()`: called between the passes (tests read the coarse activations)."""
        import torch
        dev = torch.device("cuda", self.device)
        to = lambda a: None if a is None else (a if isinstance(a, torch.Tensor) else torch.as_tensor(a)).to(device=dev, dtype=torch.float32).contiguous()
        rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true = (to(a) for a in (rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true))
        n, S, NI = rays_o.reshape(-1, 3).shape[0], self.n_samples, self.n_importance
        if seed is None:
            seed = self._calls
        self._calls += 1
        wts = torch.empty((n, S), device=dev)
        _lib.check(_lib.lib.ntx_trainer_composite_weights(self.coarse._h, wts.data_ptr()))
        try:
            val_c, cc, ac = self.coarse.gradients_step(rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true, loss, composite_bkgd=composite_bkgd,
                                                       bkgd_color=bkgd_color, seed=seed, rays_per_param_row=rays_per_param_row, n_samples=S)
        finally:
            _lib.check(_lib.lib.ntx_trainer_composite_weights(self.coarse._h, None))
        if on_coarse is not None:
            on_coarse()
        if not self.perturb and u is None:
            u = torch.rand((n, NI), device=dev, generator=torch.Generator(device=dev).manual_seed(int(seed) & (2 ** 63 - 1)))
        u = None if self.perturb else to(u).reshape(n, NI).contiguous()
        z_all = torch.empty((n, S + NI), device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_sample_pdf(t.data_ptr(), None, wts.data_ptr(), u.data_ptr() if u is not None else None, n, S, NI,
                                               _lib.FLAG_PERTURB if self.perturb else 0, int(seed) & (2 ** 64 - 1), None, z_all.data_ptr(), stream))
            if self.shared:
                _lib.check(_lib.lib.ntx_trainer_stash_gradients(self.fine._h, 0, stream))
        val_f, cf, af = self.fine.gradients_step(rays_o, rays_d, t, parameters, cone_scale, color_true, alpha_true, loss, composite_bkgd=composite_bkgd,
                                                 bkgd_color=bkgd_color, seed=seed, z_vals=z_all, rays_per_param_row=rays_per_param_row, n_samples=S + NI)
        if self.shared:
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.ntx_trainer_stash_gradients(self.fine._h, 1, stream))
        self.last_z = z_all
        return val_f + val_c, cf, af, cc, ac

    def apply_gradients(self) -> None:
        for tr in self.trainers:
            tr.apply_gradients()

    # ---- resuming: ONE checkpoint holds both networks under one optimizer (train.py:55: dict(model, step=..., optimizer=...)) --------------
    def save(self, prefix: str, step: int = None) -> str:
        from . import checkpoint
        parts = []
        for tr in self.trainers:
            st, table = tr.state_dict(), tr.model.layer_table()
            parts.append((tr.model.name, table, _split_blob(table, st["weights"]), _split_blob(table, st["adam_m"]), _split_blob(table, st["adam_v"])))
        f = self.fine
        hyper = {"beta_1": f.beta_1, "beta_2": f.beta_2, "decay": 0.0}
        if not f.lrate_decay > 0:
            hyper["learning_rate"] = f.lrate
        root, table, w, m, v = parts[0]
        return checkpoint.write_checkpoint(prefix, table, w, m, v, iterations=self.iterations, step=self.iterations if step is None else int(step), hyper=hyper,
                                           root=root, more=parts[1:])

    def restore(self, path: str, verify: bool = True) -> dict:
        """As `Trainer.restore`, every network from its own root of the one bundle."""
        import os
        from . import checkpoint
        prefix = checkpoint.latest_checkpoint(path) if os.path.isdir(path) else path
        info = {}
        for tr in self.trainers:
            info = tr.restore(prefix, root=tr.model.name, verify=verify)
        self._calls = int(info.get("iterations") or 0)
        return info

    def hand_weights_to_models(self) -> None:
        """Both networks' render contexts take the trainers' weights on the device (`NerfModel.set_weights_from_trainer`)."""
        for tr in self.trainers:
            tr.model.set_weights_from_trainer(tr)

    def step(self, *args, **kwargs):
        val = self.gradients_step(*args, **kwargs)[0]
        self.apply_gradients()
        return val

    def train_step(self, data: dict, loss, composite_bkgd: bool = False, bkgd_color=(1., 1., 1.), seed: Optional[int] = None) -> dict:
        """`Trainer.train_step` on the reference's batch dict; the predictions carry the coarse pass's too, as Renderer.render_rays returns them (:138)."""
        import torch
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
        ro = as_t(data["rays_o"]); B, R = int(ro.shape[0]), int(ro.shape[1])
        flat = lambda k, w: as_t(data[k]).reshape(B * R, w) if w else as_t(data[k]).reshape(B * R)
        alpha = flat("alpha", 0) if data.get("alpha") is not None else None
        val, c, a, cc, ac = self.gradients_step(flat("rays_o", 3), flat("rays_d", 3), flat("t", 2), as_t(data["parameters"]).reshape(B, -1), flat("cone_scale", 0),
                                                flat("color", 3), alpha, loss, composite_bkgd=composite_bkgd, bkgd_color=bkgd_color, seed=seed, rays_per_param_row=R)
        self.apply_gradients()
        return {"loss": val, "color_pred": c.reshape(B, R, 3), "alpha_pred": a.reshape(B, R), "color_pred_coarse": cc.reshape(B, R, 3), "alpha_pred_coarse": ac.reshape(B, R)}


def _widened(model):
    """Training is built for 8 x 256.  A ParamNerf of the same shape but NARROWER (`width` < 256, even) trains inside it:
    its kernels sit in the top-left corners of the 256-wide ones (behind the encoding rows of the skip and colour layers), everything else is
    zero and STAYS zero -- a padded unit's pre-activation is exactly 0, its ReLU gate is shut, so no gradient reaches its incoming weights, and
    its outgoing weights see an activation of exactly 0; Adam leaves a weight whose gradient was always 0 where it is.  The step is then the
    narrow network's, value for value (adding exact zeros changes no float32 sum), at the 256-wide network's cost -- as the flex render
    family does.  Returns the 256-wide twin, or None if `model` is not such a network (256 itself included)."""
    from .model import KIND_PARAMNERF, NerfModel
    if not (model.kind == KIND_PARAMNERF and model.width < 256 and model.width >= 2 and model.width % 2 == 0 and model.depth == 8 and tuple(model.skips) == (4,)
            and model.color_depth == 1 and model.param_depth == 0 and model.pos_encoding == "fourier"):
        return None
    return NerfModel(model.kind, [model.n_geo, model.n_app], model.n_pos, model.pos_freq, model.dir_freq, model.param_freq, 8, 256, (4,), 1, model.name)


def _narrow_in_wide(narrow, wide):
    """Index of every float of `narrow`'s weight blob inside `wide`'s (both in `get_weights()` order): a layer's input rows are
    [encoding rows (the same in both) | hidden rows 0 .. w-1 of the wide layer's 256 (128 for the last colour layer)], its columns 0 .. out-1."""
    import numpy as np
    idx, at = [], 0
    for (name, i_n, o_n), (name_w, i_w, o_w) in zip(narrow.layer_table(), wide.layer_table()):
        assert name == name_w
        hidden_n = 0 if name == "trunk0" else narrow.width // 2 if name == "color" else narrow.width     # the rows that come from a hidden layer
        hidden_w = 0 if name == "trunk0" else 128 if name == "color" else 256
        assert i_n - hidden_n == i_w - hidden_w and i_n >= hidden_n, (name, i_n, i_w)                # the pos_map / dir_map rows in front of them (model.py:107, 115)
        rows = np.arange(i_n)                                                 # narrow row r -> wide row r: the hidden rows follow the encoding rows in both
        kernel = at + rows[:, None] * o_w + np.arange(o_n)[None, :]
        idx += [kernel.reshape(-1), at + i_w * o_w + np.arange(o_n)]
        at += i_w * o_w + o_w
    return np.concatenate(idx).astype(np.int64)


def allreduce_mean_host(values, group=None):
    """The mean over the ranks of a float32 vector, through torch.distributed on host memory (sum in the backend's order, then / world)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    v = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32).copy())
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
        v /= dist.get_world_size(group)
    return v.numpy()


def _split_blob(table, blob):
    import numpy as np
    out, p = [], 0
    b = np.asarray(blob, np.float32).ravel()
    for _, i, o in table:
        out.append(b[p:p + i * o].reshape(i, o)); p += i * o
        out.append(b[p:p + o]); p += o
    return out


def Train(target_path: str, train_dataset=None, val_dataset=None, model_config: dict = None, loss_config: dict = None, n_iters: int = 1000, lrate: float = 5e-4,
          lrate_decay: float = 0, renderer_config: dict = None, logger_config: dict = None, max_rays: int = None, device: int = 0, weights=None,
          composite_bkgd: bool = None, bkgd_color=None, train_dataset_config: dict = None, val_dataset_config: dict = None, **kwargs) -> dict:
    """network.train.Train (train.py:7-70).  The data side either as the reference's blocks -- `train_dataset_config` / `val_dataset_config`
    (`network.dataset.Dataset` over `TFRecord` / `FileFolder` / `GenerateData`, instantiated here as train.py:23-26 does, `n_parameters` of the
    model defaulting to the dataset's, :29) -- or handed in: `train_dataset` any iterable of batch dicts as network/dataset.py maps them
    (rays_o / rays_d [B,R,3], t [B,R,2], cone_scale [B,R,1], parameters [B,P], color [B,R,3], alpha [B,R]), `val_dataset` a
    `nerf_tex_amd.dataset.Dataset` (or any iterable of ray batch dicts with height / width / composite_bkgd / bkgd_color) whose views are
    rendered every `i_img` steps.  The other arguments are the reference's: `model_config` / `loss_config` /
    `renderer_config` blocks, `n_iters`, `lrate`, `lrate_decay`, and of `logger_config` (logger.py:14) `i_print`, `i_img`, `i_checkpoint`,
    `max_to_keep`.  What the reference's Logger does at its cadences happens here: a checkpoint `checkpoints/ckpt-<step>` under `target_path`
    (model + step + optimizer, train.py:55-57) every `i_checkpoint` steps, the validation views through `Renderer` -- the inference path, the
    trainer's weights handed over on the device -- every `i_img` steps; a run whose `target_path` holds checkpoints resumes from the newest and
    takes `n_iters - step` batches (train.py:60).  Returns {'trainer', 'renderer', 'loss' [(step, value)], 'images' {step: [RGBA [H,W,4]]},
    'checkpoints' [prefix], 'step'}."""
    import os
    import torch
    from . import checkpoint, util
    from .renderer import Renderer
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    if train_dataset is None:
        if train_dataset_config is None:
            raise TypeError("Train needs train_dataset_config (the reference's block) or train_dataset (an iterable of batches)")
        block = dict(util.remap_reference_config(train_dataset_config), device=torch.device("cuda", device))
        if world > 1 and block.get("seed") is not None:
            block["seed"] = int(block["seed"]) + rank                         # every rank its own views and pixels
        train_dataset = util.instantiate(block)
    if val_dataset is None and val_dataset_config is not None:
        val_dataset = util.instantiate(dict(util.remap_reference_config(val_dataset_config), device=torch.device("cuda", device)))
    if model_config is not None and hasattr(train_dataset, "n_parameters"):
        model_config = dict(model_config)
        model_config.setdefault("n_parameters", train_dataset.n_parameters)     # train.py:29
    cfg = util.remap_reference_config(dict(model_config=model_config, loss_config=loss_config, renderer_config=renderer_config or {}, lrate=lrate,
                                           lrate_decay=lrate_decay))
    if max_rays is None:
        if hasattr(train_dataset, "batchsize") and hasattr(train_dataset, "n_samples"):
            max_rays = int(train_dataset.batchsize) * int(train_dataset.n_samples)
        else:
            first = next(iter(train_dataset))
            max_rays = int(first["rays_o"].shape[0]) * int(first["rays_o"].shape[1])
    trainer, loss_fn = Trainer.from_config(cfg, max_rays=max_rays, device=device, weights=weights)
    model = trainer.model
    rcfg = {k: v for k, v in dict(cfg["renderer_config"]).items() if k != "module"}
    two = isinstance(trainer, CoarseFineTrainer)                  # n_importance > 0: render.py:24 hands the renderer every model of the dict
    if rank == 0 and (logger_config or {}).get("print_model_summary", True):
        for m in ([trainer.coarse.model, trainer.fine.model] if two and not trainer.shared else [model]):      # train.py:35-36 (plot_model needs graphviz: not drawn)
            m.summary()
    renderer = Renderer(model=model, model_fine=trainer.model_fine if two else None, **rcfg)
    lg = dict(i_summary=10, i_print=100, i_img=5e3, i_checkpoint=1e3, max_to_keep=3, keep_every_n_hours=12)
    lg.update({k: v for k, v in (logger_config or {}).items() if k in lg})
    i_print, i_img, i_ckpt, keep, i_summary = int(lg["i_print"]), int(lg["i_img"]), int(lg["i_checkpoint"]), int(lg["max_to_keep"]), int(lg["i_summary"])
    ckpt_dir = os.path.join((logger_config or {}).get("source_path") or target_path, "checkpoints")     # logger.py:15, 30
    os.makedirs(ckpt_dir, exist_ok=True)
    step = 0
    try:                                                          # logger.py:39: restore the newest checkpoint if there is one
        info = trainer.restore(ckpt_dir)
        step = int(info["step"] if info["step"] is not None else info["iterations"])
        print(f"Restored model & optimizer from {info['prefix']}.")
    except FileNotFoundError:
        pass
    dp_comm = None
    if world > 1:
        # Data parallel (one process per GPU; the reference has one): every rank on its own batches, the gradient their mean
        # (`Trainer.train_step` -> `sync_gradients`), so the ranks must start from ONE state -- rank 0's, restored or freshly drawn --
        # and only rank 0 keeps the Logger's files.
        if two:
            raise NotImplementedError("data-parallel coarse + fine training")
        state = [dict(trainer.state_dict(), step=step) if rank == 0 else None]
        with torch.cuda.device(device):
            dist.broadcast_object_list(state, src=0)
        trainer.load_state_dict({k: v for k, v in state[0].items() if k != "step"})
        step = int(state[0]["step"])
        if rank != 0:
            i_print = i_img = i_ckpt = i_summary = 0
        if dist.get_backend() == "nccl":                           # one ncclAllReduce of the gradients a step on the device
            from .dist import Comm
            dp_comm = Comm(device)
    cb = getattr(train_dataset, "composite_bkgd", False) if composite_bkgd is None else composite_bkgd
    bc = getattr(train_dataset, "bkgd_color", (1., 1., 1.)) if bkgd_color is None else bkgd_color
    # CheckpointManager keeps the list of checkpoints it still rotates in the directory's `checkpoint` file (all_model_checkpoint_paths): a
    # resumed run goes on rotating THOSE.  A checkpoint kept for good (keep_every_n_hours) is not in the list and is never deleted; without
    # the file, the newest max_to_keep found are taken as the list (never more: nothing this run cannot account for is removed).
    import re
    listed = [p for p in checkpoint.read_manager_state(ckpt_dir) if os.path.exists(p + ".index")]
    if listed:
        found = [(0, p) for p in listed]
    else:
        found = sorted((int(m.group(1)), os.path.join(ckpt_dir, f[:-6])) for f in os.listdir(ckpt_dir) for m in [re.fullmatch(r"ckpt-(\d+)\.index", f)] if m)
        found = found[-keep:] if keep > 0 else found
    out = {"trainer": trainer, "renderer": renderer, "loss": [], "images": {}, "checkpoints": [p for _, p in found] if rank == 0 else [], "step": step}
    if step >= n_iters:
        if dp_comm is not None:
            dp_comm.close()
        return out
    todo = int(n_iters) - step
    import time
    t_print = time.perf_counter()
    # logger.py:41-44, 60-64, 79-81: the loss every i_summary steps and the validation images as TensorBoard summaries in target_path.  The
    # loss stays on the device until the loop next waits for it anyway (a print, a checkpoint, the end): asking every 10 steps would stall it.
    writer, pending = None, []
    saved_at, kept_for_good = {}, [time.time()]
    if i_summary > 0:
        from .summary import FileWriter
        writer = FileWriter(target_path)

    def flush_summaries():
        for st, value, wall in pending:
            writer.scalar("Loss", float(value.item()), st, wall)
        pending.clear()
        writer.flush()

    for data in train_dataset:                                    # train.py:60: train_dataset.take(n_iters - logger.step)
        if todo <= 0:
            break
        todo -= 1
        pred = trainer.train_step(data, loss_fn, composite_bkgd=cb, bkgd_color=bc, **({"seed": step * world + rank, "comm": dp_comm} if world > 1 else {}))
        step += 1
        if writer is not None and step % i_summary == 0:
            pending.append((step, pred["loss"], time.time()))
            if len(pending) >= 256:
                flush_summaries()
        if i_print > 0 and step % i_print == 0:
            val = float(pred["loss"].item())
            out["loss"].append((step, val))
            now = time.perf_counter()
            print(f"Step {step} | Loss {val:.3g} | Duration {now - t_print:.3g}")      # logger.py:68-73
            t_print = now
        if val_dataset is not None and i_img > 0 and step % i_img == 0:      # logger.py:76-81
            from .render import render_image
            trainer.hand_weights_to_models() if two else model.set_weights_from_trainer(trainer)
            out["images"][step] = [render_image(renderer, val_dataset, view)[0] for view in val_dataset]
            from .render import write_images, util_format                 # logger.py:76-78: media/validation/<step padded to n_iters>/<view>.png
            vdir = os.path.join(target_path, "media", "validation", util_format(step, int(n_iters))[:-4])
            for k, im in enumerate(out["images"][step]):
                write_images(vdir, im[None], k, len(out["images"][step]), int((logger_config or {}).get("downsampling_factor", 1)),
                             bool((logger_config or {}).get("write_exr", False)))
            if writer is not None:
                from .render import image_epilogue
                flush_summaries()
                u8 = [image_epilogue(im, int((logger_config or {}).get("downsampling_factor", 1)), uint8=True)[1] for im in out["images"][step][:3]]
                writer.image("Validation Rendering", torch.stack(u8).cpu().numpy(), step)
        if i_ckpt > 0 and step % i_ckpt == 0:                               # logger.py:84-86
            out["checkpoints"].append(trainer.save(os.path.join(ckpt_dir, f"ckpt-{step}"), step=step))
            saved_at[out["checkpoints"][-1]] = time.time()
            for old in out["checkpoints"][:-keep] if keep > 0 else []:
                # CheckpointManager(keep_checkpoint_every_n_hours=...): one that falls out of the newest `max_to_keep` stays for good if it was
                # saved that long after the last one kept this way (counted from the start of the run)
                if saved_at.get(old, 0.0) - kept_for_good[0] >= 3600.0 * float(lg["keep_every_n_hours"]):
                    kept_for_good[0] = saved_at[old]
                    continue
                for suffix in (".index", ".data-00000-of-00001"):
                    if os.path.exists(old + suffix):
                        os.remove(old + suffix)
            out["checkpoints"] = out["checkpoints"][-keep:] if keep > 0 else out["checkpoints"]
            checkpoint.write_manager_state(ckpt_dir, out["checkpoints"])
            if writer is not None:
                flush_summaries()
    if writer is not None:
        flush_summaries()
        writer.close()
        out["events"] = writer.path
    out["step"] = step
    torch.cuda.synchronize(torch.device("cuda", device))
    if dp_comm is not None:
        dp_comm.close()
    return out
