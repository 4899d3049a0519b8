"""Volumetric ray marcher (reference: network/renderer.py, class Renderer).

Same constructor kwargs and call contract as the reference (renderer.py:34, 47); the whole of
`__call__` -> `render_rays` -> `evaluate_model` -> `map_model_output` (renderer.py:47-213) is ONE
launch of the fused HIP kernel (`ntx_render_rays`): culling of t == inf rays, sample placement,
positional encoding, the 8x256 MLP on the matrix cores and the per-ray composite scan.
Inputs/outputs are torch tensors on the GPU instead of tf.Tensors.
"""

from __future__ import annotations

from typing import Optional

from . import _lib


class Renderer:
    """network.renderer.Renderer (renderer.py:31-213)."""

    def __init__(self, model, model_fine=None, n_samples: int = 64, n_importance: int = 0, perturb: bool = True,
                 raw_noise_std: float = 0, render_chunk: int = 32768, net_chunk: int = 65536,
                 downsampling_factor: int = 1, blur_idx: Optional[int] = None, map_exr: bool = False,
                 check_numerics: bool = True, precision: str = "float32", **kwargs) -> None:
        self.model = model
        self.model_fine = model_fine
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.perturb = perturb
        self.raw_noise_std = raw_noise_std
        # render_chunk / net_chunk bound TensorFlow's activation memory (renderer.py:72,160); the fused
        # kernel keeps activations in registers, so they are accepted and have no effect on results.
        self.render_chunk = render_chunk
        self.net_chunk = net_chunk
        self.downsampling_factor = downsampling_factor
        self.blur_idx = blur_idx
        self.map_exr = map_exr
        self.check_numerics = check_numerics     # tf.debugging.check_numerics, renderer.py:140-141
        # arithmetic of the Dense layers (include/nerftex.h: ntx_precision).  "float32" is what the reference computes in;
        # "fp16x3" is an opt-in 3-term split into IEEE halves on the 16-bit matrix cores: float32-grade (2.8e-6 from the float32
        # kernel on the bench image) but not bit-identical, and limited to |activation| <= 65504
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        self.precision = precision
        if raw_noise_std < 0:
            raise ValueError("raw_noise_std must be >= 0")
        # The reference draws jitter and noise from TensorFlow's generator and leaves numpy's alone; so does this class: the
        # per-call seeds come from a private generator, seeded ONCE (lazily, on the first draw) from numpy's global state
        # (main.py:30 seeds it from the config), so data.distribution / data.sampler see the same np.random stream as under
        # the reference however many images are rendered in between.
        self._seed_rng = None

    def __call__(self, rays_o, rays_d, t, parameters, cone_scale, composite_bkgd: bool = False,
                 bkgd_color=[1, 1, 1.], training: bool = True, z_vals=None, **kwargs) -> dict:
        """rays_o/rays_d [B,HW,3], t [B,HW,2], parameters [B,P], cone_scale [B,HW,1]
        -> {'color_pred': [B,HW,3], 'alpha_pred': [B,HW]}  (renderer.py:47-90).
        Beyond the reference: `seed` (int) fixes the jitter / noise stream of the call; `ray_index` = (index0, run_length,
        run_stride) maps the call's rays to the global ray indices that key the generators (a rank of a `dist.ShardMap` passes
        `shard.ray_index(rank)`), so that a sharded or chunked image draws exactly what the whole image draws."""
        import torch
        dev = rays_o.device
        if dev.type != "cuda":
            raise _lib.NtxError(_lib.NTX_E_NODEVICE, "Renderer inputs must live on the GPU; there is no CPU path")
        B, HW = rays_o.shape[0], rays_o.shape[1]
        n = B * HW
        rays_o = rays_o.reshape(n, 3).contiguous().float()
        rays_d = rays_d.reshape(n, 3).contiguous().float()
        t = t.reshape(n, 2).contiguous().float()
        cone = cone_scale.reshape(n).contiguous().float()
        mip = getattr(self.model, "pos_encoding", "fourier") == "ipe"
        if mip != isinstance(self, MipRenderer):
            raise _lib.NtxError(_lib.NTX_E_UNSUPPORTED, "IPE models go with MipRenderer, FourierFeatures models with Renderer")
        P = self.model.n_params + (1 if mip else 0)        # mip: the row still holds the blur parameter (renderer.py:385-386)
        params = None
        if P > 0:
            params = parameters.reshape(B, -1).contiguous().float().to(dev)
            if params.shape[1] != P:
                raise ValueError(f"parameters must be [B,{P}], got {tuple(parameters.shape)}")
        S = self.n_samples
        n_z = S + 1 if mip else S                          # mip: S+1 segment edges (renderer.py:374)
        flags = (_lib.FLAG_MAP_EXR if self.map_exr else 0) | (_lib.FLAG_COMPOSITE_BKGD if composite_bkgd else 0)
        flags |= _lib.PRECISIONS[self.precision]
        z = None
        seed = 0
        if z_vals is not None:
            z = z_vals.reshape(n, n_z).contiguous().float()
        elif self.perturb:
            # renderer.py:106-111 / 379-383: the stratified jitter is evaluated inside the kernel (counter-based
            # generator, include/nerftex.h: ntx_sample_depths); one seed per call, drawn like everything else in
            # the reference from numpy's global generator (main.py:30 seeds it), or given by the caller
            flags |= _lib.FLAG_PERTURB
        noise = float(self.raw_noise_std)
        if noise > 0:                                      # renderer.py:190-192, inside the kernel (one N(0,1) per sample)
            flags |= _lib.FLAG_RAW_NOISE
        if flags & (_lib.FLAG_PERTURB | _lib.FLAG_RAW_NOISE):
            seed = int(kwargs["seed"]) if kwargs.get("seed") is not None else self._next_seed()
        opts = None
        if noise > 0 or kwargs.get("ray_index") is not None:
            opts = _lib.render_opts(noise, seed, kwargs.get("ray_index"))
        self._last_seed = seed
        status = None
        if self.check_numerics:
            flags |= _lib.FLAG_CHECK_NUMERICS
            status = torch.zeros(1, device=dev, dtype=torch.int32)
        bk = bkgd_color.detach().cpu().tolist() if hasattr(bkgd_color, "detach") else list(bkgd_color)
        stream = torch.cuda.current_stream(dev).cuda_stream
        blur = -1 if self.blur_idx is None else int(self.blur_idx)

        def launch(model, n_s, z_in, want_weights):
            color = torch.empty((n, 3), device=dev, dtype=torch.float32)
            alpha = torch.empty((n,), device=dev, dtype=torch.float32)
            wts = torch.empty((n, n_s), device=dev, dtype=torch.float32) if want_weights else None
            model.reserve(dev.index or 0, n)               # setup-time; a no-op once the context has seen this size
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.ntx_render_rays(
                    model.ctx(dev.index or 0), rays_o.data_ptr(), rays_d.data_ptr(), t.data_ptr(),
                    params.data_ptr() if params is not None else None, HW, cone.data_ptr(), n, n_s, blur, flags,
                    _lib.f3(bk), z_in.data_ptr() if z_in is not None else None, seed, opts, color.data_ptr(), alpha.data_ptr(),
                    wts.data_ptr() if wts is not None else None, status.data_ptr() if status is not None else None, stream))
            return color, alpha, wts

        color, alpha, wts = launch(self.model, S, z, self.n_importance > 0)
        out = {"color_pred": color.reshape(B, HW, 3), "alpha_pred": alpha.reshape(B, HW)}
        if self.n_importance > 0:                                            # renderer.py:125-138
            NI = self.n_importance
            # `det=self.perturb` (renderer.py:128): perturb=True -> deterministic u, else uniform draws
            u = None if self.perturb else torch.rand((n, NI), device=dev, dtype=torch.float32)
            if kwargs.get("u") is not None:
                u = kwargs["u"].reshape(n, NI).contiguous().float()
            z_all = torch.empty((n, S + NI), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.ntx_sample_pdf(t.data_ptr(), z.data_ptr() if z is not None else None, wts.data_ptr(),
                                                   u.data_ptr() if u is not None else None, n, S, NI,
                                                   flags & _lib.FLAG_PERTURB, seed, opts, z_all.data_ptr(), stream))
            model_imp = self.model if self.model_fine is None else self.model_fine
            c2, a2, _ = launch(model_imp, S + NI, z_all, False)
            out = {"color_pred": c2.reshape(B, HW, 3), "alpha_pred": a2.reshape(B, HW),
                   "color_pred_coarse": out["color_pred"], "alpha_pred_coarse": out["alpha_pred"]}
            self._last_z, self._last_weights = z_all, wts       # (tests: the sampler's input and output)
        if status is not None:
            self._status = status          # read lazily: `raise_if_nonfinite()` syncs
        return out

    def raise_if_nonfinite(self) -> None:
        """The reference raises InvalidArgumentError inside render_rays (renderer.py:140-141); here the
        kernel sets a device flag and this (synchronising) call turns it into an exception."""
        st = getattr(self, "_status", None)
        if st is not None and int(st.item()) != 0:
            raise FloatingPointError("NaN or Inf encountered in color_pred/alpha_pred")

    def _next_seed(self) -> int:
        import numpy as np
        if self._seed_rng is None:
            # np.random.get_state() reads the global generator without advancing it
            key = np.random.get_state()[1]
            self._seed_rng = np.random.Generator(np.random.Philox(key=[int(key[0]) | (int(key[1]) << 32), int(key[2]) | (int(key[3]) << 32)]))
        return int(self._seed_rng.integers(0, 2 ** 63 - 1, dtype=np.int64))

    @staticmethod
    def sample_depths(t, n_points: int, perturb: bool = False, seed: int = 0, ray_index=None):
        """z_vals of renderer.py:101-111 on their own (`ntx_sample_depths`): t [n,2] -> [n, n_points], exactly the
        depths the fused kernel places for the same (perturb, seed, ray_index)."""
        import torch
        t = t.reshape(-1, 2).contiguous().float()
        n = t.shape[0]
        z = torch.empty((n, n_points), device=t.device, dtype=torch.float32)
        with torch.cuda.device(t.device):
            _lib.check(_lib.lib.ntx_sample_depths(t.data_ptr(), n, n_points, _lib.FLAG_PERTURB if perturb else 0, int(seed),
                                                  _lib.render_opts(ray_index=ray_index) if ray_index is not None else None,
                                                  z.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream))
        return z

    def map_model_output(self, color, alpha, z_vals, rays_d, composite_bkgd: bool, bkgd_color):
        """Renderer.map_model_output (renderer.py:170-213) on its own (`ntx_composite`):
        color [n,S,3], alpha [n,S], z_vals [n,S], rays_d [n,3] -> (color_map, alpha_map, weights)."""
        import torch
        dev = color.device
        n, S = alpha.shape
        color = color.contiguous().float(); alpha = alpha.contiguous().float()
        z_vals = z_vals.contiguous().float(); rays_d = rays_d.contiguous().float()
        c_out = torch.empty((n, 3), device=dev, dtype=torch.float32)
        a_out = torch.empty((n,), device=dev, dtype=torch.float32)
        w_out = torch.empty((n, S), device=dev, dtype=torch.float32)
        flags = (_lib.FLAG_MAP_EXR if self.map_exr else 0) | (_lib.FLAG_COMPOSITE_BKGD if composite_bkgd else 0)
        bk = bkgd_color.detach().cpu().tolist() if hasattr(bkgd_color, "detach") else list(bkgd_color)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.ntx_composite(color.data_ptr(), alpha.data_ptr(), z_vals.data_ptr(), rays_d.data_ptr(),
                                              n, S, flags, _lib.f3(bk), c_out.data_ptr(), a_out.data_ptr(),
                                              w_out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        return c_out, a_out, w_out


class MipRenderer(Renderer):
    """network.renderer.MipRenderer (renderer.py:356-473): cone-segment gaussians + integrated positional encoding.
    `model` must be an IPE ParamNerf (pos_embedding = IntegratedPositionalEncoding, n_pos = 6); `parameters` keep
    the blur parameter at `blur_idx` (it becomes the cone radius and is spliced out before the model).  Everything
    runs in the same fused kernel (`ntx_render_rays`) as the base renderer."""

    def __init__(self, blur_idx: int = None, **kwargs):
        if blur_idx is None:
            raise ValueError("MipRenderer needs blur_idx (renderer.py:385 indexes the parameters with it)")
        if kwargs.get("n_importance", 0) > 0:
            raise NotImplementedError("Importance sampling for mip-NeRF style rendering is not implemented in the reference either (renderer.py:403-404)")
        super().__init__(blur_idx=blur_idx, **kwargs)
        self.blur_idx_mip = blur_idx


class InstanceRenderer(Renderer):
    """network.renderer.InstanceRenderer (renderer.py:215-354): the renderer the shipped render configs use.

    Everything downstream of `instancer.get_model_input(rays_o, rays_d, parameters, n_samples, step_size)`
    (instancer.pyx:38-54) is one launch of `ntx_render_instanced` per render chunk.  `instancer` is any object with
    that method: this package's `nerf_tex_amd.instancer.Instancer` (the patch instancer on the GPU: the chunk then
    never leaves HBM), the reference's Cython `Instancer` on Embree, or a stand-in, returning, as numpy arrays
    or tensors: rays_d_map [n,S,3], pts [n,S,3], t [n,S], dists [n,S], color_last [n,1,3], alpha_last [n,1],
    alpha_weight [n,S], instance_id [n,S] int32, idxs (indices of the hit rays, `tf.where(hit)`-shaped [k,1]
    or a bool mask [n]), params_map [n,S,P]."""

    def __init__(self, instancer_config=None, step_size: float = 0.002, density_scale: float = 1,
                 density_reweighting: bool = True, false_color: bool = False, instancer=None, patch_scale=None,
                 **kwargs) -> None:
        kwargs.setdefault("perturb", False)       # the instancer, not the renderer, places the samples
        super().__init__(**kwargs)
        if instancer is None:
            from . import util
            instancer = util.instantiate(instancer_config)
        self.instancer = instancer
        self.step_size = step_size
        self.density_scale = density_scale
        self.density_reweighting = density_reweighting
        self.false_color = false_color
        self.instance_color = None
        if false_color:                                                       # renderer.py:226-227
            import numpy as np
            self.instance_color = np.random.uniform(size=(self.instancer.n_instances(), 3)).astype("float32")
        self.patch_scale = float(patch_scale if patch_scale is not None else instancer_config["patch_scale"])   # :228

    def __call__(self, rays_o, rays_d, t, parameters, cone_scale, composite_bkgd: bool = False,
                 bkgd_color=[1, 1, 1.], training: bool = False, **kwargs) -> dict:
        import numpy as np
        import torch
        assert training is False, "InstanceRenderer can only be used for evaluation (renderer.py:233)"
        dev = rays_o.device
        B, HW = rays_o.shape[0], rays_o.shape[1]
        n = B * HW
        o_f = rays_o.reshape(n, 3).float(); d_f = rays_d.reshape(n, 3).float(); t_f = t.reshape(n, 2).float()
        c_f = cone_scale.reshape(n).float()
        p_f = parameters.reshape(B, -1).float().repeat_interleave(HW, dim=0)                    # renderer.py:54
        keep = (t_f[:, 0] != float("inf")).nonzero(as_tuple=False)[:, 0]                        # renderer.py:58
        color = torch.zeros((n, 3), device=dev, dtype=torch.float32)
        alpha = torch.zeros((n,), device=dev, dtype=torch.float32)
        bk = bkgd_color.detach().cpu().tolist() if hasattr(bkgd_color, "detach") else list(bkgd_color)
        flags = (_lib.FLAG_MAP_EXR if self.map_exr else 0) | (_lib.FLAG_COMPOSITE_BKGD if composite_bkgd else 0)
        flags |= _lib.PRECISIONS[self.precision]
        status = None
        if self.check_numerics:
            flags |= _lib.FLAG_CHECK_NUMERICS
            status = torch.zeros(1, device=dev, dtype=torch.int32)
        inst_col = None
        if self.instance_color is not None:
            inst_col = torch.as_tensor(self.instance_color, device=dev)
        noise = float(self.raw_noise_std)                                                       # renderer.py:335-337
        noise_seed = 0
        if noise > 0:
            flags |= _lib.FLAG_RAW_NOISE
            noise_seed = int(kwargs["seed"]) if kwargs.get("seed") is not None else self._next_seed()
        self._last_seed = noise_seed
        S = self.n_samples

        def up(a, dt=torch.float32):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=dt).contiguous()
            return torch.as_tensor(np.ascontiguousarray(np.asarray(a)), device=dev).to(dt).contiguous()

        # an instancer of this package (nerf_tex_amd.instancer.Instancer) takes and returns tensors on the GPU: the ten buffers
        # never leave HBM.  Any other object with the reference's get_model_input is fed numpy arrays as the reference feeds it.
        native = bool(getattr(self.instancer, "device_native", False))
        for i in range(0, keep.shape[0], self.render_chunk):                                    # renderer.py:72-73
            sl = keep[i:i + self.render_chunk]
            k = sl.shape[0]
            ro_c, rd_c, p_c = o_f[sl], d_f[sl], p_f[sl]
            if native:
                (rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map) = \
                    self.instancer.get_model_input(ro_c, rd_c, p_c, S, self.step_size, seed=kwargs.get("instancer_seed"),
                                                   ray_index=(int(kwargs.get("hit_base", 0)) + i, k, k), sparse=kwargs.get("instancer_sparse", True),
                                                   fill=kwargs.get("instancer_fill"))
                # (sparse: the tail reads a row of the buffers only where dists > 0, renderer.py:284-288 -- the defaults behind a ray's last
                # step, three quarters of a carpet frame's bytes, are not written)
                hit = self.instancer.last_hit
            else:
                (rays_d_map, pts, tt, dists, color_last, alpha_last, alpha_weight, instance_id, idxs, params_map) = \
                    self.instancer.get_model_input(ro_c.cpu().numpy(), rd_c.cpu().numpy(), p_c.cpu().numpy(), S, self.step_size)
                idxs = np.asarray(idxs)
                hit = np.zeros(k, dtype=np.uint8)
                if idxs.dtype == np.bool_:
                    hit[idxs.reshape(-1)] = 1
                else:
                    hit[idxs.reshape(-1).astype(np.int64)] = 1
            bufs = dict(rays_d_map=up(rays_d_map), pts=up(pts), t=up(tt), dists=up(dists),
                        color_last=up(color_last).reshape(k, 3), alpha_last=up(alpha_last).reshape(k),
                        alpha_weight=up(alpha_weight) if self.density_reweighting else None,
                        instance_id=up(instance_id, torch.int32), hit=up(hit, torch.uint8), params_map=up(params_map))
            c_c = c_f[sl].contiguous()
            col_c = torch.empty((k, 3), device=dev, dtype=torch.float32)
            al_c = torch.empty((k,), device=dev, dtype=torch.float32)
            ptr = lambda x: x.data_ptr() if x is not None else None
            self.model.reserve(dev.index or 0, k)          # setup-time; a no-op once the context has seen this size
            with torch.cuda.device(dev):
                _lib.check(_lib.lib.ntx_render_instanced(
                    self.model.ctx(dev.index or 0), ptr(bufs["rays_d_map"]), ptr(bufs["pts"]), ptr(bufs["t"]),
                    ptr(bufs["dists"]), ptr(bufs["color_last"]), ptr(bufs["alpha_last"]), ptr(bufs["alpha_weight"]),
                    ptr(bufs["instance_id"]), ptr(bufs["hit"]), ptr(bufs["params_map"]) if bufs["params_map"].numel() else None,
                    ptr(c_c), k, S, -1 if self.blur_idx is None else int(self.blur_idx), self.patch_scale,
                    float(self.density_scale), flags, _lib.f3(bk), ptr(inst_col),
                    _lib.render_opts(noise, noise_seed, (i, k, k)) if noise > 0 else None,   # chunk i.. of the hit rays keys the draws
                    ptr(col_c), ptr(al_c), ptr(status),
                    torch.cuda.current_stream(dev).cuda_stream))
            color[sl] = col_c                                                                  # scatter_nd, renderer.py:83
            alpha[sl] = al_c
        if composite_bkgd:                                                                     # renderer.py:85-86
            miss = torch.ones(n, dtype=torch.bool, device=dev); miss[keep] = False
            color[miss] += torch.as_tensor(bk, device=dev, dtype=torch.float32)
        if status is not None:
            self._status = status
        return {"color_pred": color.reshape(B, HW, 3), "alpha_pred": alpha.reshape(B, HW)}


class MipInstanceRenderer(InstanceRenderer):
    """network.renderer.MipInstanceRenderer (renderer.py:475-587): the InstanceRenderer tail with an IPE model fed
    (sample point, cone covariance from t / dists / blur * cone_scale / patch_scale).  `params_map` rows keep the blur
    parameter at `blur_idx`."""

    def __init__(self, blur_idx: int = None, **kwargs):
        if blur_idx is None:
            raise ValueError("MipInstanceRenderer needs blur_idx (renderer.py:511)")
        super().__init__(blur_idx=blur_idx, **kwargs)
        self.blur_idx_mip = blur_idx
        if getattr(self.model, "pos_encoding", "fourier") != "ipe":
            raise _lib.NtxError(_lib.NTX_E_UNSUPPORTED, "MipInstanceRenderer needs an IPE model")
