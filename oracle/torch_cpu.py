"""torch-CPU float32 restatement of the render path, for bench.py's `cpu_baseline` ONLY.

TEST / MEASUREMENT INFRASTRUCTURE (same rules as nerftex_oracle.py: nothing under nerf_tex_amd/ imports it).
TensorFlow cannot run here or on the GPU box, so "the reference's TF-CPU path" is reported as this port: the same
operations TF 2.4 would dispatch on a CPU -- MatMul + BiasAdd + Relu per Dense layer through the host's BLAS, vectorised
sin/cos over the whole chunk, cumprod, reduce_sum -- with the reference's own chunking (render_chunk 32768 rays,
net_chunk 65536 samples: renderer.py:72-73, 160-163).  PARITY UNPINNED like the rest of oracle/; it is checked against
nerftex_oracle.py in tests/test_oracle.py.  Every function cites the file:line of /root/reference it follows.
"""

from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch


def fourier_features(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    """layer.FourierFeatures.call (layer.py:18-23): [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...]."""
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(out, -1)


def model_forward(w: Sequence[torch.Tensor], spec, pos, dirs, params):
    """ParamNerf / Nerf (model.py:58-125 / 9-45); `w` in Keras get_weights() order (nerftex_oracle.layer_table)."""
    g, a = spec.n_geo, spec.n_app
    pos_map = fourier_features(pos, spec.pos_freq)                                    # model.py:77
    dir_map = fourier_features(dirs, spec.dir_freq)                                   # model.py:78
    if g > 0:
        pos_map = torch.cat([pos_map, fourier_features(params[:, :g], spec.param_freq)], -1)       # :88-93
    if a > 0:
        dir_map = torch.cat([dir_map, fourier_features(params[:, g:g + a], spec.param_freq)], -1)  # :96-101
    it = iter(range(0, len(w) - 2, 2))
    h = pos_map
    for i in range(spec.depth):                                                       # :104-108
        j = next(it)
        h = torch.relu(torch.addmm(w[j + 1], h, w[j]))
        if i in spec.skips:
            h = torch.cat([pos_map, h], -1)
    alpha = torch.addmm(w[-1], h, w[-2])                                              # :111 (last in get_weights())
    j = next(it)
    h = torch.addmm(w[j + 1], h, w[j])                                                # :114
    h = torch.cat([dir_map, h], -1)                                                   # :115
    if spec.kind == "ParamNerf":
        for _ in range(spec.color_depth):                                             # :118-119
            j = next(it)
            h = torch.relu(torch.addmm(w[j + 1], h, w[j]))
    j = next(it)
    h = torch.relu(torch.addmm(w[j + 1], h, w[j]))                                    # :122
    j = next(it)
    return torch.addmm(w[j + 1], h, w[j]), alpha                                      # :123


def render_rays(w, spec, rays_o, rays_d, t, parameters, cone_scale, n_samples: int, blur_idx=None, net_chunk: int = 65536):
    """Renderer.render_rays + evaluate_model + map_model_output (renderer.py:92-213), perturb=False, no background."""
    n = rays_o.shape[0]
    rays_d_n = rays_d / torch.linalg.norm(rays_d, dim=-1, keepdim=True)               # :98
    t_vals = torch.linspace(0., 1., n_samples)                                        # :101
    z = t[:, None, 0] * (1 - t_vals) + t[:, None, 1] * t_vals                         # :102
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]                     # :114
    pos = pts.reshape(-1, 3)                                                          # :151
    dirs = rays_d_n.repeat_interleave(n_samples, 0)                                   # :152
    params = parameters.repeat_interleave(n_samples, 0)                               # :154
    if blur_idx is not None:                                                          # :155-158
        scale = (cone_scale[:, None, :] * z[:, :, None]).reshape(-1, 1)
        params = torch.cat([params[:, :blur_idx], params[:, blur_idx, None] * scale, params[:, blur_idx + 1:]], -1)
    color, alpha = [], []
    for i in range(0, pos.shape[0], net_chunk):                                       # :160-163
        c, a_ = model_forward(w, spec, pos[i:i + net_chunk], dirs[i:i + net_chunk], params[i:i + net_chunk])
        color.append(c); alpha.append(a_)
    color = torch.cat(color, 0).reshape(n, n_samples, 3); alpha = torch.cat(alpha, 0).reshape(n, n_samples)
    dists = z[:, 1:] - z[:, :-1]                                                      # :174
    dists = torch.cat([dists, dists[:, -1:]], -1) * torch.linalg.norm(rays_d, dim=-1, keepdim=True)   # :177, 180
    rgb = torch.sigmoid(color)                                                        # :187
    am = 1. - torch.exp(-torch.relu(alpha) * dists)                                   # :195
    trans = torch.cumprod(1. - am + 1e-10, -1)
    wts = am * torch.cat([torch.ones_like(trans[:, :1]), trans[:, :-1]], -1)          # :198 exclusive cumprod
    return torch.sum(wts[..., None] * rgb, -2), torch.sum(wts, -1)                    # :201, 207


def renderer_call(w_np: List[np.ndarray], spec, rays_o, rays_d, t, parameters_row, cone_scale, n_samples: int, blur_idx=None,
                  render_chunk: int = 32768, net_chunk: int = 65536):
    """Renderer.__call__ (renderer.py:47-90) for all-hit rays of ONE image: render_chunk loop + concat."""
    with torch.no_grad():
        w = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in w_np]
        ro, rd, tt, cs = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in (rays_o, rays_d, t, cone_scale))
        pr = torch.from_numpy(np.ascontiguousarray(parameters_row, dtype=np.float32)).reshape(1, -1).repeat(ro.shape[0], 1)   # :54
        cols, alps = [], []
        for i in range(0, ro.shape[0], render_chunk):                                 # :72-73
            c, a = render_rays(w, spec, ro[i:i + render_chunk], rd[i:i + render_chunk], tt[i:i + render_chunk],
                               pr[i:i + render_chunk], cs[i:i + render_chunk], n_samples, blur_idx, net_chunk)
            cols.append(c); alps.append(a)
        return torch.cat(cols, 0).numpy(), torch.cat(alps, 0).numpy()


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (cpu.max / cfs_quota_us) --
    on a container with 256 visible CPUs and a 16-CPU quota, 128 BLAS threads run several times slower than 16."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def blas_backend() -> str:
    cfg = torch.__config__.show()
    keys = [ln.strip() for ln in cfg.splitlines() if any(k in ln for k in ("BLAS_INFO", "LAPACK_INFO", "USE_MKL=", "USE_MKLDNN=", "MKL ", "oneAPI", "OpenBLAS"))]
    info = [p for ln in keys for p in ln.replace(",", " ").split() if p.startswith(("BLAS_INFO=", "LAPACK_INFO=", "USE_MKL=", "USE_MKLDNN="))]
    return " ".join(sorted(set(info))) or "; ".join(keys)[:200]
