"""One training step of the reference restated on the CPU: network/train.py:61-67 over network/renderer.py:92-213 and network/loss.py:6-59,
with torch autograd in FLOAT64 standing in for tf.GradientTape, and TF 2.4's Adam + ExponentialDecay (train.py:49-52) written out.

TEST INFRASTRUCTURE ONLY (the rules of nerftex_oracle.py: nothing under nerf_tex_amd/ imports it).  PARITY UNPINNED: TensorFlow cannot
run here and the reference ships no gradients; the forward pass is torch_cpu.py's (checked against nerftex_oracle.py in
tests/test_oracle.py), the gradients are whatever autograd derives from it -- no hand-written adjoint on this side, which is the point:
the kernels' hand-written backward pass is compared with it."""

from __future__ import annotations

import numpy as np
import torch

from . import torch_cpu


def mse(y_true, y_pred):                                                                  # loss.py:51-54
    return torch.mean((y_true - y_pred) ** 2)


def smape(y_true, y_pred, eps=1e-2):                                                      # loss.py:56-59
    return torch.mean(torch.abs(y_true - y_pred) / (y_true + y_pred + eps))


LOSS_FNS = {"mse": mse, "smape": smape}


def nerf_loss(color_true, color_pred, loss_fn="mse"):                                    # loss.py:6-19 (no coarse pass)
    return LOSS_FNS[loss_fn](color_true, color_pred)


def alpha_loss(color_true, alpha_true, color_pred, alpha_pred, loss_fn="mse", alpha_loss_fn=None, gamma=1.0, filter_color_loss=True, use_hard_mask=True):
    """loss.py:21-49 (no coarse pass)."""
    fn, afn = LOSS_FNS[loss_fn], LOSS_FNS[alpha_loss_fn or loss_fn]
    if filter_color_loss:
        mask = (alpha_true[..., None] > 0).to(color_true.dtype) if use_hard_mask else alpha_true[..., None]
        color_true = color_true * mask
        color_pred = color_pred * mask
    return fn(color_true, color_pred) + gamma * afn(alpha_true, alpha_pred)


def model_forward_masked(w, spec, pos, dirs, params, masks):
    """torch_cpu.model_forward with every ReLU replaced by a GIVEN 0/1 pattern (`masks`: trunk 0..depth-1, then the colour layers): the
    network as a float32 forward pass branched it.  A pre-activation within float32 rounding of zero falls on either side of its ReLU
    depending on summation order; autograd through this function follows the pattern it is handed instead of float64's own."""
    ff = torch_cpu.fourier_features
    g, a = spec.n_geo, spec.n_app
    pos_map = ff(pos, spec.pos_freq); dir_map = ff(dirs, spec.dir_freq)
    if g > 0:
        pos_map = torch.cat([pos_map, ff(params[:, :g], spec.param_freq)], -1)
    if a > 0:
        dir_map = torch.cat([dir_map, ff(params[:, g:g + a], spec.param_freq)], -1)
    it = iter(range(0, len(w) - 2, 2)); mk = iter(masks)
    h = pos_map
    for i in range(spec.depth):
        j = next(it)
        h = torch.addmm(w[j + 1], h, w[j]) * next(mk)
        if i in spec.skips:
            h = torch.cat([pos_map, h], -1)
    alpha = torch.addmm(w[-1], h, w[-2])
    j = next(it)
    h = torch.cat([dir_map, torch.addmm(w[j + 1], h, w[j])], -1)
    for _ in range(spec.color_depth):
        j = next(it)
        h = torch.addmm(w[j + 1], h, w[j]) * next(mk)
    j = next(it)
    h = torch.addmm(w[j + 1], h, w[j]) * next(mk)
    j = next(it)
    return torch.addmm(w[j + 1], h, w[j]), alpha


def render(w, spec, rays_o, rays_d, z, parameters, cone_scale, blur_idx=None, map_exr=False, composite_bkgd=False, bkgd=(1., 1., 1.), masks=None,
           sigma_mask=None, noise=None):
    """Renderer.render_rays on given sample depths z [n, S] (renderer.py:114-213; the depths themselves, :101-111, are the caller's:
    with perturb they come from the product's restated generator, nerftex_oracle.sample_depths)."""
    n, S = z.shape
    rays_d_n = rays_d / torch.linalg.norm(rays_d, dim=-1, keepdim=True)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[:, :, None]
    pos = pts.reshape(-1, 3)
    dirs = rays_d_n.repeat_interleave(S, 0)
    params = parameters.repeat_interleave(S, 0)
    if blur_idx is not None:
        scale = (cone_scale.reshape(n, 1, 1) * z[:, :, None]).reshape(-1, 1)
        params = torch.cat([params[:, :blur_idx], params[:, blur_idx, None] * scale, params[:, blur_idx + 1:]], -1)
    if masks is None:
        color, alpha = torch_cpu.model_forward(w, spec, pos, dirs, params)
    else:
        color, alpha = model_forward_masked(w, spec, pos, dirs, params, masks)
    color = color.reshape(n, S, 3); alpha = alpha.reshape(n, S)
    return composite(color, alpha, z, rays_d, map_exr, composite_bkgd, bkgd, sigma_mask, noise)


def composite(color, alpha, z, rays_d, map_exr=False, composite_bkgd=False, bkgd=(1., 1., 1.), sigma_mask=None, noise=None):
    """map_model_output (renderer.py:170-213) on raw network outputs color [n, S, 3], alpha [n, S]."""
    if noise is not None:                                                                   # renderer.py:190-192: [n, S], raw_noise_std * N(0,1)
        alpha = alpha + noise
    dists = z[:, 1:] - z[:, :-1]
    dists = torch.cat([dists, dists[:, -1:]], -1) * torch.linalg.norm(rays_d, dim=-1, keepdim=True)
    rgb = torch.nn.functional.elu(color) + 1 if map_exr else torch.sigmoid(color)
    am = 1. - torch.exp(-(torch.relu(alpha) if sigma_mask is None else alpha * sigma_mask) * dists)
    trans = torch.cumprod(1. - am + 1e-10, -1)
    wts = am * torch.cat([torch.ones_like(trans[:, :1]), trans[:, :-1]], -1)
    c = torch.sum(wts[..., None] * rgb, -2); a = torch.sum(wts, -1)
    if composite_bkgd:
        c = c + (1. - a[..., None]) * torch.as_tensor(bkgd, dtype=c.dtype)
    return c, a


def composite_gradients(raw_rgb, sigma, z, rays_d, color_true, alpha_true, loss, map_exr=False, composite_bkgd=False, bkgd=(1., 1., 1.), noise=None,
                        dtype=torch.float64):
    """The composite and the loss alone under autograd: (loss, color_pred, alpha_pred, dL/d raw_rgb [n, S, 3], dL/d sigma [n, S]) for GIVEN raw
    network outputs -- what a hand-written adjoint of renderer.py:170-213 + loss.py is compared with, apart from the network's own rounding."""
    t_ = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)
    rgb = torch.tensor(np.asarray(raw_rgb), dtype=dtype, requires_grad=True); sg = torch.tensor(np.asarray(sigma), dtype=dtype, requires_grad=True)
    c, a = composite(rgb, sg, t_(z), t_(rays_d), map_exr, composite_bkgd, bkgd, None, t_(noise))
    kw = {k: v for k, v in loss.items() if k != "kind"}
    val = nerf_loss(t_(color_true), c, **kw) if loss["kind"] == "nerf" else alpha_loss(t_(color_true), t_(alpha_true), c, a, **kw)
    val.backward()
    return float(val.detach()), c.detach().numpy(), a.detach().numpy(), rgb.grad.numpy(), sg.grad.numpy()


def step_gradients(w_np, spec, rays_o, rays_d, z, parameters, cone_scale, color_true, alpha_true, loss, blur_idx=None, map_exr=False,
                   composite_bkgd=False, bkgd=(1., 1., 1.), dtype=torch.float64, masks=None, sigma_mask=None, noise=None):
    """(loss, color_pred, alpha_pred, gradients in get_weights() order as a list of arrays) of one step; `loss` = dict(kind='nerf'|'alpha', **kwargs).
    `masks` / `sigma_mask`: the ReLU patterns of a float32 forward pass (model_forward_masked), as 0/1 arrays; `noise` [n, S]: the density
    regulariser's draws (raw_noise_std * N(0,1), renderer.py:190-192)."""
    w = [torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True) for a in w_np]
    t_ = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)
    mk = None if masks is None else [t_(m) for m in masks]
    hit = np.isfinite(np.asarray(z)[:, 0])
    if hit.all():
        c, a = render(w, spec, t_(rays_o), t_(rays_d), t_(z), t_(parameters), t_(cone_scale), blur_idx, map_exr, composite_bkgd, bkgd, mk,
                      None if sigma_mask is None else t_(sigma_mask), t_(noise))
    else:
        # Renderer.__call__ (renderer.py:58-86): rays whose t is inf are filtered out, the rest rendered, the results scattered back into zeros --
        # plus the background colour for the filtered ones when compositing -- and the loss runs over ALL rays
        S = np.asarray(z).shape[1]
        rows = np.repeat(hit, S)
        sub = lambda x: None if x is None else t_(np.asarray(x)[hit])
        ch, ah = render(w, spec, sub(rays_o), sub(rays_d), sub(z), sub(parameters), sub(cone_scale), blur_idx, map_exr, composite_bkgd, bkgd,
                        None if mk is None else [m[torch.as_tensor(rows)] for m in mk], sub(sigma_mask), sub(noise))
        idx = torch.as_tensor(np.nonzero(hit)[0])
        c = torch.zeros((hit.size, 3), dtype=dtype).index_put((idx,), ch)
        a = torch.zeros((hit.size,), dtype=dtype).index_put((idx,), ah)
        if composite_bkgd:
            c = c + torch.as_tensor((~hit)[:, None] * np.asarray(bkgd, np.float64)[None, :], dtype=dtype)
    kw = {k: v for k, v in loss.items() if k != "kind"}
    val = nerf_loss(t_(color_true), c, **kw) if loss["kind"] == "nerf" else alpha_loss(t_(color_true), t_(alpha_true), c, a, **kw)
    val.backward()
    return float(val.detach()), c.detach().numpy(), a.detach().numpy(), [x.grad.numpy() for x in w]


def adam_step(w, g, m, v, iterations, lrate, decay_steps=0.0, decay_rate=0.1, beta_1=0.9, beta_2=0.999, epsilon=1e-7, dtype=np.float64):
    """tf.keras.optimizers.Adam (TF 2.4 ApplyAdam, non-amsgrad) under ExponentialDecay (train.py:49-52): flat arrays in, (w, m, v) out."""
    w, g, m, v = (np.asarray(x, dtype) for x in (w, g, m, v))
    lr = lrate * decay_rate ** (iterations / decay_steps) if decay_steps > 0 else lrate
    t = iterations + 1.0
    # Keras holds beta_1 / beta_2 as float32 hyper-parameters and forms 1 - beta in float32 (optimizer_v2/adam.py: _prepare_local):
    # 1 - 0.999f = 9.99987e-4, not 1e-3
    b1, b2 = np.float32(beta_1), np.float32(beta_2)
    omb1, omb2 = float(np.float32(1) - b1), float(np.float32(1) - b2)
    lr_t = lr * np.sqrt(1.0 - float(b2) ** t) / (1.0 - float(b1) ** t)
    m = m + (g - m) * omb1
    v = v + (g * g - v) * omb2
    return w - lr_t * m / (np.sqrt(v) + epsilon), m, v


def step_gradients_chunked(w_np, spec, rays_o, rays_d, z, parameters, cone_scale, color_true, alpha_true, loss, chunk_rays=16, masks=None, sigma_mask=None,
                           noise=None, workers=None, **kw):
    """`step_gradients` on a batch too large for one autograd pass in float64 (the configs' 1024 rays x 256 samples): the losses of loss.py
    are MEANS over the rays, so the batch's loss and gradient are the ray-count-weighted sums of its chunks' -- evaluated `chunk_rays` rays at
    a time (bounded memory, minutes of CPU; `workers` chunks at once on Python threads: the matrices of a chunk are too small to keep every
    BLAS thread busy), added up in chunk order.  Every ray must hit (the filter-and-scatter branch of step_gradients is per call).
    `masks`: [M, width] arrays (bool is fine), `sigma_mask` / `noise`: [n, S]."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    if workers is None:
        workers = int(os.environ.get("NTX_ORACLE_WORKERS", "4"))
    z = np.asarray(z)
    n, S = z.shape
    assert np.isfinite(z).all(), "chunked evaluation is for all-hit batches"

    def one(r0):
        r1 = min(n, r0 + chunk_rays)
        sl, rows = slice(r0, r1), slice(r0 * S, r1 * S)
        val, c, a, g = step_gradients(w_np, spec, np.asarray(rays_o)[sl], np.asarray(rays_d)[sl], z[sl], np.asarray(parameters)[sl],
                                      None if cone_scale is None else np.asarray(cone_scale)[sl], np.asarray(color_true)[sl],
                                      None if alpha_true is None else np.asarray(alpha_true)[sl], loss,
                                      masks=None if masks is None else [np.asarray(m[rows], np.float64) for m in masks],
                                      sigma_mask=None if sigma_mask is None else np.asarray(sigma_mask[sl], np.float64),
                                      noise=None if noise is None else np.asarray(noise)[sl], **kw)
        return (r1 - r0) / n, val, c, a, g

    starts = list(range(0, n, chunk_rays))
    if workers > 1 and len(starts) > 1:
        with ThreadPoolExecutor(workers) as ex:
            parts = list(ex.map(one, starts))
    else:
        parts = [one(r0) for r0 in starts]
    total, grads, cs, al = 0.0, None, [], []
    for wgt, val, c, a, g in parts:                                                      # in chunk order, whatever order they finished in
        total += val * wgt
        grads = [x * wgt for x in g] if grads is None else [acc + x * wgt for acc, x in zip(grads, g)]
        cs.append(c); al.append(a)
    return total, np.concatenate(cs), np.concatenate(al), grads


def step_gradients_coarse_fine(w_coarse_np, w_fine_np, spec, rays_o, rays_d, z_coarse, z_fine, parameters, cone_scale, color_true, alpha_true, loss,
                               masks_coarse=None, sigma_mask_coarse=None, masks_fine=None, sigma_mask_fine=None, noise_coarse=None, noise_fine=None, dtype=torch.float64, **kw):
    """A step with n_importance > 0 (renderer.py:125-138, loss.py:15-16, 41-47): the coarse pass on z_coarse, the fine pass on z_fine -- GIVEN:
    the sampler carries no gradient (`tf.stop_gradient`, :129) and is restated and tested on its own (nerftex_oracle.sample_pdf) --, the
    loss of both added.  w_fine_np None: one network runs both passes (model_fine is None, :132) and its gradient is the sum.
    Returns (loss, (color, alpha) fine, (color, alpha) coarse, gradients of the coarse network, of the fine one (None when shared))."""
    t_ = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dtype)
    wc = [torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True) for a in w_coarse_np]
    wf = wc if w_fine_np is None else [torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True) for a in w_fine_np]
    mk = lambda m: None if m is None else [t_(x) for x in m]
    args = (t_(rays_o), t_(rays_d))
    c1, a1 = render(wc, spec, *args, t_(z_coarse), t_(parameters), t_(cone_scale), masks=mk(masks_coarse), sigma_mask=t_(sigma_mask_coarse), noise=t_(noise_coarse), **kw)
    c2, a2 = render(wf, spec, *args, t_(z_fine), t_(parameters), t_(cone_scale), masks=mk(masks_fine), sigma_mask=t_(sigma_mask_fine), noise=t_(noise_fine), **kw)
    lk = {k: v for k, v in loss.items() if k != "kind"}
    one = (lambda c, a: nerf_loss(t_(color_true), c, **lk)) if loss["kind"] == "nerf" else (lambda c, a: alpha_loss(t_(color_true), t_(alpha_true), c, a, **lk))
    val = one(c2, a2) + one(c1, a1)
    val.backward()
    g = lambda w: [x.grad.numpy() for x in w]
    return float(val.detach()), (c2.detach().numpy(), a2.detach().numpy()), (c1.detach().numpy(), a1.detach().numpy()), g(wc), None if w_fine_np is None else g(wf)
