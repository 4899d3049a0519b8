"""One training step on the GPU (`ntx_trainer_*`, ABI v5) against the reference's step restated with float64 autograd
(oracle/train_oracle.py <- network/train.py:61-67, renderer.py:92-213, loss.py:6-59).  `-m gpu`.

The bar of VERDICT r3 #6: the gradient of every layer within 1e-4 rel-Linf of float64 autograd on three model families, an optimiser
step that is bit-reproducible, Adam against its restatement.  Beyond it: sample counts off every granule of the kernels, the density
regulariser and rays that miss the proxy (what the shipped training configs and their pixel sampler bring), the five shipped training configs'
blocks as written, the reference's batch dict, independence of capacity and history, data parallel steps (two ranks sharing the GPU; the
all-reduce on a one-rank RCCL communicator).  tools/dev/soak_train.py runs random mixtures of all of it."""

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from oracle import train_oracle as tro
from tests.common import make_model

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = np.float32


def dev():
    return torch.device("cuda", 0)


def rel_linf(got, want):
    return float(np.max(np.abs(np.asarray(got, np.float64) - want)) / max(np.max(np.abs(want)), 1e-300))


@pytest.mark.parametrize("ak,bk,M,N,K", [(1, 0, 300, 200, 77), (1, 0, 257, 129, 256), (0, 0, 337, 256, 1000), (1, 0, 128, 128, 32), (0, 0, 72, 3, 5000), (1, 0, 4096, 256, 128),
                                         (1, 0, 512, 256, 340), (0, 0, 340, 256, 2048)])
def test_gemm_kernel_against_float64(ak, bk, M, N, K):
    """The contraction the trainer is made of, in its three operand layouts, at sizes off the 128 x 128 x 16 tiles."""
    import ctypes as C
    from nerf_tex_amd import _lib
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(size=(M, K) if ak else (K, M)).astype(F); B = rng.normal(size=(N, K) if bk else (K, N)).astype(F)
    bias = rng.normal(size=N).astype(F)
    want = (A.astype(np.float64) if ak else A.astype(np.float64).T) @ (B.astype(np.float64).T if bk else B.astype(np.float64)) + bias
    want = np.maximum(want, 0)
    dA, dB, db = (torch.as_tensor(x, device=dev()) for x in (A, B, bias))
    out = torch.full((M, N + 3), 7.0, device=dev())                                # a wider row: ldc > N, the columns beyond stay untouched
    with torch.cuda.device(dev()):
        _lib.check(_lib.lib.ntx_gemm_f32(dA.data_ptr(), A.shape[1], ak, dB.data_ptr(), B.shape[1], bk, out.data_ptr(), N + 3, M, N, K, db.data_ptr(), 1,
                                         torch.cuda.current_stream(dev()).cuda_stream))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert (got[:, N:] == 7).all()
    assert rel_linf(got[:, :N], want) <= 2e-6 * np.sqrt(K)


def batch(seed, n, S, P, fam):
    """A training batch in the shape of config_carpet_train.py: rays from a few cameras through the family's box (all hit), per-image
    parameters, targets."""
    from nerf_tex_amd import synthetic
    rng = np.random.default_rng(seed)
    f = synthetic.FAMILIES[fam]
    ro, rd, t, cone = synthetic.all_hit_rays(n, f["b_0"], f["b_1"], f["cam"])
    params = np.tile(np.asarray([f["params"]], F), (n, 1))
    params[:, :] *= rng.uniform(0.8, 1.2, size=(n, P)).astype(F)
    color = rng.uniform(0, 1, size=(n, 3)).astype(F)
    alpha = (rng.uniform(0, 1, size=n) > 0.3).astype(F) * rng.uniform(0.5, 1, size=n).astype(F)
    return ro, rd, t, cone, params, color, alpha


LOSSES = {"alpha_smape": (dict(kind="alpha", loss_fn="smape", alpha_loss_fn="mse"), dict(loss_fn="network.loss.smape", alpha_loss_fn="network.loss.mse")),
          "alpha_mse_soft": (dict(kind="alpha", loss_fn="mse", gamma=0.5, use_hard_mask=False), dict(loss_fn="network.loss.mse", gamma=0.5, use_hard_mask=False)),
          "nerf_mse": (dict(kind="nerf", loss_fn="mse"), dict(loss_fn="network.loss.mse"))}


def make_loss(name):
    from nerf_tex_amd import loss as L
    okw, pkw = LOSSES[name]
    return okw, (L.AlphaLoss(**pkw) if okw["kind"] == "alpha" else L.NerfLoss(**pkw))


def layer_slices(spec):
    out, p = [], 0
    for name, i, o in orc.layer_table(spec):
        out.append((name + ".kernel", slice(p, p + i * o))); p += i * o
        out.append((name + ".bias", slice(p, p + o))); p += o
    return out


@pytest.mark.parametrize("fam,npar,blur,loss_name,bkgd", [("carpet", (1, 6), None, "alpha_smape", False), ("grass", (1, 4), None, "nerf_mse", True),
                                                         ("grass_filtered", (2, 3), 0, "alpha_mse_soft", False)])
@pytest.mark.parametrize("perturb", [False, True])
def test_gradients_match_float64_autograd(fam, npar, blur, loss_name, bkgd, perturb):
    """dL/dW of all 13 layers (kernels and biases) after one forward + backward of the fused step, per layer within 1e-4 rel-Linf of float64
    autograd through the restated renderer and loss; also the loss value and the predictions."""
    check_gradients(fam, npar, blur, loss_name, bkgd, perturb, 96, 48)


@pytest.mark.parametrize("fam", ["carpet", "fur", "grass", "grass_filtered", "plush"])
def test_shipped_training_configs_run_as_written(fam):
    """Every shipped training config's model / loss / schedule / renderer blocks, verbatim (tests/golden/train_configs.json, made by
    oracle/gen_golden.py from the reference's config modules), through `Trainer.from_config`: the trainer is what the config says -- 256
    samples, perturb, AlphaLoss(smape, mse), Adam from 5e-4 decaying over 5e5 steps, grass_filtered with blur_idx 0 and raw_noise_std 0.1 --
    sized for its batch of 4 x 256 rays, and a step on a slice of such a batch gives the restated step's loss and predictions."""
    import json, os
    from nerf_tex_amd.train import Trainer
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_configs.json")))[fam]
    npar = tuple(cfg["model_config"]["n_parameters"])
    tr, loss = Trainer.from_config(cfg)
    r = cfg["renderer_config"]
    assert tr.max_rays == 1024 and tr.n_samples == 256 and tr.perturb is True and tr.lrate == 5e-4 and tr.lrate_decay == 500
    assert tr.raw_noise_std == r.get("raw_noise_std", 0.0) and tr.blur_idx == r.get("blur_idx") and type(loss).__name__ == "AlphaLoss"
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=npar)
    n, S, P = 48, 256, sum(npar)
    ro, rd, t, cone, params, color, alpha = batch(2, n, S, P, fam if fam != "plush" else "grass")
    wts = orc.split_blob(spec, tr.weights())
    val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=5)
    z = orc.z_values_perturbed(t, S, 5, np.float32)
    noise = tr.raw_noise_std * orc.noise_normals(n, S, 5, dtype=np.float32).astype(np.float64) if tr.raw_noise_std > 0 else None
    want_val, wc, wa, _ = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, dict(kind="alpha", loss_fn="smape", alpha_loss_fn="mse"),
                                              blur_idx=tr.blur_idx, noise=noise)
    assert orc.rel_linf(np.concatenate([cp.cpu().numpy(), ap.cpu().numpy()[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)) <= 1e-4
    assert abs(float(val.item()) - want_val) <= 1e-4 * abs(want_val)
    g = tr.gradients()
    assert np.isfinite(g).all() and np.abs(g).max() > 1e-6
    tr.apply_gradients()
    assert tr.iterations == 1


@pytest.mark.parametrize("bkgd", [False, True])
def test_rays_that_miss_the_proxy_stay_in_the_batch(bkgd):
    """pixel_sampler.Proxy tests the proxy on an 8x coarser grid (pixel_sampler.py:46-62), so a training batch can hold rays with t = inf;
    Renderer.__call__ filters them out, scatters 0 -- or the background -- back and the loss runs over all rays (renderer.py:58-86).  Here they
    stay in the batch as rays whose every alpha is 0: predictions, loss and every layer's gradient equal the restated filter-and-scatter."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((2, 3), dense_media=True)
    n, S = 96, 48
    ro, rd, t, cone, params, color, alpha = batch(6, n, S, 5, "grass_filtered")
    miss = np.zeros(n, bool); miss[[0, 5, 17, 31, 32, 33, 64, 95]] = True
    t = t.copy(); t[miss] = np.inf
    cone = cone.copy(); cone[miss] = np.nan                                       # whatever a ray sampler leaves there
    okw, loss = make_loss("alpha_smape")
    tr = Trainer(model, max_rays=n, n_samples=S, perturb=True, blur_idx=0, raw_noise_std=0.1)
    val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, composite_bkgd=bkgd, bkgd_color=(1., .5, .25), seed=11)
    torch.cuda.synchronize()
    cp, ap = cp.cpu().numpy(), ap.cpu().numpy()
    assert (ap[miss] == 0).all() and (cp[miss] == (np.asarray([1., .5, .25], np.float32) if bkgd else 0)).all()
    M = n * S
    z = orc.z_values_perturbed(np.where(np.isfinite(t), t, 0).astype(np.float32), S, 11, np.float32)
    z[miss] = np.inf
    noise = 0.1 * orc.noise_normals(n, S, 11, dtype=np.float32).astype(np.float64)
    masks = [(tr.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
    sigma_mask = ((tr.activation(10, M).reshape(n, S) + noise.astype(np.float32)) > 0).astype(np.float64)
    want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, np.nan_to_num(cone), color, alpha, okw, masks=masks, sigma_mask=sigma_mask, blur_idx=0,
                                              composite_bkgd=bkgd, bkgd=(1., .5, .25), noise=noise)
    assert abs(float(val.item()) - want_val) <= 1e-5 * abs(want_val) + 1e-7
    assert orc.rel_linf(np.concatenate([cp, ap[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)) <= 1e-4
    got, flat = tr.gradients(), np.concatenate([g.ravel() for g in wg])
    assert np.isfinite(got).all() and np.abs(flat).max() > 1e-6
    worst = {name: rel_linf(got[sl], flat[sl]) for name, sl in layer_slices(spec)}
    assert max(worst.values()) <= 1e-4, {k: v for k, v in worst.items() if v > 1e-5}


def test_train_step_takes_the_reference_batch_dict():
    """`Trainer.train_step(data, ...)` on the dict network/dataset.py hands train.py:61 -- [B, R, ...] tensors, one parameter row per image --
    is `step` on the flattened rays with every ray of image b using parameter row b (renderer.py:54): same weights afterwards, bit for bit."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    B, R, S = 3, 32, 40
    ro, rd, t, cone, params, color, alpha = batch(13, B * R, S, 7, "carpet")
    per_image = params[::R].copy()                                               # [B, P]
    okw, loss = make_loss("alpha_smape")
    data = dict(rays_o=ro.reshape(B, R, 3), rays_d=rd.reshape(B, R, 3), t=t.reshape(B, R, 2), cone_scale=cone.reshape(B, R, 1), parameters=per_image,
                color=color.reshape(B, R, 3), alpha=alpha.reshape(B, R))
    a = Trainer(model, max_rays=B * R, n_samples=S); b = Trainer(model, max_rays=B * R, n_samples=S)
    out = a.train_step(data, loss, composite_bkgd=True, bkgd_color=(.2, .4, .6), seed=9)
    assert out["color_pred"].shape == (B, R, 3) and out["alpha_pred"].shape == (B, R) and out["loss"].shape == (1,)
    val = b.step(ro, rd, t, np.repeat(per_image, R, 0), cone, color, alpha, loss, composite_bkgd=True, bkgd_color=(.2, .4, .6), seed=9)
    assert float(val.item()) == float(out["loss"].item()) and np.array_equal(a.weights(), b.weights()) and a.iterations == 1


def test_sample_noise_matches_the_restated_draws():
    """ntx_sample_noise: raw_noise_std * N(0,1) per (seed, ray, sample) -- Philox4x32-10 counter (sample, ray, 1), Box-Muller as tf.random.normal
    does it -- against the restated generator, identity and strided ray index maps."""
    import ctypes as C
    from nerf_tex_amd import _lib
    n, S, seed = 37, 19, 0x1234567890ab
    for ray_index in (None, (1000, 8, 64)):
        out = torch.empty((n, S), device="cuda")
        opts = _lib.render_opts(raw_noise_std=0.25, ray_index=ray_index)
        _lib.check(_lib.lib.ntx_sample_noise(n, S, seed, C.byref(opts), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        want = 0.25 * orc.noise_normals(n, S, seed, ray_index=ray_index, dtype=np.float64)
        assert np.abs(out.cpu().numpy() - want).max() <= 2e-6 and 0.2 < out.std().item() < 0.3
    assert _lib.lib.ntx_sample_noise(n, S, seed, None, out.data_ptr(), None) == _lib.NTX_E_INVALID       # the flag's opts are not optional


def test_gradients_with_the_density_regulariser():
    """config_grass_filtered_train.py as written: perturb, blur_idx 0 AND raw_noise_std 0.1 (:96-102) -- N(0, 0.1) added to every sample's
    density before its ReLU (renderer.py:190-195), drawn inside the step like the jitter; the oracle gets the restated draws."""
    check_gradients("grass_filtered", (2, 3), 0, "alpha_smape", False, True, 96, 48, raw_noise_std=0.1)


@pytest.mark.parametrize("n,S", [(2, 3), (3, 5), (50, 37), (301, 33), (700, 64)])
def test_gradients_at_ragged_sizes(n, S):
    """The same at sample counts off every granule of the kernels: 6 and 15 samples (less than one block of 32 rows), 1850 samples (the last block of 32 rows ragged, fewer groups of four blocks
    than workgroups, a workgroup's second block of a pair empty), 9933 (ragged, an odd number of groups), 44 800 (more groups than the 256
    persistent workgroups: some walk two pairs of blocks through the layer chain, some one and a half)."""
    check_gradients("carpet", (1, 6), None, "alpha_smape", False, False, n, S, floor_check=100 <= n * S <= 20000)      # (the free-branch comparison too, where a batch is one and two more autograd passes take seconds)


def check_gradients(fam, npar, blur, loss_name, bkgd, perturb, n, S, floor_check=True, raw_noise_std=0.0):
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model(npar, dense_media=True)
    P = sum(npar)
    ro, rd, t, cone, params, color, alpha = batch(3, n, S, P, fam)
    okw, loss = make_loss(loss_name)
    tr = Trainer(model, max_rays=n, n_samples=S, perturb=perturb, blur_idx=blur, raw_noise_std=raw_noise_std)
    # the sample depths the kernel places itself (renderer.py:101-111; with perturb: the product's Philox jitter), restated for the oracle
    z = orc.z_values_perturbed(t, S, 11, np.float32) if perturb else orc.z_values(t, S, np.float32)
    val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, composite_bkgd=bkgd, bkgd_color=(1., .5, .25), seed=11)
    torch.cuda.synchronize()
    got = tr.gradients()
    # float64 autograd, branched like the float32 forward pass was: the signs of the activations the step kept (a pre-activation within
    # rounding of zero falls on either side of its ReLU depending on summation order -- in TensorFlow's float32 as much as here)
    M = n * S
    masks = [(tr.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
    # the density regulariser (renderer.py:190-192): the product's restated draws, keyed like the jitter by (seed, ray, sample)
    noise = raw_noise_std * orc.noise_normals(n, S, 11, dtype=np.float32).astype(np.float64) if raw_noise_std > 0 else None
    sigma_mask = ((tr.activation(10, M).reshape(n, S) + (0 if noise is None else noise.astype(np.float32))) > 0).astype(np.float64)
    kw = dict(blur_idx=blur, composite_bkgd=bkgd, bkgd=(1., .5, .25), noise=noise)
    want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, masks=masks, sigma_mask=sigma_mask, **kw)
    # (a handful of rays with a handful of coarse steps each: 1 - exp(-sigma dist) at dist ~ 0.5 carries a float32 sigma's rounding five times as
    # far, and nothing averages out -- those cases are here for the kernels' granules, at five times the tolerance)
    tiny = 5.0 if n * S < 100 else 1.0
    assert abs(float(val.item()) - want_val) <= (1e-5 if n * S >= 1000 else 1e-4) * tiny * abs(want_val) + 1e-7
    assert orc.rel_linf(np.concatenate([cp.cpu().numpy(), ap.cpu().numpy()[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)) <= 1e-4 * tiny
    flat = np.concatenate([g.ravel() for g in wg])
    assert flat.size == got.size == tr.n_weights
    worst = {name: rel_linf(got[sl], flat[sl]) for name, sl in layer_slices(spec)}
    assert max(worst.values()) <= 1e-4 * tiny, {k: v for k, v in worst.items() if v > 1e-5}
    assert np.abs(flat).max() > 1e-6 or n * S < 100                              # a gradient worth the name (a handful of samples may see none)
    if not floor_check:
        return
    # ... and against float64 autograd left to its own branches: as close as float32 autograd of the same restatement gets (the float32 floor
    # of this comparison, measured beside it), and the two patterns differ in a handful of units
    free = np.concatenate([g.ravel() for g in tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, **kw)[3]])
    f32 = np.concatenate([g.ravel() for g in tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, dtype=torch.float32, **kw)[3]])
    for name, sl in layer_slices(spec):
        floor = rel_linf(f32[sl], free[sl])
        assert rel_linf(got[sl], free[sl]) <= max(1e-4, 4 * floor), (name, rel_linf(got[sl], free[sl]), floor)


def step_errors(model, spec, wts, fam, n, S, loss_name, perturb=False, bkgd=False, noise_std=0.0, blur=None, seed=11, batch_seed=3, miss=None, cap=None, floors=False):
    """One `gradients_step` of a batch described by its knobs against float64 autograd branched like the float32 forward pass: relative
    error of the loss, rel-Linf of the predictions and of every layer's gradient (relative to the layer's largest entry; a layer that
    vanishes beside the others -- 1e-13 against 1e-3 -- to 1e-6 of the batch's largest).  `floors`: the same restatement under float32 torch
    autograd against the float64 one, per layer (what float32 itself can do on this batch).  tools/dev/soak_train.py draws the knobs at random."""
    from nerf_tex_amd.train import Trainer
    ro, rd, t, cone, params, color, alpha = batch(batch_seed, n, S, sum(spec.n_parameters), fam)
    miss = np.zeros(n, bool) if miss is None else np.asarray(miss, bool)
    t = t.copy(); t[miss] = np.inf
    okw, loss = make_loss(loss_name)
    tr = Trainer(model, max_rays=cap or n, n_samples=S, perturb=perturb, blur_idx=blur, raw_noise_std=noise_std)
    val, cp, ap_ = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, composite_bkgd=bkgd, bkgd_color=(1., .5, .25), seed=seed)
    torch.cuda.synchronize()
    M = n * S
    tf = np.where(np.isfinite(t), t, 0).astype(np.float32)
    z = orc.z_values_perturbed(tf, S, seed, np.float32) if perturb else orc.z_values(tf, S, np.float32)
    z = z.copy(); z[miss] = np.inf
    noise = noise_std * orc.noise_normals(n, S, seed, dtype=np.float32).astype(np.float64) if noise_std > 0 else None
    masks = [(tr.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
    sg = tr.activation(10, M).reshape(n, S)
    sigma_mask = ((sg + (0 if noise is None else noise.astype(np.float32))) > 0).astype(np.float64)
    kw = dict(masks=masks, sigma_mask=sigma_mask, blur_idx=blur, composite_bkgd=bkgd, bkgd=(1., .5, .25), noise=noise)
    want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, **kw)
    got, flat = tr.gradients(), np.concatenate([g.ravel() for g in wg])
    gmax = float(np.abs(flat).max())
    lerr = lambda sl: float(np.abs(got[sl] - flat[sl]).max() / max(float(np.abs(flat[sl]).max()), 1e-6 * gmax, 1e-30))
    want_pred = np.concatenate([wc, wa[:, None]], -1)
    out = dict(e_loss=abs(float(val.item()) - want_val) / (abs(want_val) + 1e-7),
               e_pred=orc.rel_linf(np.concatenate([cp.cpu().numpy(), ap_.cpu().numpy()[:, None]], -1), want_pred),
               layers={nm: lerr(sl) for nm, sl in layer_slices(spec)} if gmax > 0 else {"all": float(np.abs(got).max())},
               finite=bool(np.isfinite(got).all()), got=got, want=flat, gmax=gmax, alpha_pred=wa, sigma=sg, z=z)
    out["e_grad"] = max(out["layers"].values())
    # the composite's adjoint on its own: dL/d raw colour, dL/d raw density as the step left them against float64 autograd of the composite and
    # the loss on the step's OWN float32 network outputs (the network's rounding, which exp(-sigma dist) amplifies by sigma dist, stays out)
    hit = ~miss
    raw, dg = tr.activation(11, M).reshape(n, S, 3), tr.activation(30, M).reshape(n, S, 4)
    if hit.any():
        # the loss is a mean over ALL rays of the batch: the hit rays' share of it, scaled back
        sub = lambda x: None if x is None else np.asarray(x)[hit]
        _, _, _, d_rgb, d_sg = tro.composite_gradients(raw[hit], sg[hit], z[hit], rd[hit], color[hit], alpha[hit], okw, composite_bkgd=bkgd, bkgd=(1., .5, .25), noise=sub(noise))
        scale = hit.sum() / n
        # (the ReLU of the density: autograd's own branch on the same float32 value, so the patterns agree)
        out["e_dsigma"] = rel_linf(dg[hit][..., 3], d_sg * scale) if np.abs(d_sg).max() > 0 else float(np.abs(dg[hit][..., 3]).max())
        out["e_drgb"] = rel_linf(dg[hit][..., :3], d_rgb * scale) if np.abs(d_rgb).max() > 0 else float(np.abs(dg[hit][..., :3]).max())
        out["dsigma_max"] = float(np.abs(d_sg).max() * scale)
    if floors:
        _, fc, fa, fg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, dtype=torch.float32, **kw)
        f32 = np.concatenate([g.ravel() for g in fg])
        out["floor_pred"] = orc.rel_linf(np.concatenate([fc, fa[:, None]], -1), want_pred)
        out["floors"] = {nm: rel_linf(f32[sl], flat[sl]) for nm, sl in layer_slices(spec)}
    return out


def test_saturated_rays_of_soak_seed_3_case_297():
    """The batch tools/dev/soak_train.py --seed 3 drew as case 297 (profiles/r05/soak_train_seed3.txt): grass_filtered, 2 rays x 5 samples, both
    saturated (alpha_pred 1.0 and 0.9999975) against alpha targets of 0, alpha_mse_soft over a background, jittered depths.  The whole gradient
    hangs on a transmittance of 2.5e-6.  Round 5's composite adjoint divided a suffix sum by (1 - a_i) + 1e-10 and was 4.6e-3 .. 7.2e-3 per
    layer from float64 here; the division-free reverse scan (csrc/ntx_train.hip composite_loss_kernel) is held to the plain 1e-4."""
    model, spec, wts = make_model((2, 3), dense_media=True)
    e = step_errors(model, spec, wts, "grass_filtered", 2, 5, "alpha_mse_soft", perturb=True, bkgd=True, seed=840038945, batch_seed=996772, cap=45)
    assert e["alpha_pred"].min() > 0.99999, e["alpha_pred"]                      # the case is the saturated one
    assert e["finite"] and e["e_loss"] <= 1e-4 and e["e_pred"] <= 1e-4
    # the adjoint itself, on the step's own sigma and raw colour: float32 rounding of a five-term scan
    assert e["e_dsigma"] <= 2e-6 and e["e_drgb"] <= 2e-6, (e["e_dsigma"], e["e_drgb"])
    # end to end the float32 NETWORK's rounding of sigma (1e-5 relative, the MLP's own tolerance) reaches the transmittance multiplied by
    # sigma * dist = 13 on these rays: 1.25e-4 on every layer alike, whatever the adjoint does.  The gate: twice the plain one, and a quarter
    # of what float32 autograd of the restatement itself makes of this batch (0.7e-3 .. 1.2e-3)
    f = step_errors(model, spec, wts, "grass_filtered", 2, 5, "alpha_mse_soft", perturb=True, bkgd=True, seed=840038945, batch_seed=996772, cap=45, floors=True)
    assert e["e_grad"] <= 2e-4, {k: v for k, v in e["layers"].items() if v > 1e-5}
    assert all(v <= 0.25 * f["floors"][k] for k, v in e["layers"].items() if f["floors"][k] > 4e-4), (e["layers"], f["floors"])


@pytest.mark.parametrize("n,S,loss_name,bkgd,shift", [(2, 5, "alpha_mse_soft", True, 12.0), (7, 6, "alpha_smape", False, 20.0), (64, 9, "nerf_mse", True, 15.0),
                                                      (33, 70, "alpha_mse_soft", True, 10.0), (128, 130, "alpha_smape", False, 10.0)])
def test_saturated_rays_keep_their_gradient(n, S, loss_name, bkgd, shift):
    """A family of saturated batches: the density head's bias raised until most rays end opaque (alpha_pred = 1 to five digits or more) -- by
    one to three samples with sigma * dist of 10 .. 23 where a ray has a handful of coarse steps (a_i = 1 to float32: behind them the ray's
    transmittance is down at the 1e-10 of renderer.py:198), by accumulation where it has 70 or 130 (the scan's 64-lane chunks and their carry) --,
    with and without the background term.  The composite's adjoint on the step's own outputs within float32 rounding of float64 autograd,
    every layer's gradient end to end at 1e-4 (+ the network's own rounding of sigma through the exponential).  (Raised, not scaled: a head scaled x8 has sigma = -1000 .. 140 and the float32
    network's 1e-5 of that is 1e-3 of every alpha -- the batch would test the network's rounding, not the composite.)"""
    model, spec, wts = make_model((1, 6), dense_media=True)
    blob = model.get_blob().copy()
    blob[dict(layer_slices(spec))["alpha.bias"]] += np.float32(shift)
    model.set_blob(blob); wts = orc.split_blob(spec, blob)
    e = step_errors(model, spec, wts, "carpet", n, S, loss_name, perturb=True, bkgd=bkgd, seed=77, batch_seed=n + S)
    assert (e["alpha_pred"] > 0.99999).sum() >= max(1, n // 4), e["alpha_pred"]         # the batch is what its name says
    assert e["finite"] and e["gmax"] > 1e-8
    assert e["e_loss"] <= 1e-5 and e["e_pred"] <= 1e-5
    assert e["e_dsigma"] <= 5e-5 and e["e_drgb"] <= 5e-6, (e["e_dsigma"], e["e_drgb"])
    # end to end: the plain 1e-4, plus what the float32 NETWORK's rounding of sigma (1e-5 relative: its own tolerance, tests/test_gpu_parity.py)
    # becomes in exp(-sigma dist) -- 1e-5 * sigma * dist, on every layer alike and whatever the adjoint does; it shows where a batch's whole
    # gradient hangs on one saturated sample's transmittance (the two-ray case: 1.4e-4 at sigma * dist = 12)
    dist = np.diff(e["z"], axis=-1); dist = np.concatenate([dist, dist[:, -1:]], -1)
    assert e["e_grad"] <= 1e-4 + 1e-5 * float((np.maximum(e["sigma"], 0) * dist).max()), {k: v for k, v in e["layers"].items() if v > 1e-5}


def test_direction_segment_per_ray_and_per_sample_agree(monkeypatch):
    """The forward chain evaluates the colour layer's direction segment once per RAY (dirrow_kernel + the HOIST builds) whenever nothing varies
    along a ray -- no blur_idx on an appearance parameter, a sample count that is a multiple of 32 -- and per sample otherwise
    (renderer.py:152-158).  Both against float64 autograd: the per-ray path (S = 64), the per-sample path forced by the environment on the same
    batch (the same gradients BIT FOR BIT), by a blur_idx on an APPEARANCE parameter, and by S = 60."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 4), dense_media=True)
    e_ray = step_errors(model, spec, wts, "grass", 48, 64, "alpha_smape", perturb=True, seed=5, batch_seed=9)
    monkeypatch.setenv("NERFTEX_TRAIN_NO_DIR_HOIST", "1")
    e_smp = step_errors(model, spec, wts, "grass", 48, 64, "alpha_smape", perturb=True, seed=5, batch_seed=9)
    monkeypatch.delenv("NERFTEX_TRAIN_NO_DIR_HOIST")
    for e in (e_ray, e_smp):
        assert e["finite"] and e["e_loss"] <= 1e-5 and e["e_pred"] <= 1e-4 and e["e_grad"] <= 1e-4, {k: v for k, v in e["layers"].items() if v > 1e-5}
    # the row is bias + sum_k dir_map[k] W[k][f] as an fmaf chain in ascending k, which is what the matrix cores' f32 path computes for the segment's
    # k-steps (an output column depends on its own B column only): not merely close, the same bits -- as the render kernels' hoist
    assert np.array_equal(e_ray["got"], e_smp["got"])
    e_blur = step_errors(model, spec, wts, "grass", 48, 64, "alpha_smape", perturb=True, blur=2, seed=5, batch_seed=9)      # parameter 2 = an appearance parameter of [1, 4]
    e_s60 = step_errors(model, spec, wts, "grass", 48, 60, "alpha_smape", perturb=True, seed=5, batch_seed=9)
    for e in (e_blur, e_s60):
        assert e["finite"] and e["e_loss"] <= 1e-5 and e["e_pred"] <= 1e-4 and e["e_grad"] <= 1e-4, {k: v for k, v in e["layers"].items() if v > 1e-5}


def test_training_step_is_bit_reproducible_and_adam_matches_its_restatement():
    """Two trainers from the same weights take the same step bit for bit (weight gradients are summed over the samples in a fixed order);
    Adam under ExponentialDecay (train.py:49-52) within float32 rounding of the float64 restatement, over three iterations."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    n, S = 128, 64
    ro, rd, t, cone, params, color, alpha = batch(5, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    runs = []
    for _ in range(2):
        tr = Trainer(model, max_rays=n, n_samples=S, lrate=5e-4, lrate_decay=0.002, perturb=True)      # decay_steps = 2: the schedule shows within three steps
        w0 = tr.weights(); hist = []
        for it in range(3):
            tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=it)
            g = tr.gradients(); m_before, v_before = tr.adam_state(); w_before = tr.weights()
            tr.apply_gradients()
            hist.append((g, w_before, m_before, v_before, tr.weights(), *tr.adam_state()))
        assert tr.iterations == 3
        runs.append((w0, hist))
    for (g, wb, mb, vb, wa, ma, va), (g2, wb2, mb2, vb2, wa2, ma2, va2) in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(g, g2) and np.array_equal(wa, wa2) and np.array_equal(ma, ma2) and np.array_equal(va, va2)
    for it, (g, wb, mb, vb, wa, ma, va) in enumerate(runs[0][1]):
        ww, mm, vv = tro.adam_step(wb, g, mb, vb, it, 5e-4, decay_steps=2.0, decay_rate=0.1)
        g64, mb64, vb64 = g.astype(np.float64), mb.astype(np.float64), vb.astype(np.float64)
        assert (np.abs(ma - mm) <= 4e-7 * (np.abs(g64) + np.abs(mb64)) + 1e-30).all()                  # float32 rounding of m + (g - m)(1 - beta_1)
        assert (np.abs(va - vv) <= 4e-7 * (g64 * g64 + vb64) + 1e-38).all()
        step = ww - wb.astype(np.float64)
        assert (np.abs(wa.astype(np.float64) - ww) <= 1.01 * np.spacing(np.abs(wa)) + 1e-6 * np.abs(step)).all()   # the updated weight, to its last place
        assert np.abs(step).max() > 1e-5


def test_gradients_do_not_depend_on_capacity_or_history():
    """A trainer created for more rays than a step brings, and one whose buffers still hold a larger batch's activations, gradients and mask
    bits, take the step of a fresh right-sized one bit for bit (nothing behind the last sample is read)."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    n, S = 75, 41                                                                # 3075 samples: ragged
    ro, rd, t, cone, params, color, alpha = batch(7, n, S, 7, "carpet")
    big = batch(8, 200, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    grads = []
    for cap, history in ((n, False), (200, False), (200, True)):
        tr = Trainer(model, max_rays=cap, n_samples=S, perturb=True)
        if history:
            tr.gradients_step(*big[:3], big[4], big[3], big[5], big[6], loss, seed=1)
        tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=3)
        grads.append(tr.gradients())
    assert np.abs(grads[0]).max() > 1e-6
    assert np.array_equal(grads[0], grads[1]) and np.array_equal(grads[0], grads[2])


def test_gradient_allreduce_on_a_one_rank_communicator():
    """ntx_trainer_allreduce_gradients on real RCCL with one rank (ncclAllReduce in place on the trainer's own buffer, on the step's stream):
    the mean over one rank is the gradient itself, bit for bit, and the step after it is the step without it."""
    from nerf_tex_amd.dist import Comm
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    n, S = 64, 32
    ro, rd, t, cone, params, color, alpha = batch(4, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    comm = Comm(0)
    assert comm.world == 1
    a = Trainer(model, max_rays=n, n_samples=S, perturb=False); b = Trainer(model, max_rays=n, n_samples=S, perturb=False)
    a.gradients_step(ro, rd, t, params, cone, color, alpha, loss); g = a.gradients()
    a.sync_gradients(comm)
    torch.cuda.synchronize()
    assert np.array_equal(a.gradients(), g) and np.abs(g).max() > 1e-6
    a.apply_gradients()
    b.step(ro, rd, t, params, cone, color, alpha, loss, comm=comm)
    assert np.array_equal(a.weights(), b.weights())
    comm.close()


def test_two_ranks_train_data_parallel():
    """Data-parallel training with everything but RCCL itself on a 1-GPU box: two processes (torch.distributed.run, gloo) share GPU 0, each
    takes the gradient of its half of a batch, `Trainer.sync_gradients` averages -- the whole batch's gradient to float32 rounding -- and both
    take the same Adam step bit for bit (tests/_dp_train_worker.py).  With a communicator the same mean is one ncclAllReduce behind
    `ntx_trainer_allreduce_gradients`."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29653",
                          os.path.join(root, "tests", "_dp_train_worker.py")], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "DP_TRAIN_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_a_few_steps_fit_a_target():
    """The loop of train.py:61-67 does what a training loop is for: fitting one batch, the loss falls."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    n, S = 256, 64
    ro, rd, t, cone, params, color, alpha = batch(9, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    tr = Trainer(model, max_rays=n, n_samples=S, lrate=5e-4, lrate_decay=500, perturb=True)
    losses = [float(tr.step(ro, rd, t, params, cone, color, alpha, loss).item()) for _ in range(40)]
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0] and np.mean(losses[-5:]) < np.mean(losses[5:10]) < np.mean(losses[:5]), losses[::5]   # random targets: smape cannot fall far, it falls steadily
    model.set_blob(tr.weights())                                                 # the trained weights render through the inference path


def test_the_configs_batch_size_runs_and_refusals():
    """config_carpet_train.py's step -- 4 images x 256 rays x 256 samples = 262 144 samples -- runs; other architectures are refused."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    n, S = 1024, 256
    ro, rd, t, cone, params, color, alpha = batch(1, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    tr = Trainer(model, max_rays=n, n_samples=S, lrate=5e-4, lrate_decay=500)
    l0 = float(tr.step(ro, rd, t, params, cone, color, alpha, loss).item())
    g = tr.gradients()
    assert np.isfinite(l0) and np.isfinite(g).all() and np.abs(g).max() > 0
    for arch in (dict(depth=6), dict(skips=[2]), dict(color_depth=2), dict(width=128, depth=4)):     # (a narrower 8-layer network does train: below)
        flex, _, _ = make_model((1, 6), arch=arch)
        with pytest.raises(_lib.NtxError) as e:
            Trainer(flex, max_rays=8, n_samples=8)
        assert e.value.code == _lib.NTX_E_UNSUPPORTED


@pytest.mark.parametrize("fam,npar,blur,noise_std,n,S", [("carpet", (1, 6), None, 0.0, 1024, 256), ("carpet", (1, 6), None, 0.0, 1021, 255),
                                                         ("grass_filtered", (2, 3), 0, 0.1, 1024, 256), ("fur", (1, 4), None, 0.0, 1024, 256)])
def test_gradients_at_the_configs_batch(fam, npar, blur, noise_std, n, S):
    """The step the row is about: the shipped training configs' 4 images x 256 rays x 256 samples = 262 144 samples
    (config_carpet_train.py:23, 33, 101; config_grass_filtered_train.py with blur_idx 0 and raw_noise_std 0.1, :96-102; config_fur_train.py) --
    8192 blocks of 32 samples over 1024 persistent waves, every workgroup of the weight gradients with its share of every layer, each
    family on its own build of the forward chain -- and a ragged neighbour.  Every layer's gradient, the loss and the predictions against
    float64 autograd, which takes the batch 16 rays at a time, four chunks at once (the loss is a mean over rays: oracle/train_oracle.py
    step_gradients_chunked; 9 s a case on the GPU box's host, 26 s one chunk at a time), branched by the signs of the activations the step kept."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model(npar, dense_media=True)
    ro, rd, t, cone, params, color, alpha = batch(21, n, S, sum(npar), fam)
    okw, loss = make_loss("alpha_smape")
    tr = Trainer(model, max_rays=n, n_samples=S, perturb=True, blur_idx=blur, raw_noise_std=noise_std)
    val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=4)
    torch.cuda.synchronize()
    got = tr.gradients()
    M = n * S
    masks = [tr.activation(k, M) > 0 for k in list(range(8)) + [8, 9]]                   # bool: 67 MB each
    noise = noise_std * orc.noise_normals(n, S, 4, dtype=np.float32).astype(np.float64) if noise_std > 0 else None
    sigma_mask = (tr.activation(10, M).reshape(n, S) + (0 if noise is None else noise.astype(np.float32))) > 0
    z = orc.z_values_perturbed(t, S, 4, np.float32)
    want_val, wc, wa, wg = tro.step_gradients_chunked(wts, spec, ro, rd, z, params, cone, color, alpha, okw, chunk_rays=16, masks=masks, sigma_mask=sigma_mask,
                                                      noise=noise, blur_idx=blur)
    assert abs(float(val.item()) - want_val) <= 1e-5 * abs(want_val)
    assert orc.rel_linf(np.concatenate([cp.cpu().numpy(), ap.cpu().numpy()[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)) <= 1e-4
    flat = np.concatenate([g.ravel() for g in wg])
    worst = {name: rel_linf(got[sl], flat[sl]) for name, sl in layer_slices(spec)}
    assert max(worst.values()) <= 1e-4, {k: v for k, v in worst.items() if v > 1e-5}
    assert np.abs(flat).max() > 1e-6


@pytest.mark.parametrize("npar,freqs", [((0, 3), None), ((1, 2), (4, 2, 2)), ((3, 0), (6, 4, 3))])
def test_gradients_with_narrow_encodings(npar, freqs):
    """Models whose pos_map / dir_map are narrower than the shipped families' (no geometry parameters: 63 position features; fewer bands:
    30 ... 57) run on the forward chain's longest build with their streams padded by zero rows -- the same gradients."""
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model(npar, dense_media=True, freqs=freqs)
    n, S, P = 80, 40, sum(npar)
    ro, rd, t, cone, params, color, alpha = batch(17, n, S, 7, "carpet")
    params = np.ascontiguousarray(params[:, :P])
    okw, loss = make_loss("alpha_smape")
    tr = Trainer(model, max_rays=n, n_samples=S, perturb=False)
    val, cp, ap = tr.gradients_step(ro, rd, t, params if P else None, cone, color, alpha, loss)
    torch.cuda.synchronize()
    M = n * S
    masks = [(tr.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
    sigma_mask = (tr.activation(10, M).reshape(n, S) > 0).astype(np.float64)
    z = orc.z_values(t, S, np.float32)
    want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, masks=masks, sigma_mask=sigma_mask)
    assert abs(float(val.item()) - want_val) <= 1e-5 * abs(want_val)
    got, flat = tr.gradients(), np.concatenate([g.ravel() for g in wg])
    worst = {name: rel_linf(got[sl], flat[sl]) for name, sl in layer_slices(spec)}
    assert max(worst.values()) <= 1e-4, {k: v for k, v in worst.items() if v > 1e-5}


@pytest.mark.parametrize("width", [128, 64, 30])
def test_a_narrower_network_trains_inside_the_256_wide_one(width):
    """`ParamNerf(width=w)` with w < 256 (model.py:58: the trunk w wide, the last colour layer w // 2): its weights sit in the corners of
    the 256-wide kernels, the rest is zero and stays zero (nerf_tex_amd/train.py `_widened`).  Loss, predictions and every layer's gradient
    against float64 autograd of the NARROW network; after three Adam steps the native trainer's padding is still exactly zero, the narrow
    weights moved, and they render through the inference path (the flex family) to what the trainer predicted."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.renderer import Renderer
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True, arch=dict(width=width))
    n, S = 96, 40
    ro, rd, t, cone, params, color, alpha = batch(23, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    tr = Trainer(model, max_rays=n, n_samples=S, perturb=False)
    assert tr.n_weights == model.n_weight_floats() < tr._n_native and np.array_equal(tr.weights(), np.asarray(model.get_blob(), np.float32).reshape(-1))
    val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss)
    torch.cuda.synchronize()
    M = n * S
    masks = [(tr.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
    assert [m.shape[1] for m in masks] == [width] * 9 + [width // 2]
    sigma_mask = (tr.activation(10, M).reshape(n, S) > 0).astype(np.float64)
    z = orc.z_values(t, S, np.float32)
    want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, masks=masks, sigma_mask=sigma_mask)
    assert abs(float(val.item()) - want_val) <= 1e-5 * abs(want_val)
    assert orc.rel_linf(np.concatenate([cp.cpu().numpy(), ap.cpu().numpy()[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)) <= 1e-4
    got, flat = tr.gradients(), np.concatenate([g.ravel() for g in wg])
    worst = {name: rel_linf(got[sl], flat[sl]) for name, sl in layer_slices(spec)}
    assert max(worst.values()) <= 1e-4 and np.abs(flat).max() > 1e-6, {k: v for k, v in worst.items() if v > 1e-5}
    tr.apply_gradients()
    for _ in range(2):
        pred = tr.step(ro, rd, t, params, cone, color, alpha, loss)
    native = np.empty(tr._n_native, np.float32)
    for what in (_lib.TRAINER_WEIGHTS, _lib.TRAINER_GRADIENTS, _lib.TRAINER_ADAM_M, _lib.TRAINER_ADAM_V):
        _lib.check(_lib.lib.ntx_trainer_get(tr._h, what, native.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float)), native.size))
        outside = np.ones(native.size, bool); outside[tr._pad] = False
        assert not native[outside].any() and native[tr._pad].any()                       # the padding: exactly zero, weights and moments
    assert tr.iterations == 3 and not np.array_equal(tr.weights(), np.asarray(model.get_blob(), np.float32).reshape(-1))
    st = tr.state_dict()                                                              # resuming speaks the narrow network's sizes
    other = Trainer(make_model((1, 6), seed=5, arch=dict(width=width))[0], max_rays=n, n_samples=S, perturb=False)
    other.load_state_dict(st)
    a = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss)
    b = other.gradients_step(ro, rd, t, params, cone, color, alpha, loss)
    assert torch.equal(a[1], b[1]) and np.array_equal(tr.gradients(), other.gradients())
    model.set_weights_from_trainer(tr)                                                # through the host: the flex render family
    d = lambda x: torch.as_tensor(x, device=dev())
    view = dict(rays_o=d(ro)[None], rays_d=d(rd)[None], t=d(t)[None], parameters=d(params[:1]), cone_scale=d(cone).reshape(1, -1, 1))
    params_row = np.repeat(params[:1], n, 0)
    c_train = tr.gradients_step(ro, rd, t, params_row, cone, color, alpha, loss)[1]
    img = Renderer(model=model, n_samples=S, perturb=False)(**view, training=False)
    assert float((img["color_pred"][0] - c_train).abs().max()) <= 1e-4 * max(1.0, float(c_train.abs().max()))


def test_training_resumes_bit_for_bit(tmp_path):
    """train.py:55-60 / logger.py:30-39, 84-86: a run is checkpointed as model + step + optimizer and continued from there.  Six steps in one go
    equal three steps, `save`, a NEW trainer from other weights, `restore`, three more -- weights, both of Adam's moments and the iteration
    count bit for bit (the moments and the count drive the bias correction and the decayed rate); the checkpoint is a TensorBundle with the
    keys tf.train.Checkpoint gives these objects."""
    from nerf_tex_amd import checkpoint
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    other, _, _ = make_model((1, 6), seed=5)
    n, S = 64, 32
    ro, rd, t, cone, params, color, alpha = batch(12, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    kw = dict(max_rays=n, n_samples=S, lrate=5e-4, lrate_decay=0.004, perturb=True, raw_noise_std=0.05)
    a = Trainer(model, **kw)
    for _ in range(6):
        a.step(ro, rd, t, params, cone, color, alpha, loss)
    b = Trainer(model, **kw)
    for _ in range(3):
        b.step(ro, rd, t, params, cone, color, alpha, loss)
    prefix = b.save(str(tmp_path / "checkpoints" / "ckpt-3"), step=3)
    keys = checkpoint.read_bundle_index(prefix + ".index")
    for k in ("model/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE", "model/layer_with_weights-12/bias/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE",
              "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", "optimizer/beta_1/.ATTRIBUTES/VARIABLE_VALUE", "step/.ATTRIBUTES/VARIABLE_VALUE", "_CHECKPOINTABLE_OBJECT_GRAPH"):
        assert k in keys, k
    assert "optimizer/learning_rate/.ATTRIBUTES/VARIABLE_VALUE" not in keys          # a schedule is not a variable
    c = Trainer(other, **kw)
    info = c.restore(str(tmp_path / "checkpoints"))
    assert info["step"] == 3 and info["iterations"] == 3 and c.iterations == 3
    for _ in range(3):
        c.step(ro, rd, t, params, cone, color, alpha, loss)
    assert a.iterations == c.iterations == 6
    assert np.array_equal(a.weights(), c.weights())
    for x, y in zip(a.adam_state(), c.adam_state()):
        assert np.array_equal(x, y)
    d = Trainer(other, **kw)                                                        # and through state_dict, in memory
    d.load_state_dict(b.state_dict())
    for _ in range(3):
        d.step(ro, rd, t, params, cone, color, alpha, loss)
    assert np.array_equal(a.weights(), d.weights())


def test_weights_reach_the_renderer_without_leaving_the_device():
    """`ntx_set_weights_device`: the render context's weight image remade from the trainer's device memory is the image `ntx_set_weights`
    packs on the host from the same weights -- the renders are bit-identical -- for every family the trainer builds; fp16x3 is refused until
    the images are remade from the host."""
    from nerf_tex_amd import _lib
    from nerf_tex_amd.renderer import Renderer
    from nerf_tex_amd.train import Trainer
    for npar, fam in (((1, 6), "carpet"), ((2, 3), "grass_filtered")):
        model, spec, wts = make_model(npar, dense_media=True)
        twin, _, _ = make_model(npar, seed=9)
        n, S, P = 200, 24, sum(npar)
        ro, rd, t, cone, params, color, alpha = batch(2, n, S, P, fam)
        okw, loss = make_loss("alpha_smape")
        tr = Trainer(model, max_rays=n, n_samples=S)
        for _ in range(2):
            tr.step(ro, rd, t, params, cone, color, alpha, loss)
        d = lambda x: torch.as_tensor(x, device=dev())
        view = dict(rays_o=d(ro)[None], rays_d=d(rd)[None], t=d(t)[None], parameters=d(params[:1]), cone_scale=d(cone).reshape(1, -1, 1))
        r = Renderer(model=twin, n_samples=S, perturb=False)
        twin.set_weights_from_trainer(tr)
        on_device = r(**view, training=False)
        if twin.ctx(0):
            with pytest.raises(_lib.NtxError):
                Renderer(model=twin, n_samples=S, perturb=False, precision="fp16x3")(**view, training=False)
        twin.set_blob(tr.weights())
        through_host = r(**view, training=False)
        for k in ("color_pred", "alpha_pred"):
            assert torch.equal(on_device[k], through_host[k]) and on_device[k].abs().max() > 0


def test_train_loop_renders_validation_views_and_checkpoints(tmp_path):
    """network.train.Train (train.py:7-70) over an in-memory dataset: 200 steps on one batch of a synthetic target -- the loss falls --, the
    validation view rendered every i_img steps through the inference path with the weights handed over on the device (bit-identical to the same
    weights through the host), a checkpoint every i_checkpoint steps of which the newest max_to_keep stay, and a second call that finds them
    resumes at the step they hold and trains only what is left."""
    import json, os
    from nerf_tex_amd import dataset as ds
    from nerf_tex_amd.render import render_image
    from nerf_tex_amd.train import Train
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_configs.json")))["carpet"]
    B, R, S = 2, 128, 32
    ro, rd, t, cone, params, color, alpha = batch(31, B * R, S, 7, "carpet")
    data = dict(rays_o=ro.reshape(B, R, 3), rays_d=rd.reshape(B, R, 3), t=t.reshape(B, R, 2), cone_scale=cone.reshape(B, R, 1), parameters=params[::R].copy(),
                color=color.reshape(B, R, 3), alpha=alpha.reshape(B, R))

    class Batches:                                                # the same batch, for ever (train.py:60 takes what it needs)
        composite_bkgd, bkgd_color = False, (1., 1., 1.)
        def __iter__(self):
            while True:
                yield data

    class Views:                                                  # one 24 x 24 validation view
        height, width, composite_bkgd, bkgd_color = 24, 24, False, (1., 1., 1.)
        def __iter__(self):
            from nerf_tex_amd import synthetic
            f = synthetic.FAMILIES["carpet"]
            vo, vd, vt, vc = synthetic.all_hit_rays(24 * 24, f["b_0"], f["b_1"], f["cam"], seed=3)
            d = lambda x: torch.as_tensor(x, device=dev())
            yield dict(rays_o=d(vo)[None], rays_d=d(vd)[None], t=d(vt)[None], cone_scale=d(vc).reshape(1, -1, 1), parameters=d(params[:1]), seed=77)   # (the config renders with perturb: a fixed jitter stream)

    rcfg = dict(cfg["renderer_config"]); rcfg["n_samples"] = S
    common = dict(model_config=cfg["model_config"], loss_config=cfg["loss_config"], lrate=cfg["lrate"], lrate_decay=cfg["lrate_decay"], renderer_config=rcfg)
    out = Train(str(tmp_path), Batches(), Views(), n_iters=200, logger_config=dict(i_print=20, i_img=100, i_checkpoint=50, max_to_keep=2), **common)
    losses = [v for _, v in out["loss"]]
    assert out["step"] == 200 and len(losses) == 10 and np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0]
    assert sorted(out["images"]) == [100, 200] and out["images"][200][0].shape == (24, 24, 4)
    assert not torch.equal(out["images"][100][0], out["images"][200][0])
    from nerf_tex_amd import png                                   # the Logger's files (logger.py:76-78, 139-144): media/validation/<step>/<view>.png
    shot = np.asarray(png.read_png(str(tmp_path / "media" / "validation" / "200" / "0.png")))
    assert shot.shape[:2] == (24, 24) and sorted(os.listdir(tmp_path / "media" / "validation")) == ["100", "200"]
    kept = sorted(f for f in os.listdir(tmp_path / "checkpoints") if f.endswith(".index"))
    assert kept == ["ckpt-150.index", "ckpt-200.index"]
    from nerf_tex_amd import summary                              # the Logger's TensorBoard file (logger.py:41-44, 60-64, 79-81): the loss every 10 steps, the validation images
    ev = summary.read_events(out["events"])
    assert os.path.dirname(out["events"]) == str(tmp_path) and os.path.basename(out["events"]).startswith("events.out.tfevents.") and ev[0]["file_version"] == "brain.Event:2"
    scal = [e for e in ev if e.get("plugin") == "scalars"]
    assert [e["step"] for e in scal] == list(range(10, 201, 10)) and all(e["tag"] == "Loss" for e in scal)
    assert [round(e["value"], 6) for e in scal if e["step"] % 20 == 0] == [round(float(np.float32(v)), 6) for v in losses]
    pics = [e for e in ev if e.get("plugin") == "images"]
    assert [e["step"] for e in pics] == [100, 200] and pics[1]["tag"] == "Validation Rendering" and pics[1]["size"] == (24, 24) and np.array_equal(pics[1]["images"][0], shot)
    assert open(tmp_path / "checkpoints" / "checkpoint").read() == 'model_checkpoint_path: "ckpt-200"\nall_model_checkpoint_paths: "ckpt-150"\nall_model_checkpoint_paths: "ckpt-200"\n'
    tr, model, renderer = out["trainer"], out["trainer"].model, out["renderer"]
    model.set_blob(tr.weights())                                   # the same weights through the host: the same image
    again = render_image(renderer, Views(), next(iter(Views())))[0]
    assert torch.equal(again, out["images"][200][0])
    more = Train(str(tmp_path), Batches(), None, n_iters=230, logger_config=dict(i_print=10, i_img=0, i_checkpoint=15, max_to_keep=2), **common)
    assert more["step"] == 230 and more["trainer"].iterations == 230 and [s for s, _ in more["loss"]] == [210, 220, 230]
    kept = sorted(f for f in os.listdir(tmp_path / "checkpoints") if f.endswith(".index"))      # a resumed run goes on rotating what the first one left
    assert kept == ["ckpt-210.index", "ckpt-225.index"], kept


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("perturb", [True, False])
def test_coarse_and_fine_training(shared, perturb):
    """n_importance > 0 (renderer.py:125-138, loss.py:41-47, model.py:47-56): the coarse pass, the importance sampler on its weights (no
    gradient through it), the fine pass on the merged depths by a second network -- or by the same one, whose gradient is then the sum of
    both passes' --, the loss of both added.  Losses, all four predictions and every layer's gradient of both networks against float64
    autograd of the two passes on the depths the step itself placed; the fine depths are the coarse ones merged with the restated sampler's
    (within its own conditioning, tests/test_gpu_parity.py)."""
    from nerf_tex_amd.train import CoarseFineTrainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    fine, _, wts_f = make_model((1, 6), seed=3, dense_media=True)
    n, S, NI = 72, 24, 16
    ro, rd, t, cone, params, color, alpha = batch(14, n, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    tr = CoarseFineTrainer(model, None if shared else fine, max_rays=n, n_samples=S, n_importance=NI, perturb=perturb)
    kept = {}
    M, MF = n * S, n * (S + NI)
    def on_coarse():
        torch.cuda.synchronize()
        kept["masks"] = [(tr.coarse.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
        kept["sigma"] = (tr.coarse.activation(10, M).reshape(n, S) > 0).astype(np.float64)
        kept["grad"] = tr.coarse.gradients()
    val, cf, af, cc, ac = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=6, on_coarse=on_coarse)
    torch.cuda.synchronize()
    z_c = orc.z_values_perturbed(t, S, 6, np.float32) if perturb else orc.z_values(t, S, np.float32)
    z_f = tr.last_z.cpu().numpy()
    assert z_f.shape == (n, S + NI) and (np.diff(z_f, axis=-1) >= 0).all()
    merged = np.sort(np.concatenate([z_c, z_f], -1), -1)                          # every coarse depth is among the fine ones
    assert all(np.isin(z_c[r], z_f[r]).all() for r in range(n)) and merged.shape[1] == 2 * S + NI
    masks_f = [(tr.fine.activation(k, MF) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
    sigma_f = (tr.fine.activation(10, MF).reshape(n, S + NI) > 0).astype(np.float64)
    want, (wc2, wa2), (wc1, wa1), g_c, g_f = tro.step_gradients_coarse_fine(wts, None if shared else wts_f, spec, ro, rd, z_c, z_f, params, cone, color, alpha, okw,
                                                                              masks_coarse=kept["masks"], sigma_mask_coarse=kept["sigma"], masks_fine=masks_f, sigma_mask_fine=sigma_f)
    assert abs(float(val.item()) - want) <= 1e-5 * abs(want)
    pred = lambda c, a: np.concatenate([c.cpu().numpy(), a.cpu().numpy()[:, None]], -1)
    assert orc.rel_linf(pred(cf, af), np.concatenate([wc2, wa2[:, None]], -1)) <= 1e-4 and orc.rel_linf(pred(cc, ac), np.concatenate([wc1, wa1[:, None]], -1)) <= 1e-4
    flat = lambda g: np.concatenate([x.ravel() for x in g])
    # The merged depths hold pairs a hair apart (a sampled depth beside a coarse one): 1 - exp(-sigma dist) at dist ~ 1e-6 of the ray carries a
    # float32 sigma's rounding a long way, and the density head's gradient is a sum of such terms.  The bar is 1e-4 or four times what float32
    # AUTOGRAD of the same restatement with the same branches is off by (the float32 floor of the comparison, measured beside it).
    _, _, _, f_c, f_f = tro.step_gradients_coarse_fine(wts, None if shared else wts_f, spec, ro, rd, z_c, z_f, params, cone, color, alpha, okw, masks_coarse=kept["masks"],
                                                       sigma_mask_coarse=kept["sigma"], masks_fine=masks_f, sigma_mask_fine=sigma_f, dtype=torch.float32)
    checks = [(tr.fine.gradients(), flat(g_c), flat(f_c))] if shared else [(tr.coarse.gradients(), flat(g_c), flat(f_c)), (tr.fine.gradients(), flat(g_f), flat(f_f))]
    for got, wantg, f32 in checks:
        assert np.abs(wantg).max() > 1e-6
        for name, sl in layer_slices(spec):
            floor = rel_linf(f32[sl], wantg[sl])
            assert rel_linf(got[sl], wantg[sl]) <= max(1e-4, 4 * floor), (name, rel_linf(got[sl], wantg[sl]), floor)
    if shared:
        assert not np.array_equal(kept["grad"], tr.fine.gradients())             # the coarse pass's gradient alone is not the step's
    w0 = [x.weights() for x in tr.trainers]
    tr.apply_gradients()
    assert all(x.iterations == 1 for x in tr.trainers) and all(not np.array_equal(a, x.weights()) for a, x in zip(w0, tr.trainers))


def test_coarse_fine_trainer_from_a_config():
    """`Trainer.from_config` on a reference-format config with n_importance and network.model.CoarseFine: two networks, two trainers."""
    import json, os
    from nerf_tex_amd.train import CoarseFineTrainer, Trainer
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_configs.json")))["carpet"]
    cfg = dict(cfg, model_config={"module": "network.model.CoarseFine", "model_config": dict(cfg["model_config"])},
               renderer_config=dict(cfg["renderer_config"], n_samples=32, n_importance=16))
    tr, loss = Trainer.from_config(cfg, max_rays=64)
    assert isinstance(tr, CoarseFineTrainer) and not tr.shared and tr.n_samples == 32 and tr.n_importance == 16 and tr.model_fine.name == "model_fine"
    B, R = 2, 32
    ro, rd, t, cone, params, color, alpha = batch(3, B * R, 32, 7, "carpet")
    data = dict(rays_o=ro.reshape(B, R, 3), rays_d=rd.reshape(B, R, 3), t=t.reshape(B, R, 2), cone_scale=cone.reshape(B, R, 1), parameters=params[::R].copy(),
                color=color.reshape(B, R, 3), alpha=alpha.reshape(B, R))
    out = tr.train_step(data, loss)
    assert sorted(out) == ["alpha_pred", "alpha_pred_coarse", "color_pred", "color_pred_coarse", "loss"] and np.isfinite(float(out["loss"].item())) and tr.iterations == 1


def test_train_loop_with_a_coarse_and_a_fine_network(tmp_path):
    """`Train` on a configuration with n_importance > 0 and network.model.CoarseFine: both networks step, the validation view goes through
    `Renderer(model, model_fine, n_importance)` with both weight sets handed over on the device, ONE checkpoint holds 'model' and 'model_fine'
    with their optimizer slots (train.py:55), and a run continued from it ends bit for bit where the uninterrupted one does."""
    import json, os
    from nerf_tex_amd.train import Train
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_configs.json")))["carpet"]
    B, R, S, NI = 2, 64, 24, 16
    ro, rd, t, cone, params, color, alpha = batch(41, B * R, S, 7, "carpet")
    data = dict(rays_o=ro.reshape(B, R, 3), rays_d=rd.reshape(B, R, 3), t=t.reshape(B, R, 2), cone_scale=cone.reshape(B, R, 1), parameters=params[::R].copy(),
                color=color.reshape(B, R, 3), alpha=alpha.reshape(B, R))

    class Batches:
        composite_bkgd, bkgd_color = False, (1., 1., 1.)
        def __iter__(self):
            while True:
                yield data

    class Views:
        height, width, composite_bkgd, bkgd_color = 16, 16, False, (1., 1., 1.)
        def __iter__(self):
            from nerf_tex_amd import synthetic
            f = synthetic.FAMILIES["carpet"]
            vo, vd, vt, vc = synthetic.all_hit_rays(256, f["b_0"], f["b_1"], f["cam"], seed=3)
            d = lambda x: torch.as_tensor(x, device=dev())
            yield dict(rays_o=d(vo)[None], rays_d=d(vd)[None], t=d(vt)[None], cone_scale=d(vc).reshape(1, -1, 1), parameters=d(params[:1]), seed=5)

    common = dict(model_config={"module": "network.model.CoarseFine", "model_config": dict(cfg["model_config"])}, loss_config=cfg["loss_config"], lrate=cfg["lrate"],
                  lrate_decay=cfg["lrate_decay"], renderer_config=dict(cfg["renderer_config"], n_samples=S, n_importance=NI))
    np.random.seed(11)
    whole = Train(str(tmp_path / "a"), Batches(), Views(), n_iters=30, logger_config=dict(i_print=10, i_img=30, i_checkpoint=0), **common)
    np.random.seed(11)
    first = Train(str(tmp_path / "b"), Batches(), None, n_iters=20, logger_config=dict(i_print=10, i_img=0, i_checkpoint=20), **common)
    keys = __import__("nerf_tex_amd.checkpoint", fromlist=["x"]).read_bundle_index(str(tmp_path / "b" / "checkpoints" / "ckpt-20") + ".index")
    assert "model/layer_with_weights-3/kernel/.ATTRIBUTES/VARIABLE_VALUE" in keys and "model_fine/layer_with_weights-12/bias/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE" in keys
    np.random.seed(99)                                             # other initial weights: everything comes from the checkpoint
    rest = Train(str(tmp_path / "b"), Batches(), Views(), n_iters=30, logger_config=dict(i_print=10, i_img=30, i_checkpoint=0), **common)
    assert whole["step"] == rest["step"] == 30 and first["step"] == 20
    for x, y in zip(whole["trainer"].trainers, rest["trainer"].trainers):
        assert np.array_equal(x.weights(), y.weights()) and x.iterations == y.iterations == 30
    assert torch.equal(whole["images"][30][0], rest["images"][30][0]) and whole["images"][30][0].shape == (16, 16, 4)
    losses = [v for _, v in whole["loss"]]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_full_batch_properties_without_an_oracle():
    """At the configs' 1024 x 256 batch, properties that need no restatement: the step is bit-reproducible across trainers (262 144 samples
    summed in a fixed order by 256 persistent workgroups); the batch's gradient is the mean of its four images' gradients taken alone (the
    losses are means over rays: loss.py:51-59) to float32 summation order; and the jitter is keyed by the ray's index in the batch, so image
    b alone with the same seed samples other depths than image b inside the batch -- the comparison runs on given depths."""
    from nerf_tex_amd.renderer import Renderer
    from nerf_tex_amd.train import Trainer
    model, spec, wts = make_model((1, 6), dense_media=True)
    B, R, S = 4, 256, 256
    ro, rd, t, cone, params, color, alpha = batch(5, B * R, S, 7, "carpet")
    okw, loss = make_loss("alpha_smape")
    z = Renderer.sample_depths(torch.as_tensor(t, device=dev()), S, perturb=True, seed=3)
    whole = Trainer(model, max_rays=B * R, n_samples=S)
    v0, c0, a0 = whole.gradients_step(ro, rd, t, params, cone, color, alpha, loss, z_vals=z)
    g = whole.gradients()
    again = Trainer(model, max_rays=B * R, n_samples=S)
    v1, c1, a1 = again.gradients_step(ro, rd, t, params, cone, color, alpha, loss, z_vals=z)
    assert np.array_equal(g, again.gradients()) and torch.equal(c0, c1) and torch.equal(a0, a1) and float(v0.item()) == float(v1.item())
    part = Trainer(model, max_rays=R, n_samples=S)
    acc, vals = np.zeros_like(g, dtype=np.float64), []
    for b in range(B):
        sl = slice(b * R, (b + 1) * R)
        v, c, a = part.gradients_step(ro[sl], rd[sl], t[sl], params[sl], cone[sl], color[sl], alpha[sl], loss, z_vals=z[sl])
        assert torch.equal(c, c0[sl]) and torch.equal(a, a0[sl])                  # a ray's prediction does not depend on its batch
        acc += part.gradients(); vals.append(float(v.item()))
    assert abs(np.mean(vals) - float(v0.item())) <= 1e-6 * abs(float(v0.item()))
    worst = max(rel_linf(g[sl_], (acc / B)[sl_]) for _, sl_ in layer_slices(spec))
    assert worst <= 2e-5 and np.abs(g).max() > 1e-6, worst


def test_a_student_network_learns_a_teacher_s_render():
    """The loop end to end on a target that CAN be learnt: the colours and alphas a teacher network renders along 512 rays (the inference
    path, no jitter) are the training targets of a student started from other weights; 1500 steps of AlphaLoss(mse) under Adam with jitter
    bring the loss down by more than an order of magnitude and the student's own render (weights handed over on the device) close to the
    teacher's -- what gradient parity on random targets cannot show: that the steps add up."""
    from nerf_tex_amd.loss import AlphaLoss
    from nerf_tex_amd.renderer import Renderer
    from nerf_tex_amd.train import Trainer
    teacher, _, _ = make_model((1, 6), seed=2, dense_media=True)
    student, _, _ = make_model((1, 6), seed=1)
    n, S = 512, 64
    ro, rd, t, cone, params, _, _ = batch(8, n, S, 7, "carpet")
    d = lambda x: torch.as_tensor(x, device=dev())
    view = dict(rays_o=d(ro)[None], rays_d=d(rd)[None], t=d(t)[None], parameters=d(params[:1]), cone_scale=d(cone).reshape(1, -1, 1))
    params = np.repeat(params[:1], n, 0)
    want = Renderer(model=teacher, n_samples=S, perturb=False)(**view, training=False)
    color, alpha = want["color_pred"][0].contiguous(), want["alpha_pred"][0].contiguous()
    assert float(alpha.std()) > 0.05 and float(color.std()) > 0.03                              # something to learn: alphas 0.2 .. 1 over the rays
    loss = AlphaLoss(loss_fn="network.loss.mse", filter_color_loss=False)
    tr = Trainer(student, max_rays=n, n_samples=S, lrate=5e-4, lrate_decay=0, perturb=True)
    hist = [float(tr.step(ro, rd, t, params, cone, color, alpha, loss).item()) for _ in range(1500)]
    first, last = np.mean(hist[:5]), np.mean(hist[-20:])
    assert np.isfinite(hist).all() and last < first / 10, (first, last, hist[::150])
    student.set_weights_from_trainer(tr)
    got = Renderer(model=student, n_samples=S, perturb=False)(**view, training=False)
    err = float((got["color_pred"] - want["color_pred"]).abs().mean()), float((got["alpha_pred"] - want["alpha_pred"]).abs().mean())
    assert max(err) < 0.05, err
