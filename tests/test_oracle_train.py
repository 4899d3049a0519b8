"""The training-step restatement (oracle/train_oracle.py <- network/train.py:49-70, network/loss.py) against hand-computed cases and against
finite differences: it is what the GPU trainer's gradients are compared with (tests/test_gpu_train.py), so it gets its own checks.  CPU only."""

import numpy as np
import pytest

from oracle import nerftex_oracle as orc
from oracle import train_oracle as tro

torch = pytest.importorskip("torch")
F = np.float32


def test_losses_known_answers():
    t = torch.tensor([[0.5, 0.0, 1.0]], dtype=torch.float64); p = torch.tensor([[0.25, 0.5, 1.0]], dtype=torch.float64)
    assert float(tro.mse(t, p)) == pytest.approx((0.0625 + 0.25 + 0) / 3)                                   # loss.py:51-54
    assert float(tro.smape(t, p)) == pytest.approx((0.25 / 0.76 + 0.5 / 0.51 + 0) / 3)                       # loss.py:56-59, eps 1e-2
    at = torch.tensor([0.0], dtype=torch.float64); ap = torch.tensor([0.3], dtype=torch.float64)
    # AlphaLoss (loss.py:21-49): a pixel with alpha_true = 0 is masked out of the colour term (hard mask), the alpha term stays
    assert float(tro.alpha_loss(t, at, p, ap)) == pytest.approx(0.0 + 0.09)
    assert float(tro.alpha_loss(t, at + 0.5, p, ap, gamma=2.0, use_hard_mask=False)) == pytest.approx(float(tro.mse(0.5 * t, 0.5 * p)) + 2.0 * 0.04)
    assert float(tro.nerf_loss(t, p, "smape")) == pytest.approx(float(tro.smape(t, p)))


def test_adam_first_step_known_answer():
    # t = 1: m = (1 - b1) g, v = (1 - b2) g^2, lr_t = lr sqrt(1 - b2) / (1 - b1)  ->  w -= lr g / (|g| + eps / sqrt(1 - b2)): a step of lr against the sign
    w = np.asarray([1.0, -2.0, 0.5]); g = np.asarray([0.3, -4.0, 1e-3])
    w1, m1, v1 = tro.adam_step(w, g, np.zeros(3), np.zeros(3), 0, 5e-4)
    omb1, omb2 = float(F(1) - F(0.9)), float(F(1) - F(0.999))                  # Keras forms 1 - beta in float32
    assert np.allclose(m1, omb1 * g) and np.allclose(v1, omb2 * g * g)
    assert np.allclose(w1, w - 5e-4 * np.sign(g), rtol=0, atol=5e-4 * 5e-3)      # (eps / sqrt(1 - b2) = 3.2e-6 against |g| >= 1e-3)
    # ExponentialDecay(lrate, decay_steps, 0.1) without staircase: the rate at iteration 5e5 of config_carpet_train.py (lrate_decay 500) is a tenth
    wa, _, _ = tro.adam_step(w, g, m1, v1, 500000, 5e-4, decay_steps=5e5)
    wb, _, _ = tro.adam_step(w, g, m1, v1, 500000, 5e-5)
    assert np.allclose(wa, wb, rtol=0, atol=1e-12)


def tiny_batch(seed=0, n=6, S=5):
    rng = np.random.default_rng(seed)
    ro = rng.normal(size=(n, 3)); rd = rng.normal(size=(n, 3)); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    t = np.stack([np.full(n, 0.5), np.full(n, 1.5)], -1)
    z = orc.z_values(t.astype(F), S, F).astype(np.float64)
    return ro, rd, z, rng.uniform(0.2, 1, size=(n, 3)), rng.uniform(1e-3, 5e-3, size=(n, 1)), rng.uniform(0, 1, size=(n, 3)), (rng.uniform(0, 1, size=n) > 0.3) * 0.8


@pytest.mark.parametrize("loss", [dict(kind="alpha", loss_fn="smape", alpha_loss_fn="mse"), dict(kind="nerf", loss_fn="mse")])
def test_autograd_matches_finite_differences(loss):
    """The float64 gradients the GPU trainer is held to, against central differences of the same restated step on a small network."""
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=(1, 2), depth=3, width=8, skips=(1,), pos_freq=2, dir_freq=1, param_freq=1)
    rng = np.random.default_rng(1)
    w = [rng.normal(size=(i, o)) * 0.5 if b == 0 else rng.normal(size=o) * 0.1 for _, i, o in orc.layer_table(spec) for b in (0, 1)]
    ro, rd, z, par, cone, ct, at = tiny_batch()
    noise = 0.1 * rng.normal(size=z.shape)                                       # the density regulariser (renderer.py:190-192) rides along
    val, c, a, g = tro.step_gradients(w, spec, ro, rd, z, par, cone, ct, at, loss, blur_idx=0, composite_bkgd=True, bkgd=(1., .5, .2), noise=noise)
    assert abs(val - tro.step_gradients(w, spec, ro, rd, z, par, cone, ct, at, loss, blur_idx=0, composite_bkgd=True, bkgd=(1., .5, .2))[0]) > 1e-6
    f = lambda ws: tro.step_gradients(ws, spec, ro, rd, z, par, cone, ct, at, loss, blur_idx=0, composite_bkgd=True, bkgd=(1., .5, .2), noise=noise)[0]
    for k in range(len(w)):
        idx = tuple(rng.integers(0, s_) for s_ in w[k].shape)
        h = 1e-6
        wp = [x.copy() for x in w]; wp[k][idx] += h
        wm = [x.copy() for x in w]; wm[k][idx] -= h
        fd = (f(wp) - f(wm)) / (2 * h)
        assert fd == pytest.approx(g[k][idx], rel=2e-5, abs=1e-9), (k, idx)
    # handing the restatement its own ReLU pattern changes nothing
    n, S = z.shape
    wt = [torch.tensor(x) for x in w]
    pos = torch.tensor(ro)[:, None, :] + torch.tensor(rd)[:, None, :] * torch.tensor(z)[:, :, None]
    assert val == pytest.approx(float(tro.step_gradients(w, spec, ro, rd, z, par, cone, ct, at, loss, blur_idx=0, composite_bkgd=True, bkgd=(1., .5, .2), noise=noise)[0]))


def _dp_worker(rank, world, port, q):
    """Rank `rank` of a data-parallel step on CPU: the restated step on ITS half of the rays, the product's host-side mean over ranks."""
    import os
    import torch.distributed as dist
    from nerf_tex_amd.train import allreduce_mean_host
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=(1, 2), depth=3, width=8, skips=(1,), pos_freq=2, dir_freq=1, param_freq=1)
    rng = np.random.default_rng(1)
    w = [rng.normal(size=(i, o)) * 0.5 if b == 0 else rng.normal(size=o) * 0.1 for _, i, o in orc.layer_table(spec) for b in (0, 1)]
    ro, rd, z, par, cone, ct, at = tiny_batch(seed=3, n=8, S=5)
    ok = True
    for loss in (dict(kind="alpha", loss_fn="smape", alpha_loss_fn="mse"), dict(kind="nerf", loss_fn="mse")):
        whole = np.concatenate([g.ravel() for g in tro.step_gradients(w, spec, ro, rd, z, par, cone, ct, at, loss, blur_idx=0)[3]])
        lo, hi = rank * 8 // world, (rank + 1) * 8 // world
        mine = np.concatenate([g.ravel() for g in tro.step_gradients(w, spec, ro[lo:hi], rd[lo:hi], z[lo:hi], par[lo:hi], cone[lo:hi], ct[lo:hi], at[lo:hi], loss, blur_idx=0)[3]])
        mean = allreduce_mean_host(mine)
        ok = ok and mean.dtype == np.float32 and bool(np.allclose(mean, whole, rtol=2e-6, atol=1e-7 * np.abs(whole).max())) and not np.allclose(mine, whole, rtol=1e-3)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_step_world2_gloo():
    """N > 1 training on CPU: the losses of loss.py are means over rays, so the mean over ranks of the gradients of equal shards is the whole
    batch's gradient -- the one collective of data-parallel training (`Trainer.sync_gradients`: ncclAllReduce behind the C ABI on GPUs, this
    host-side mean otherwise), executed by two gloo ranks on the restated step."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(60) for p in procs]
    assert res == [(0, True), (1, True)]


def test_train_config_fixture_is_what_the_trainer_is_built_from():
    """tests/golden/train_configs.json (oracle/gen_golden.py train_configs, from the reference's config modules): the five shipped training
    configs' blocks -- what `Trainer.from_config` reads on the GPU box (tests/test_gpu_train.py) -- and the reference paths they name remap
    to this package's classes."""
    import json, os
    from nerf_tex_amd import util
    cfgs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_configs.json")))
    assert sorted(cfgs) == ["carpet", "fur", "grass", "grass_filtered", "plush"]
    for fam, c in cfgs.items():
        assert c["source"] == f"configs/config_{fam}_train.py" and c["batchsize"] * c["rays_per_image"] == 1024 and c["renderer_config"]["n_samples"] == 256
        r = util.remap_reference_config(c)
        assert r["model_config"]["module"] == "nerf_tex_amd.model.ParamNerf" and r["loss_config"]["module"] == "nerf_tex_amd.loss.AlphaLoss"
        loss = util.instantiate(dict(r["loss_config"]))
        d = loss.desc()
        assert d.size > 0                                                        # (the descriptor is built without a device)
    assert cfgs["grass_filtered"]["renderer_config"]["raw_noise_std"] == 0.1 and cfgs["grass_filtered"]["renderer_config"]["blur_idx"] == 0
    assert all("raw_noise_std" not in cfgs[f]["renderer_config"] for f in ("carpet", "fur", "grass", "plush"))


def test_from_config_refuses_what_a_step_does_not_do():
    """`Trainer.from_config` on a config's blocks: a renderer key without a meaning in a training step is an error, and so is the `Train` loop
    on a coarse + fine configuration (n_importance > 0 is a `CoarseFineTrainer`'s: tests/test_gpu_train.py) -- before any device is touched."""
    import copy, json, os
    from nerf_tex_amd.train import Trainer
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_configs.json")))["carpet"]
    c = copy.deepcopy(cfg); c["renderer_config"]["no_such_key"] = 1
    with pytest.raises(TypeError):
        Trainer.from_config(c)
