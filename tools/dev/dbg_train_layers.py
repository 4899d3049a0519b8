# GPU box: one training step, every kept activation / gradient against numpy on the kernels' own upstream values, then every layer's
# weight gradient against the float64 restatement.  python tools/dev/dbg_train_layers.py [n_rays] [n_samples]
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.test_gpu_train import *
from nerf_tex_amd.train import Trainer
from oracle import torch_cpu
model, spec, wts = make_model((1, 6), dense_media=True)
n, S, P = int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 48, 7
ro, rd, t, cone, params, color, alpha = batch(3, n, S, P, "carpet")
okw, loss = make_loss("alpha_smape")
tr = Trainer(model, max_rays=n, n_samples=S, perturb=False)
val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=11)
torch.cuda.synchronize()
M = n * S
blob = tr.weights(); sl = dict(layer_slices(spec))
K = lambda name, rows: blob[sl[name + ".kernel"]].reshape(rows, -1).astype(np.float64)
B = lambda name: blob[sl[name + ".bias"]].astype(np.float64)
def report(tag, got, want, scale=None):
    err = np.abs(got - want); s = scale or max(np.abs(want).max(), 1e-30)
    bad = np.where(err.max(1) > 1e-4 * s)[0]
    cols = np.where(err[bad[0]] > 1e-4 * s)[0][:16] if bad.size else ''
    print(f"{tag:12s} max err {err.max():.3e} of {s:.3e}; bad rows {bad.size}/{got.shape[0]}: {bad[:8]}; cols of first: {cols}", flush=True)
z = orc.z_values(t, S, np.float32)
pos = (ro[:, None, :] + rd[:, None, :] * z[..., None]).reshape(-1, 3)
dirs = np.repeat(rd / np.linalg.norm(rd, axis=-1, keepdims=True), S, 0)
par = np.repeat(params, S, 0)
ff = torch_cpu.fourier_features; T64 = lambda a: torch.tensor(a, dtype=torch.float64)
pm = torch.cat([ff(T64(pos), spec.pos_freq), ff(T64(par[:, :spec.n_geo]), spec.param_freq)], -1).numpy()
dm = torch.cat([ff(T64(dirs), spec.dir_freq), ff(T64(par[:, spec.n_geo:]), spec.param_freq)], -1).numpy()
Kp, Kd = pm.shape[1], dm.shape[1]
names = [nm for nm, _ in layer_slices(spec)]
print(names[::2])
h = [tr.activation(i, M).astype(np.float64) for i in range(8)]
report("trunk0", h[0], np.maximum(pm @ K("trunk0", Kp) + B("trunk0"), 0))
for i in range(1, 8):
    X = np.concatenate([pm, h[i - 1]], 1) if i == 5 else h[i - 1]
    report(f"trunk{i}", h[i], np.maximum(X @ K(f"trunk{i}", X.shape[1]) + B(f"trunk{i}"), 0))
sig = tr.activation(10, M).astype(np.float64)
report("sigma", sig, h[7] @ K("alpha", 256) + B("alpha"))
feat = h[7] @ K("feature", 256) + B("feature")
c1 = tr.activation(8, M).astype(np.float64)
c1name = [nm for nm in names if nm.startswith("color") and nm.endswith(".kernel")]
print(c1name)
report("c1o", c1, np.maximum(np.concatenate([dm, feat], 1) @ K(c1name[0][:-7], Kd + 256) + B(c1name[0][:-7]), 0))
c2 = tr.activation(9, M).astype(np.float64)
report("c2o", c2, np.maximum(c1 @ K(c1name[1][:-7], 256) + B(c1name[1][:-7]), 0))
# backward
g_c1 = tr.activation(28, M).astype(np.float64)
Wc1 = K(c1name[0][:-7], Kd + 256)
g_f = tr.activation(29, M).astype(np.float64)
report("d feature", g_f, g_c1 @ Wc1[Kd:].T)
dy = {i: tr.activation(20 + i, M).astype(np.float64) for i in range(8)}
for i in range(7, 0, -1):
    Wi = K(f"trunk{i}", 256 + (Kp if i == 5 else 0))[-256:]
    report(f"d h{i - 1}", dy[i - 1], (dy[i] @ Wi.T) * (h[i - 1] > 0))
# weight gradients against the oracle (branched by the kept activations)
masks = [(tr.activation(k, M) > 0).astype(np.float64) for k in list(range(8)) + [8, 9]]
sigma_mask = (tr.activation(10, M).reshape(n, S) > 0).astype(np.float64)
want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, masks=masks, sigma_mask=sigma_mask)
print("loss", float(val.item()), want_val, "pred rel-Linf", orc.rel_linf(np.concatenate([cp.cpu().numpy(), ap.cpu().numpy()[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)))
got = tr.gradients(); flat = np.concatenate([g.ravel() for g in wg])
for name, s_ in layer_slices(spec):
    print(name.ljust(24), '%.3e' % rel_linf(got[s_], flat[s_]), '%.3e' % np.abs(flat[s_]).max(), flush=True)
# d h7 needs the oracle's d feature and d sigma: from the weight gradients above being right it follows; here its own consistency
Wf = K("feature", 256); Wa = K("alpha", 256)[:, 0]
