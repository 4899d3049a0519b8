# GPU box: every 256 -> 256 trunk layer of one forward pass against numpy on the kernel's own input activation
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.test_gpu_train import *
from nerf_tex_amd.train import Trainer
model, spec, wts = make_model((1, 6), dense_media=True)
n, S, P = int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 48, 7
ro, rd, t, cone, params, color, alpha = batch(3, n, S, P, "carpet")
okw, loss = make_loss("alpha_smape")
tr = Trainer(model, max_rays=n, n_samples=S, perturb=False)
val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=11)
torch.cuda.synchronize()
M = n * S
names = [nm for nm, _ in layer_slices(spec)]
blob = tr.weights()
sl = dict(layer_slices(spec))
for i in (1, 2, 3, 6, 7):
    W = blob[sl[f"trunk{i}.kernel"]].reshape(256, 256).astype(np.float64); b = blob[sl[f"trunk{i}.bias"]].astype(np.float64)
    X = tr.activation(i - 1, M).astype(np.float64); Y = tr.activation(i, M)
    want = np.maximum(X @ W + b, 0)
    err = np.abs(Y - want)
    bad_rows = np.where(err.max(1) > 1e-3)[0]
    print(f"trunk{i}: max err {err.max():.3e}; bad rows {bad_rows.size} of {M}; first bad {bad_rows[:8]}; bad cols of first bad row {np.where(err[bad_rows[0]] > 1e-3)[0][:12] if bad_rows.size else ''}")
W = blob[sl["color_half.kernel"]].reshape(256, 128).astype(np.float64); b = blob[sl["color_half.bias"]].astype(np.float64)
X = tr.activation(8, M).astype(np.float64); Y = tr.activation(9, M)
want = np.maximum(X @ W + b, 0); err = np.abs(Y - want)
bad_rows = np.where(err.max(1) > 1e-3)[0]
print(f"c2: max err {err.max():.3e}; bad rows {bad_rows.size} of {M}; first bad {bad_rows[:8]}; bad cols of first bad row {np.where(err[bad_rows[0]] > 1e-3)[0][:40] if bad_rows.size else ''}")
# trunk 0 from the encodings
from oracle import torch_cpu
z = orc.z_values(t, S, np.float32)
pos = (ro[:, None, :] + rd[:, None, :] * z[..., None]).reshape(-1, 3)
par = np.repeat(params, S, 0) if params.shape[0] == n else np.repeat(params, M // params.shape[0], 0)
ff = torch_cpu.fourier_features
pm = torch.cat([ff(torch.tensor(pos, dtype=torch.float64), spec.pos_freq), ff(torch.tensor(par[:, :spec.n_geo], dtype=torch.float64), spec.param_freq)], -1).numpy()
W = blob[sl["trunk0.kernel"]].reshape(-1, 256).astype(np.float64); b = blob[sl["trunk0.bias"]].astype(np.float64)
Y = tr.activation(0, M); want = np.maximum(pm @ W + b, 0); err = np.abs(Y - want)
bad_rows = np.where(err.max(1) > 1e-3)[0]
print(f"trunk0: K = {W.shape[0]}; max err {err.max():.3e}; bad rows {bad_rows.size} of {M}; first bad {bad_rows[:12]}")
# the kept gradients, layer by layer down the chain, against numpy on the kernel's own upstream gradient
g_c1 = tr.activation(28, M).astype(np.float64)          # d c1o (masked)
W = blob[sl["color_hidden0.kernel"]].reshape(-1, 256).astype(np.float64)
Kd = W.shape[0] - 256
want = g_c1 @ W[Kd:].T
got = tr.activation(29, M); err = np.abs(got - want); bad = np.where(err.max(1) > 1e-5 * np.abs(want).max())[0]
print(f"d feature: max err {err.max():.3e} of {np.abs(want).max():.3e}; bad rows {bad.size}: {bad[:10]} .. {bad[-5:] if bad.size else ''}")
Wf = blob[sl["feature.kernel"]].reshape(256, 256).astype(np.float64); Wa = blob[sl["alpha.kernel"]].reshape(256).astype(np.float64)
prev = tr.activation(27, M).astype(np.float64)
for i in range(7, 0, -1):
    Wi = blob[sl[f"trunk{i}.kernel"]].reshape(-1, 256).astype(np.float64)[-256:]
    want = (prev @ Wi.T) * (tr.activation(i - 1, M) > 0)
    got = tr.activation(20 + i - 1, M); err = np.abs(got - want); bad = np.where(err.max(1) > 1e-5 * np.abs(want).max())[0]
    print(f"d h{i - 1}: max err {err.max():.3e} of {np.abs(want).max():.3e}; bad rows {bad.size}: {bad[:10]} .. {bad[-5:] if bad.size else ''}")
    prev = got.astype(np.float64)
i = 7
Wi = blob[sl[f"trunk{i}.kernel"]].reshape(-1, 256).astype(np.float64)[-256:]
um = tr.activation(27, M).astype(np.float64) @ Wi.T
mask = tr.activation(6, M) > 0
got = tr.activation(26, M)
for r in (1840, 1841, 1842, 1843):
    g = got[r]; w = um[r] * mask[r]
    print(r, "valid mask frac %.2f" % mask[r].mean(), "got nonzero frac %.2f" % (g != 0).mean(), "got==unmasked where mask: %.3e" % np.abs(g - w).max(),
          "cols wrong:", np.where(np.abs(g - w) > 1e-7)[0][:16], "kept though masked:", int(((g != 0) & ~mask[r]).sum()), "dropped though kept:", int(((g == 0) & mask[r] & (um[r] != 0)).sum()))
eff = (got[1841] != 0)
sel = um[1841] != 0
best = []
for l in list(range(8)) + [8]:
    A = tr.activation(l, M) > 0
    mism = (A[:, sel] != eff[sel]).sum(1)
    j = int(mism.argmin()); best.append((int(mism[j]), l, j))
print("effective mask of d h6 row 1841 matches (mismatches, layer, row):", sorted(best)[:4])
