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
