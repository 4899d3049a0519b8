import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.test_gpu_train import *
from nerf_tex_amd.train import Trainer
model, spec, wts = make_model((1, 6), dense_media=True)
n, S, P = (int(sys.argv[1]) if len(sys.argv) > 1 else 96), (int(sys.argv[2]) if len(sys.argv) > 2 else 48), 7
ro, rd, t, cone, params, color, alpha = batch(3, n, S, P, "carpet")
okw, loss = make_loss("alpha_smape")
tr = Trainer(model, max_rays=n, n_samples=S, perturb=False)
z = orc.z_values(t, S, np.float32)
val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=11)
torch.cuda.synchronize()
got = tr.gradients()
want_val, wc, wa, wg = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw)
flat = np.concatenate([g.ravel() for g in wg])
for name, sl in layer_slices(spec):
    print(name.ljust(24), '%.3e' % rel_linf(got[sl], flat[sl]), '%.3e' % np.abs(flat[sl]).max())
# float32 autograd for comparison
want32 = tro.step_gradients(wts, spec, ro, rd, z, params, cone, color, alpha, okw, dtype=torch.float32)[3]
f32 = np.concatenate([g.ravel() for g in want32])
for name, sl in layer_slices(spec):
    print('f32 autograd', name.ljust(24), '%.3e' % rel_linf(f32[sl], flat[sl]))
