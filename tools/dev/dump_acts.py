import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.test_gpu_train import *
from nerf_tex_amd.train import Trainer
model, spec, wts = make_model((1, 6), dense_media=True)
n, S, P = 96, 48, 7
ro, rd, t, cone, params, color, alpha = batch(3, n, S, P, "carpet")
okw, loss = make_loss("alpha_smape")
tr = Trainer(model, max_rays=n, n_samples=S, perturb=False)
val, cp, ap = tr.gradients_step(ro, rd, t, params, cone, color, alpha, loss, seed=11)
torch.cuda.synchronize()
d = {f"a{i}": np.asarray(tr.activation(i, n * S)) for i in range(11)}
d["grads"] = np.asarray(tr.gradients()); d["cp"] = cp.cpu().numpy() if hasattr(cp, "cpu") else np.asarray(cp); d["ap"] = ap.cpu().numpy() if hasattr(ap, "cpu") else np.asarray(ap)
np.savez(f"gpurun_out/acts_{sys.argv[1]}.npz", **d)
print(sorted(d))
