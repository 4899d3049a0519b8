"""dev helper (GPU box): soak -- the same launch over and over, every result compared bit for bit with the first one
(render kernel at both precisions with and without the in-kernel jitter, instanced kernel with its dynamic ray hand-out).
usage: python tools/dev/soak.py [seconds per case]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_tex_amd import synthetic
from nerf_tex_amd.model import ParamNerf
from nerf_tex_amd.renderer import Renderer

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
fam = synthetic.FAMILIES["carpet"]
emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
dev = torch.device("cuda", 0)
ro, rd, t, cone = synthetic.all_hit_rays(200000, fam["b_0"], fam["b_1"], fam["cam"], seed=1)
t[::7] = np.inf
d = lambda a: torch.as_tensor(a, device=dev)[None]
batch = dict(rays_o=d(ro), rays_d=d(rd), t=d(t), cone_scale=d(cone), parameters=torch.as_tensor(np.asarray([fam["params"]], np.float32), device=dev))
model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0, dense_media=True))
for prec in ("float32", "fp16x3"):
    for perturb in (False, True):
        r = Renderer(model=model, n_samples=64, perturb=perturb, precision=prec)
        first = None; n = 0; bad = 0; t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            o = r(**batch, seed=4242)
            cur = torch.cat([o["color_pred"][0], o["alpha_pred"][0][:, None]], -1)
            if first is None: first = cur.clone()
            elif not torch.equal(cur, first): bad += 1
            n += 1
        r.raise_if_nonfinite()
        print(f"SOAK render {prec} perturb={perturb}: {n} launches, {bad} differing from the first", flush=True)
