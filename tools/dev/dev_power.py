"""dev helper (GPU box): is the fp16x3 kernel's time data-dependent (power throttling)?  Times the carpet bench launch
with the seeded glorot weights, with all-zero weights, and with constant weights."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_tex_amd import synthetic
from nerf_tex_amd.model import ParamNerf
from nerf_tex_amd.renderer import Renderer

fam = synthetic.FAMILIES["carpet"]
emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
dev = torch.device("cuda", 0)
n_rays, S = 640000, 64
ro, rd, t, cone = synthetic.all_hit_rays(n_rays, fam["b_0"], fam["b_1"], fam["cam"], seed=1)
d = lambda a: torch.as_tensor(a, device=dev)[None]
batch = dict(rays_o=d(ro), rays_d=d(rd), t=d(t), cone_scale=d(cone),
             parameters=torch.as_tensor(np.asarray([fam["params"]], np.float32), device=dev))
for prec in sys.argv[1:] or ["fp16x3", "float32"]:
    for name in ("glorot", "zero", "const"):
        model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
        blob = synthetic.synthetic_weights(model.layer_table(), seed=0)
        if name == "zero":
            blob = np.zeros_like(blob)
        if name == "const":
            blob = np.full_like(blob, 2.0 ** -6)
        model.set_blob(blob)
        r = Renderer(model=model, n_samples=S, perturb=False, check_numerics=False, precision=prec)
        r(**batch); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            r(**batch)
        e1.record(); torch.cuda.synchronize()
        print(f"POWER {prec:8s} weights={name:6s} {e0.elapsed_time(e1) / 3:.2f} ms", flush=True)
