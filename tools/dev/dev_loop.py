"""dev helper (GPU box): keep one render launch running back to back for `seconds` (power / clock traces need a busy
phase longer than one launch).  usage: dev_loop.py <precision> <seconds> [weights: glorot|zero|const]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_tex_amd import synthetic
from nerf_tex_amd.model import ParamNerf
from nerf_tex_amd.renderer import Renderer

prec, seconds = sys.argv[1], float(sys.argv[2])
wname = sys.argv[3] if len(sys.argv) > 3 else "glorot"
fam = synthetic.FAMILIES["carpet"]
emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
dev = torch.device("cuda", 0)
ro, rd, t, cone = synthetic.all_hit_rays(640000, fam["b_0"], fam["b_1"], fam["cam"], seed=1)
d = lambda a: torch.as_tensor(a, device=dev)[None]
batch = dict(rays_o=d(ro), rays_d=d(rd), t=d(t), cone_scale=d(cone), parameters=torch.as_tensor(np.asarray([fam["params"]], np.float32), device=dev))
model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
blob = synthetic.synthetic_weights(model.layer_table(), seed=0)
if wname == "zero": blob = np.zeros_like(blob)
if wname == "const": blob = np.full_like(blob, 2.0 ** -6)
model.set_blob(blob)
r = Renderer(model=model, n_samples=64, perturb=False, check_numerics=False, precision=prec)
r(**batch); torch.cuda.synchronize()
time.sleep(2.0)                                  # idle baseline in the trace
t0 = time.perf_counter(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < seconds:
    r(**batch); n += 1
    if n % 4 == 0: torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
print(f"LOOP {prec} weights={wname} launches={n} ms_per_launch={e0.elapsed_time(e1) / n:.2f}", flush=True)
