"""dev helper (GPU box): soak of ntx_render_instanced -- the same launch over and over (the ray hand-out, the chunks a wave gets and
the company in every packed batch differ from launch to launch), every image compared bit for bit with the first one.  Three
workloads: the bench's (16 384 rays x 1024, runs of 16), many short rays (40 960 x 64: mostly tails, open batches across bundles),
few very long rays (96 x 4096: the execution list slides its window).
usage: python tools/dev/soak_instanced.py [seconds per case]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nerf_tex_amd import _lib, synthetic
from nerf_tex_amd.model import ParamNerf

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
fam = synthetic.FAMILIES["carpet"]
emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
dev = torch.device("cuda", 0)
model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0, dense_media=True))
P = model.n_params
g = torch.Generator(device=dev); g.manual_seed(0)
u = lambda *shape: torch.rand(*shape, device=dev, generator=g)
for name, n, S, run, p_in in (("bench", 16384, 1024, 16, 0.125), ("short rays", 40960, 64, 4, 0.2), ("long rays", 96, 4096, 32, 0.7)):
    per_run = lambda *tail: u(n, S // run, *tail).repeat_interleave(run, dim=1)
    rays_d_map = torch.nn.functional.normalize(per_run(3) - 0.5, dim=-1).contiguous()
    pts = (u(n, S, 3) * 2.4 - 1.2).contiguous()
    t = torch.sort(u(n, S) * 6 + 2, dim=-1).values.contiguous()
    inside = (u(n, S // run) < p_in).repeat_interleave(run, dim=1)
    dists = torch.where(inside, (u(n, S) * 1.5 + 0.5) * 0.002, torch.zeros((), device=dev)).contiguous()
    color_last = u(n, 3).contiguous(); alpha_last = (u(n) < 0.5).float().contiguous()
    alpha_weight = (1.0 / torch.randint(1, 4, (n, S), device=dev, generator=g)).float().contiguous()
    hit = (u(n) < 0.97).to(torch.uint8).contiguous()
    params_map = torch.as_tensor(fam["params"], device=dev, dtype=torch.float32)[None, None, :].repeat(n, S, 1)
    params_map[..., :1] *= u(n, S, 1) * 0.5 + 0.5
    params_map[..., 1:] *= per_run(1) * 0.5 + 0.5
    params_map = params_map.contiguous()
    cone = (u(n) * 4e-3 + 1e-3).contiguous()
    model.reserve(0, n)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for prec in ("float32", "fp16x3"):
        first = None; k = 0; bad = 0; t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            color = torch.full((n, 3), float("nan"), device=dev); alpha = torch.full((n,), float("nan"), device=dev)
            _lib.check(_lib.lib.ntx_render_instanced(
                model.ctx(0), rays_d_map.data_ptr(), pts.data_ptr(), t.data_ptr(), dists.data_ptr(), color_last.data_ptr(),
                alpha_last.data_ptr(), alpha_weight.data_ptr(), None, hit.data_ptr(), params_map.data_ptr(), cone.data_ptr(),
                n, S, -1, 0.09, 400.0, _lib.PRECISIONS[prec], _lib.f3([1, 1, 1.]), None, None, color.data_ptr(), alpha.data_ptr(), None, stream))
            cur = torch.cat([color, alpha[:, None]], -1)
            if first is None:
                first = cur.clone()
                assert bool(torch.isfinite(first).all())
            elif not torch.equal(cur, first):
                bad += 1
            k += 1
        print(f"SOAK instanced {name} ({n} x {S}) {prec}: {k} launches, {bad} differing from the first", flush=True)
