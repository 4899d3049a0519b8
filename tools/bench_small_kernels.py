#!/usr/bin/env python3
"""HBM roofline of the small kernels around the fused render kernel: achieved algorithmic GB/s of
ntx_generate_rays, ntx_composite, ntx_sample_pdf, ntx_sample_depths, ntx_image_epilogue, at
BASELINE sizes, through the C ABI with HIP events.  GPU box only:  python tools/bench_small_kernels.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_tex_amd import _lib, synthetic                      # noqa: E402
from nerf_tex_amd.proxy import AABB                           # noqa: E402
from nerf_tex_amd.ray_sampler import Proxy                    # noqa: E402
from nerf_tex_amd.render import image_epilogue                # noqa: E402
from nerf_tex_amd.renderer import Renderer                    # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md
dev = torch.device("cuda", 0)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, nbytes, note):
    gbps = nbytes / (ms * 1e-3) / 1e9
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "GBps": round(gbps, 1),
                      "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4), "note": note}), flush=True)


fam = synthetic.FAMILIES["carpet"]
H = W = 1600
n = H * W
# ray generation: 36 B written per ray (rays_o 12, rays_d 12, t 8, cone 4)
c2w = np.eye(4, dtype=np.float32); c2w[:3, 3] = fam["cam"]
ps = Proxy(H, W, 1400.0, AABB(fam["b_0"], fam["b_1"]))
report("raygen_kernel", timed(lambda: ps((0, n), c2w, device=dev)), 36 * n, f"{H}x{W} pixels, 36 B/ray out (includes 4 torch.empty)")

# stand-alone composite: color[N,S,3] + sigma[N,S] + z[N,S] in, RGBA + weights[N,S] out
N, S = 640000, 64
r = Renderer(model=None, n_samples=S, perturb=False)
color = torch.randn((N, S, 3), device=dev); sigma = torch.randn((N, S), device=dev) * 10
z = torch.sort(torch.rand((N, S), device=dev) * 4 + 2, -1).values.contiguous()
rd = torch.nn.functional.normalize(torch.randn((N, 3), device=dev), dim=-1)
report("composite_kernel", timed(lambda: r.map_model_output(color, sigma, z, rd, False, [1, 1, 1.])),
       N * S * (12 + 4 + 4 + 4) + N * (12 + 16), f"{N} rays x {S}: 20 B/sample in, 4 B/sample weights out")

# sample_pdf: weights[N,S] + t in, z_all[N,S+NI] out
NI = 64
t = torch.stack([torch.full((N,), 2.0, device=dev), torch.full((N,), 6.0, device=dev)], -1).contiguous()
wts = torch.rand((N, S), device=dev)
z_all = torch.empty((N, S + NI), device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
report("sample_pdf_kernel", timed(lambda: _lib.check(_lib.lib.ntx_sample_pdf(t.data_ptr(), None, wts.data_ptr(), None, N, S, NI, 0, 0, None, z_all.data_ptr(), st))),
       N * (S * 4 + (S + NI) * 4 + 8), f"{N} rays, {S} coarse + {NI} importance depths (deterministic u)")

# image epilogue: RGBA in, float32 + uint8 out, with and without the gaussian downsample
img = torch.rand((H, W, 4), device=dev)
report("epilogue_kernel f=1", timed(lambda: image_epilogue(img, 1, uint8=True)), H * W * (16 + 16 + 4), f"{H}x{W}, un-premultiply + uint8")
report("epilogue_kernel f=2", timed(lambda: image_epilogue(img, 2, uint8=True)), H * W * 16 + (H // 2) * (W // 2) * 20, f"{H}x{W} -> /2, 6x6 gaussian")

# sample depths with the in-kernel jitter: t in, z[N,S] out (one Philox4x32-10 block per depth)
zj = Renderer.sample_depths(t, S, perturb=True, seed=1)
report("sample_depths_kernel perturb", timed(lambda: Renderer.sample_depths(t, S, perturb=True, seed=1)), N * (S * 4 + 8), f"{N} rays x {S} jittered depths (includes torch.empty)")
