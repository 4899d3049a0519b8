#!/usr/bin/env python3
"""Throughput of the NeRF-Tex render path on MI355X (BASELINE.json metric: ray-samples/sec through
PE + MLP + composite at 800x800x64).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload carpet|grass|fur|grass_filtered]

One step = one pass of the fused HIP render kernel over one 800x800 image worth of synthetic
all-hit rays (640 000 rays x 64 samples = 40.96 M ray-samples), inputs
already resident in HBM, followed (N > 1) by the one gather of the finished RGBA to rank 0.  With
N > 1 each rank renders its own 800x800 band of an (800 N) x 800 image (weak scaling, one process per
GPU, launched by torch.distributed.run); `value` is the whole-job aggregate.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant (only) kernel: algorithmic FLOPs =
2 * MACs(model) per ray-sample / average launch duration measured with HIP
events on the launch stream.  `cpu_baseline` times the float32 numpy restatement in oracle/ on a
bounded sample of the same workload on this host's cores (rank 0, N = 1 only).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_{f16,bf16} dense peak (~2.5 PF, no sparsity)

WORKLOADS = {   # name -> (family, H, W, samples per ray)
    "carpet": ("carpet", 800, 800, 64),             # BASELINE configs[1] -- the metric's configuration
    "grass": ("grass", 800, 800, 128),              # configs[2]
    "fur": ("fur", 800, 800, 64),                   # configs[3] per-GPU share when sharded
    "grass_filtered": ("grass_filtered", 800, 800, 128),
}


def cpu_baseline(family: str, n_samples: int, target_seconds: float = 12.0):
    """float32 oracle (numpy + BLAS threads) on a bounded number of rays of the same workload."""
    from oracle import nerftex_oracle as orc
    from nerf_tex_amd import synthetic
    fam = synthetic.FAMILIES[family]
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
    w = orc.split_blob(spec, synthetic.synthetic_weights(orc.layer_table(spec), seed=0))
    params = np.asarray([fam["params"]], np.float32)

    def run(n_rays):
        ro, rd, t, cone = synthetic.all_hit_rays(n_rays, fam["b_0"], fam["b_1"], fam["cam"])
        t0 = time.perf_counter()
        orc.renderer_call(w, spec, ro[None], rd[None], t[None], params, cone[None], n_samples, False, (1, 1, 1.),
                          fam["blur_idx"], False, render_chunk=32768, net_chunk=65536, dtype=np.float32)
        return time.perf_counter() - t0

    run(256)                                   # warm BLAS
    n = 2048
    dt = run(n)
    rate = n * n_samples / dt
    n2 = int(min(32768, max(n, rate * target_seconds / n_samples)))   # at most one reference render_chunk
    if n2 > n:
        dt = run(n2); n = n2
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": n * n_samples / dt, "unit": "ray-samples/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} rays x {n_samples} samples of the same workload, float32 numpy restatement (oracle/), "
                      f"reference chunking 32768/65536, {dt:.2f} s, host has {os.cpu_count()} logical cpus"}


def measured_traffic(workload: str, precision: str = "float32"):
    """HBM-side bytes per launch of the render kernel from the committed rocprofv3 PMC summary of this very
    command (separate --pmc passes, FETCH_SIZE x2 for gfx950's wide reads; tools/summarize_profile.py).
    PMC collection cannot run inside the timed bench, so the latest committed profile is quoted; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", ("bench_" if precision == "float32" else "benchx3_") + f"{workload}_*pmc_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))["derived"]
    rd, wr = d.get("hbm_side_read_bytes_corrected"), d.get("hbm_side_write_bytes_uncalibrated")
    if rd is None:
        return None, None
    return float(rd) + float(wr or 0.0), os.path.relpath(files[-1], ROOT)


def bench_instanced(args) -> None:
    """`--workload carpet_instanced`: the InstanceRenderer tail (what config_carpet_render.py runs) on one
    render chunk of synthetic instancer output resident in HBM: 16 384 rays x 1024 marching samples
    (config_carpet_render.py:78-79), 1/8 of them inside a patch (dists > 0) in runs of 16, i.e. ~128 network
    evaluations per ray after the in-kernel compaction.  value = in-patch ray-samples/s; N = 1 only (the patch instancer
    that feeds this path is CPU code outside the hot path)."""
    import torch
    from nerf_tex_amd import _lib, synthetic
    from nerf_tex_amd.model import ParamNerf
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("carpet_instanced is a single-GPU workload")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    fam = synthetic.FAMILIES["carpet"]
    emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
    model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
    model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0))
    n, S, P = 16384, 1024, model.n_params
    g = torch.Generator(device=dev); g.manual_seed(0)
    u = lambda *shape: torch.rand(*shape, device=dev, generator=g)
    rays_d_map = torch.nn.functional.normalize(u(n, S, 3) - 0.5, dim=-1).contiguous()
    pts = (u(n, S, 3) * 2.4 - 1.2).contiguous()
    t = torch.sort(u(n, S) * 6 + 2, dim=-1).values.contiguous()
    inside = (u(n, S // 16) < 0.125).repeat_interleave(16, dim=1)              # runs of 16 marching steps inside a patch
    dists = torch.where(inside, (u(n, S) * 1.5 + 0.5) * 0.002, torch.zeros((), device=dev)).contiguous()
    color_last = u(n, 3).contiguous(); alpha_last = (u(n) < 0.5).float().contiguous()
    alpha_weight = (1.0 / torch.randint(1, 4, (n, S), device=dev, generator=g)).float().contiguous()
    instance_id = torch.randint(0, 7, (n, S), device=dev, generator=g, dtype=torch.int32).contiguous()
    hit = torch.ones(n, device=dev, dtype=torch.uint8)
    params_map = (torch.as_tensor(fam["params"], device=dev, dtype=torch.float32)[None, None, :] * (u(n, S, 1) * 0.5 + 0.5)).contiguous()
    cone = (u(n) * 4e-3 + 1e-3).contiguous()
    color = torch.empty((n, 3), device=dev); alpha = torch.empty((n,), device=dev)
    n_in = int(inside.sum().item())
    stream = torch.cuda.current_stream(dev).cuda_stream

    _lib.check(_lib.lib.ntx_set_precision(model.ctx(0), _lib.PRECISIONS[args.precision]))

    def step():
        _lib.check(_lib.lib.ntx_render_instanced(
            model.ctx(0), rays_d_map.data_ptr(), pts.data_ptr(), t.data_ptr(), dists.data_ptr(), color_last.data_ptr(),
            alpha_last.data_ptr(), alpha_weight.data_ptr(), instance_id.data_ptr(), hit.data_ptr(), params_map.data_ptr(),
            cone.data_ptr(), n, S, -1, 0.09, 400.0, 0, _lib.f3([1, 1, 1.]), None, color.data_ptr(), alpha.data_ptr(), None, stream))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    flops_per_sample = 2 * model.macs_per_sample()
    achieved = n_in * flops_per_sample / (kernel_ms * 1e-3) / 1e12
    peak = F32_MFMA_PEAK_TFLOPS if args.precision == "float32" else F16_MFMA_PEAK_TFLOPS
    in_bytes = n * S * 4 * (3 + 3 + 1 + 1 + 1 + 1 + P)
    print(json.dumps({
        "metric": "in-patch ray-samples/sec (InstanceRenderer tail: compaction + MLP + composite)",
        "value": n_in * args.steps / elapsed, "unit": "ray-samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "float32" else "fp16x3 (f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"carpet_instanced: one render chunk of {n} rays x {S} marching samples of synthetic instancer "
                               f"output (config_carpet_render.py:78-98), {n_in} in-patch samples ({n_in / n:.1f} per ray, runs of 16), "
                               f"ParamNerf n_parameters={list(fam['n_parameters'])}, buffers resident in HBM",
                   "rays": n, "marching_samples_per_ray": S, "in_patch_samples": n_in, "flops_per_sample": flops_per_sample},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": None,
                     "algorithmic_bytes": in_bytes, "algorithmic_GBps": in_bytes / (kernel_ms * 1e-3) / 1e9,
                     "kernel": "ntx::instance_kernel" if args.precision == "float32" else "ntx::instance_kernel_x3",
                     "kernel_ms": kernel_ms}}), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="carpet", choices=sorted(WORKLOADS) + ["carpet_instanced"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="float32", choices=["float32", "fp16x3"],
                    help="arithmetic of the Dense layers (include/nerftex.h: ntx_precision); float32 = the reference's")
    args = ap.parse_args()
    if args.workload == "carpet_instanced":
        return bench_instanced(args)

    import torch
    import torch.distributed as dist
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.dist import gather_image
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.renderer import Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)       # nccl backend == RCCL on ROCm

    family, H, W, S = WORKLOADS[args.workload]
    fam = synthetic.FAMILIES[family]
    emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
    model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
    model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0))
    renderer = Renderer(model=model, n_samples=S, perturb=False, blur_idx=fam["blur_idx"], check_numerics=False,
                        precision=args.precision)

    n_rays = H * W                                           # per GPU (weak scaling)
    ro, rd, t, cone = synthetic.all_hit_rays(n_rays, fam["b_0"], fam["b_1"], fam["cam"], seed=1 + rank)
    d = lambda a: torch.as_tensor(a, device=dev)[None]
    batch = dict(rays_o=d(ro), rays_d=d(rd), t=d(t), cone_scale=d(cone),
                 parameters=torch.as_tensor(np.asarray([fam["params"]], np.float32), device=dev))

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev0[i].record()
        out = renderer(**batch)
        if i is not None:
            ev1[i].record()                                  # same stream the kernel was launched on
        rgba = torch.cat([out["color_pred"][0], out["alpha_pred"][0][:, None]], -1)
        if world > 1:
            return gather_image(rgba, n_rays * world)        # the one collective: RGBA -> rank 0
        return rgba

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        img = step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))

    # the opt-in fp16x3 precision on the same inputs, outside the timed region (rank 0, N = 1): a second, clearly
    # labelled figure next to the float32 headline -- never `value`
    alt = None
    if world == 1 and args.precision == "float32":
        r2 = Renderer(model=model, n_samples=S, perturb=False, blur_idx=fam["blur_idx"], check_numerics=False,
                      precision="fp16x3")
        o2 = r2(**batch)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(args.steps):
            o2 = r2(**batch)
        a1.record()
        torch.cuda.synchronize()
        ms2 = a0.elapsed_time(a1) / args.steps
        rgba2 = torch.cat([o2["color_pred"][0], o2["alpha_pred"][0][:, None]], -1)
        alt = {"precision": "fp16x3 (3-term split of weights and activations into IEEE halves on v_mfma_f32_32x32x16_f16, f32 accumulate)",
               "value": n_rays * S / (ms2 * 1e-3), "unit": "ray-samples/s", "kernel_ms": ms2,
               "rel_linf_vs_float32_kernel": float((rgba2 - img).abs().max() / img.abs().max()),
               # canonical FLOPs (2 * MACs per ray-sample) over the 16-bit dense peak; the kernel issues 3 half-precision MFMA products
               # per canonical MAC, so the matrix pipe is 3x busier than this fraction
               "canonical_frac_of_f16_peak": n_rays * S * 2 * model.macs_per_sample() / (ms2 * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
               "mfma_products_per_mac": 3}

    if rank == 0:
        samples_per_step = n_rays * S * world
        flops_per_sample = 2 * model.macs_per_sample()
        achieved = n_rays * S * flops_per_sample / (kernel_ms * 1e-3) / 1e12
        # canonical FLOPs (2 * MACs per ray-sample; split-precision multiplicity does not count) against the
        # dense peak of the issued MFMA dtype
        peak = F32_MFMA_PEAK_TFLOPS if args.precision == "float32" else F16_MFMA_PEAK_TFLOPS
        line = {
            "metric": "ray-samples/sec (MLP+composite) at 800x800x64",
            "value": samples_per_step * args.steps / elapsed,
            "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "float32" else "fp16x3 (f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{args.workload} {H}x{W}x{S}: {n_rays} all-hit rays x {S} samples per GPU "
                                   f"(BASELINE configs[{ {'carpet': 1, 'grass': 2, 'fur': 3, 'grass_filtered': 4}[args.workload] }]), "
                                   f"ParamNerf n_parameters={list(fam['n_parameters'])}, seeded glorot weights, "
                                   f"inputs resident in HBM, fused PE+MLP+composite"
                                   + (", + gather of RGBA to rank 0" if world > 1 else ""),
                       "rays_per_gpu": n_rays, "samples_per_ray": S, "flops_per_sample": flops_per_sample},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": measured_traffic(args.workload, args.precision)[0],
                         "traffic_unit": "bytes/launch (HBM side, rocprofv3 PMC)", "traffic_source": measured_traffic(args.workload, args.precision)[1],
                         "algorithmic_bytes": n_rays * (4 * (3 + 3 + 2 + 1 + 4) + 0) + 4 * model.n_params,
                         "traffic_note": "the render kernel re-reads each ray's 1 KiB direction vector (dirbias_kernel output, "
                                         f"{n_rays * 1024} B per launch) once per 32-sample batch, mostly from L2/MALL; bound is MFMA, not HBM",
                         "kernel": "ntx::render_kernel" if args.precision == "float32" else "ntx::render_kernel_x3",
                         "kernel_ms": kernel_ms},
        }
        if alt is not None:
            line["fp16x3"] = alt
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(family, S)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
