#!/usr/bin/env python3
"""Throughput of the NeRF-Tex render path on MI355X (BASELINE.json metric: ray-samples/sec through
PE + MLP + composite at 800x800x64).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--precision float32|fp16x3] [--perturb]

`--gpus N` works as typed: when no launcher has set WORLD_SIZE, bench.py starts its N ranks itself (one process per
GPU, rendezvous on 127.0.0.1); under `python -m torch.distributed.run --nproc-per-node N` it uses the launcher's ranks.

Workloads
  carpet | grass | fur | grass_filtered   (weak scaling; `carpet` = BASELINE configs[1] is the default and the metric's
      configuration) one step = one pass of the fused HIP render kernel over one 800x800 image worth of synthetic ALL-HIT
      rays per GPU (640 000 rays x S samples), inputs resident in HBM.  N > 1: every rank renders
      its own 800x800 band of an (800 N) x 800 image, then the one gather of the finished RGBA to rank 0.
  fur_sharded | grass_filtered_sharded    (strong scaling; BASELINE configs[3] / configs[4]) ONE image -- fur 800x800x64 /
      grass_filtered 1600x1600x128 with the cone-filter conditioning -- from the config's true camera, rays generated on
      the device, pixel rows dealt round-robin over the N ranks (`--shard rows`, balances the rays the proxy culls; or
      `--shard bands`), gathered to rank 0 through `ntx_gather_image` (RCCL ncclGather).  value = rays that hit the proxy
      x S / time.  After the timed region rank 0 renders the whole image alone and the line reports whether the
      gathered image is bit-identical to it.
  carpet_instanced                        the InstanceRenderer tail on synthetic instancer buffers, N = 1.
  carpet_instanced_scene                  what the shipped render configs run per chunk, from rays: the patch instancer on the GPU
      (`ntx_instancer_model_input`) on a synthetic scene in the carpet config's shape, then the tail on ITS output; N = 1.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel: algorithmic FLOPs = 2 * MACs(model) per
ray-sample / average launch duration measured with HIP events on the launch stream.
`cpu_baseline` times the torch-CPU float32 restatement in oracle/ on a bounded sample of the same workload on this
host's cores (rank 0, N = 1 only).  `parity` is the second half of the metric: rel-Linf of 256 seeded rays of the image the
timed region just produced against the float64 and float32 restatements (oracle/, the checker).  `with_ray_setup` repeats the
step with ray generation (`ntx_generate_rays`) inside the timed loop ("with and without ray setup").
N > 1 adds `per_rank` (kernel / gather time, rays, hits of every rank), `gather_bytes`, `imbalance` and an efficiency figure
against rank 0 re-timed alone after the barrier.

A multi-rank run is BOUNDED: the self-launcher polls its ranks, ends the others as soon as one fails, enforces an overall
deadline (`--deadline`, default 600 s) and keeps every rank's stderr in `logs/rank<r>.err` (tail echoed on failure); under an
external launcher every rank arms a watchdog that dumps its stacks and exits after the same deadline, so a rank stuck in a
collective cannot hold the node.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_{f16,bf16} dense peak (~2.5 PF, no sparsity)

WORKLOADS = {   # name -> (family, H, W, samples per ray, BASELINE config index)
    "carpet": ("carpet", 800, 800, 64, 1),             # the metric's configuration
    "grass": ("grass", 800, 800, 128, 2),
    "fur": ("fur", 800, 800, 64, 3),                   # per-GPU share of config 3 as all-hit rays
    "grass_filtered": ("grass_filtered", 800, 800, 128, 4),
}
SHARDED = {     # one image from the true camera, sharded over the ranks
    "fur_sharded": ("fur", 800, 800, 64, 3),
    "grass_filtered_sharded": ("grass_filtered", 1600, 1600, 128, 4),
}


def cpu_baseline(family: str, n_samples: int, target_seconds: float = 15.0):
    """The reference's CPU path as a float32 torch-CPU port (oracle/torch_cpu.py: MatMul/BiasAdd/Relu per Dense layer on
    the host BLAS, vectorised sin/cos, reference chunking 32768/65536), timed on up to one full render_chunk
    (32 768 rays) of the same workload, bounded to about `target_seconds`."""
    import torch
    from oracle import nerftex_oracle as orc
    from oracle import torch_cpu
    from nerf_tex_amd import synthetic
    fam = synthetic.FAMILIES[family]
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
    w = orc.split_blob(spec, synthetic.synthetic_weights(orc.layer_table(spec), seed=0))
    params = np.asarray(fam["params"], np.float32)

    def run(n_rays):
        ro, rd, t, cone = synthetic.all_hit_rays(n_rays, fam["b_0"], fam["b_1"], fam["cam"])
        t0 = time.perf_counter()
        torch_cpu.renderer_call(w, spec, ro, rd, t, params, cone, n_samples, fam["blur_idx"], render_chunk=32768, net_chunk=65536)
        return time.perf_counter() - t0

    cores = torch_cpu.effective_cpus()         # affinity mask capped by the cgroup CPU quota of this container
    torch.set_num_threads(cores)
    run(1024)                                  # warm the BLAS threads
    n = 4096
    dt = run(n)
    rate = n * n_samples / dt
    n2 = int(min(32768, max(n, rate * target_seconds / n_samples)))   # at most one reference render_chunk
    if n2 > n:
        dt = run(n2); n = n2
    return {"value": n * n_samples / dt, "unit": "ray-samples/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "cpu_count": os.cpu_count(), "cpu_quota": cores, "blas": torch_cpu.blas_backend(),
            "sample": f"{n} rays x {n_samples} samples of the same workload ({'one full' if n == 32768 else 'part of a'} "
                      f"reference render_chunk), float32 torch-CPU port of the reference's TF ops (oracle/torch_cpu.py), "
                      f"reference chunking 32768/65536, {dt:.2f} s on {torch.get_num_threads()} threads (host: {os.cpu_count()} logical CPUs, "
                      f"this container may use {cores})"}


def measured_traffic(workload: str, precision: str = "float32"):
    """HBM-side bytes per ntx_render_rays call (ALL its kernels: hit compaction + render) from the committed rocprofv3 PMC
    summary of this very command (separate --pmc passes, FETCH_SIZE x2 for gfx950's wide reads; tools/summarize_profile.py).
    PMC collection cannot run inside the timed bench, so the latest committed profile is quoted; None if absent.
    Returns (bytes, file, profile) with profile = {"git_head", "kernel_sources_sha16", "current"}: the tree the profile was taken on and
    whether the kernel sources of THIS tree hash the same (None: the profile predates the record)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", ("bench_" if precision == "float32" else "benchx3_") + f"{workload}_v[0-9]*pmc_summary.json")))
    if not files:
        return None, None, None
    d = json.load(open(files[-1]))["derived"]
    rd = d.get("call_hbm_side_read_bytes_corrected", d.get("hbm_side_read_bytes_corrected"))
    wr = d.get("call_hbm_side_write_bytes_uncalibrated", d.get("hbm_side_write_bytes_uncalibrated"))
    if rd is None:
        return None, None, None
    prof = {"git_head": d.get("git_head"), "kernel_sources_sha16": d.get("kernel_sources_sha16"), "current": None}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import kernel_sources_sha16
        if prof["kernel_sources_sha16"]:
            prof["current"] = prof["kernel_sources_sha16"] == kernel_sources_sha16(ROOT)
    except Exception:
        pass
    if prof["current"] is not True:
        print(f"bench.py: roofline.traffic is quoted from {os.path.relpath(files[-1], ROOT)} (taken at {prof['git_head'] or 'an unrecorded commit'}); "
              + ("the kernel sources have changed since" if prof["current"] is False else "whether the kernels changed since is not recorded")
              + " -- re-run tools/profile_bench.sh + tools/summarize_profile.py", file=sys.stderr)
    return float(rd) + float(wr or 0.0), os.path.relpath(files[-1], ROOT), prof


def train_step_traffic():
    """HBM-side bytes of ONE training step (every kernel of the trainer: WRITE_SIZE + 2 x FETCH_SIZE per launch x launches a step) from the newest
    committed rocprofv3 summary of `bench.py --workload carpet_train_step` (profiles/r*/train_step_pmc_summary.json, tools/dev/r5_train_profiles.sh),
    quoted like `measured_traffic` quotes the render kernel's."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "train_step_pmc_summary.json")))
    if not files:
        return {"traffic": None, "traffic_source": None}
    d = json.load(open(files[-1]))
    total = sum((float(k.get("hbm_read_MB", 0.0)) + float(k.get("hbm_written_MB", 0.0))) * 1e6 * float(k.get("launches_per_5_steps") or 0) / 5.0 for k in d["kernels"].values())
    tree = str(d.get("tree", "")).split()
    cur = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import kernel_sources_sha16
        if tree:
            cur = tree[0] == kernel_sources_sha16(ROOT)
    except Exception:
        pass
    return {"traffic": total, "traffic_unit": "bytes per training step, all kernels (HBM side, rocprofv3 PMC)", "traffic_source": os.path.relpath(files[-1], ROOT),
            "traffic_profile_head": tree[1] if len(tree) > 1 and tree[1] != "None" else None, "traffic_profile_current": cur}


def train_step_algorithmic_bytes(model, n_samples_total: int) -> int:
    """HBM bytes a stored-activation training step cannot do without, per the layout of ntx_train_device.h: per sample, the eleven stored layer
    outputs (10 x 256 + 128 floats) and the eleven stored gradients written once and read once by the weight gradients; pos_map / dir_map
    (padded to tiles of 32 rows) written once, read by the chain (layer 0, the skip, the colour layer's concatenation) and by the weight
    gradients (pos_map twice each: layer 0 and the skip); the heads' few floats; weights and moments are 2.7 MB and do not count."""
    kp = -(-model.pos_map_dim // 32) * 32
    kd = -(-model.dir_map_dim // 32) * 32
    acts = 10 * 256 + 128
    per_sample = 4 * (2 * acts                       # forward outputs: written, read by dW
                      + 2 * acts                     # gradients at them: written, read by dW
                      + (kp + kd) + (2 * kp + kd) + (2 * kp + kd)          # encodings: written, read by the chain (pos_map twice), read by dW (pos_map twice)
                      + 4 + 4 + 4 + 32 + 32)         # raw rgb / sigma written and read, the composite's adjoint written (plain and as a tile) and read
    return int(per_sample * n_samples_total)


def instancer_traffic():
    """HBM-side bytes of one ntx_instancer_model_input call on the bench scene (WRITE_SIZE + 2 x FETCH_SIZE of its three kernels) from the
    newest committed rocprofv3 summary of `tools/bench_instancer.py` (profiles/r*/instancer_base_pmc_summary.json), quoted like
    `measured_traffic` quotes the render kernel's."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "instancer_base_pmc_summary.json")))
    if not files:
        return {"traffic": None, "traffic_source": None}
    d = json.load(open(files[-1]))
    total = sum(float(k.get("hbm_side_write_bytes", 0.0)) + float(k.get("hbm_side_read_bytes_corrected", 0.0)) for k in d["kernels"].values())
    cur = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import kernel_sources_sha16
        cur = d.get("kernel_sources_sha16") == kernel_sources_sha16(ROOT) if d.get("kernel_sources_sha16") else None
    except Exception:
        pass
    return {"traffic": total, "traffic_source": os.path.relpath(files[-1], ROOT) + " (WRITE_SIZE + 2 x FETCH_SIZE of the three kernels, same scene)",
            "traffic_profile_head": d.get("git_head"), "traffic_profile_current": cur}


def bench_instanced(args) -> None:
    """`--workload carpet_instanced`: the InstanceRenderer tail (what config_carpet_render.py runs) on one
    render chunk of synthetic instancer output resident in HBM: 16 384 rays x 1024 marching samples
    (config_carpet_render.py:78-79), 1/8 of them inside a patch (dists > 0) in runs of 16, i.e. ~128 network
    evaluations per ray after the in-kernel compaction.  As the reference's instancer fills its buffers
    (instancer/src/instancer.cpp:889-960), the direction in the patch frame, the light direction and the other appearance
    parameters are constant along a run (one patch instance: getDir(ray, instance)), while the position and the texture-mapped
    geometry parameter change from sample to sample (`--instanced-per-sample-dirs`: the round-2 workload, every sample its own
    direction and parameters).  value = in-patch ray-samples/s; N = 1 only.  (`carpet_instanced_scene` runs the same tail on the
    output of this package's own patch instancer.)"""
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
    n, S, P = int(os.environ.get("NTX_INSTANCED_RAYS", "16384")), 1024, model.n_params   # (the env knob is for scaling experiments)
    g = torch.Generator(device=dev); g.manual_seed(0)
    u = lambda *shape: torch.rand(*shape, device=dev, generator=g)
    per_run = (lambda *tail: u(n, S // 16, *tail).repeat_interleave(16, dim=1)) if not args.instanced_per_sample_dirs else (lambda *tail: u(n, S, *tail))
    rays_d_map = torch.nn.functional.normalize(per_run(3) - 0.5, dim=-1).contiguous()
    pts = (u(n, S, 3) * 2.4 - 1.2).contiguous()
    t = torch.sort(u(n, S) * 6 + 2, dim=-1).values.contiguous()
    inside = (u(n, S // 16) < 0.125).repeat_interleave(16, dim=1)              # runs of 16 marching steps inside a patch
    dists = torch.where(inside, (u(n, S) * 1.5 + 0.5) * 0.002, torch.zeros((), device=dev)).contiguous()
    color_last = u(n, 3).contiguous(); alpha_last = (u(n) < 0.5).float().contiguous()
    alpha_weight = (1.0 / torch.randint(1, 4, (n, S), device=dev, generator=g)).float().contiguous()
    instance_id = torch.randint(0, 7, (n, S), device=dev, generator=g, dtype=torch.int32).contiguous()
    hit = torch.ones(n, device=dev, dtype=torch.uint8)
    params_map = torch.as_tensor(fam["params"], device=dev, dtype=torch.float32)[None, None, :].repeat(n, S, 1)
    g_ = fam["n_parameters"][0]
    params_map[..., :g_] *= u(n, S, 1) * 0.5 + 0.5                 # geometry (fibre length: texture-mapped, instancer.cpp:913-918): per sample
    params_map[..., g_:] *= per_run(1) * 0.5 + 0.5                 # appearance (colour, light direction in the patch frame): per run
    params_map = params_map.contiguous()
    cone = (u(n) * 4e-3 + 1e-3).contiguous()
    color = torch.empty((n, 3), device=dev); alpha = torch.empty((n,), device=dev)
    n_in = int(inside.sum().item())
    stream = torch.cuda.current_stream(dev).cuda_stream
    flags = _lib.PRECISIONS[args.precision]

    def step():
        _lib.check(_lib.lib.ntx_render_instanced(
            model.ctx(0), rays_d_map.data_ptr(), pts.data_ptr(), t.data_ptr(), dists.data_ptr(), color_last.data_ptr(),
            alpha_last.data_ptr(), alpha_weight.data_ptr(), instance_id.data_ptr(), hit.data_ptr(), params_map.data_ptr(),
            cone.data_ptr(), n, S, -1, 0.09, 400.0, flags, _lib.f3([1, 1, 1.]), None, None, color.data_ptr(), alpha.data_ptr(), None, stream))

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
    traffic_ = measured_traffic("instanced", args.precision)
    print(json.dumps({
        "metric": "in-patch ray-samples/sec (InstanceRenderer tail: compaction + MLP + composite)",
        "value": n_in * args.steps / elapsed, "unit": "ray-samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "float32" else "fp16x3 (f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"carpet_instanced: one render chunk of {n} rays x {S} marching samples of synthetic instancer "
                               f"output (config_carpet_render.py:78-98), {n_in} in-patch samples ({n_in / n:.1f} per ray, runs of 16; "
                               + ("direction / appearance parameters per SAMPLE" if args.instanced_per_sample_dirs else
                                  "direction / appearance parameters per run = per patch instance, geometry parameter and position per sample, as instancer.cpp:889-960 fills them")
                               + f"), "
                               f"ParamNerf n_parameters={list(fam['n_parameters'])}, buffers resident in HBM",
                   "rays": n, "marching_samples_per_ray": S, "in_patch_samples": n_in, "flops_per_sample": flops_per_sample},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": traffic_[0], "traffic_source": traffic_[1], "traffic_profile_head": (traffic_[2] or {}).get("git_head"),
                     "traffic_profile_current": (traffic_[2] or {}).get("current"),
                     "algorithmic_bytes": in_bytes, "algorithmic_GBps": in_bytes / (kernel_ms * 1e-3) / 1e9,
                     "kernel": "ntx::instance_kernel" if args.precision == "float32" else "ntx::instance_kernel_x3",
                     "kernel_ms": kernel_ms}}), flush=True)


def train_needed_macs(model) -> int:
    """MACs per ray-sample a training step NEEDS: the forward pass, the weight gradients (one contraction of the forward's size per layer) and
    the gradients at the layers' INPUTS -- of which the encoded inputs' are not wanted: layer 0's (pos_map rows), the skip's pos_map columns and
    the colour layer's dir_map columns (model.py:104-115; 256 outputs each).  carpet: 2 x 680 832 + 623 232."""
    m = model.macs_per_sample()
    return 2 * m + (m - (2 * model.pos_map_dim + model.dir_map_dim) * 256)


def bench_train_step(args, emit: bool = True, data_side: bool = True, parity_rays: int = 64):
    """`--workload carpet_train_step`: one iteration of the reference's training loop (network/train.py:61-67) at the batch of
    configs/config_carpet_train.py -- 4 images x 256 rays x 256 samples = 262 144 ray-samples (:23, 33, 101), perturb=True, AlphaLoss with smape /
    mse (:95-99), Adam under ExponentialDecay (lrate 5e-4, lrate_decay 500) -- forward with every activation kept, loss, backward, optimiser
    step, on `ntx_trainer_*`.  value = ray-samples/s through a whole step; roofline: 3 x the forward's canonical FLOPs
    (2 MACs per weight per sample; the backward pass is two contractions of the forward's size per layer) against the f32 MFMA
    peak.  cpu_baseline: the same step as float32 torch autograd on the host (oracle/train_oracle.py), a bounded sample.
    N > 1: data parallel, weak scaling -- every rank its own 1024 rays, ONE collective a step: the mean of the 2.7 MB of gradients over the
    ranks (`Trainer.sync_gradients`: ncclAllReduce behind the C ABI), then the same Adam step everywhere; value = all ranks' ray-samples over
    the slowest rank's time."""
    import torch
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.loss import AlphaLoss
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.train import Trainer
    import torch.distributed as dist
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    json_out = None
    if emit:
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")                 # RCCL prints its banner on stdout: the line goes to the original one
        os.dup2(2, 1)
    share_gpu = world > 1 and os.environ.get("NTX_BENCH_SHARE_GPU") == "1"          # development: all ranks on GPU 0, gloo (see main)
    if share_gpu:
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    comm, how = None, None
    if world > 1:
        import faulthandler
        from nerf_tex_amd.dist import Comm
        faulthandler.dump_traceback_later(args.deadline, exit=True)
        if share_gpu:
            dist.init_process_group("gloo")
            how = "torch.distributed (gloo) through host memory: NTX_BENCH_SHARE_GPU=1, all ranks on one GPU -- timings are not a measurement"
        else:
            dist.init_process_group("nccl", device_id=dev)
            err = None
            try:
                comm = Comm(local_rank)
            except Exception as e:                           # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            flag = torch.tensor([0 if err is None else 1], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                if comm is not None:
                    comm.close()
                comm = None
                how = "torch.distributed all_reduce through host memory (fallback: no ntx_comm on some rank" + (f"; here: {err})" if err else ")")
                print("bench.py: " + how, file=sys.stderr)
            else:
                how = f"ntx_trainer_allreduce_gradients (one ncclAllReduce through the C ABI, {comm.library})"
    fam_name = args.workload[:-len("_train_step")]              # carpet (the default and the quoted line), grass_filtered, fur: the shipped training configs' families
    fam = synthetic.FAMILIES[fam_name]
    emb = lambda n_: {"module": "network.model.FourierFeatures", "n_freq_bands": n_}
    model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
    model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0, dense_media=True))
    n, S = 4 * 256, 256
    box = ([-1.5, -1.3, -.2], [1.3, 1.3, 1.9]) if fam_name == "carpet" else (fam["b_0"], fam["b_1"])                    # config_carpet_train.py:28-31; config_<family>_train.py:29-30
    ro, rd, t, cone = synthetic.all_hit_rays(n, box[0], box[1], fam["cam"], seed=1 + rank)
    rng = np.random.default_rng(3 + rank)
    params = np.tile(np.asarray([fam["params"]], np.float32), (n, 1))
    color = rng.uniform(0, 1, size=(n, 3)).astype(np.float32)
    alpha = ((rng.uniform(0, 1, size=n) > 0.3) * rng.uniform(0.5, 1, size=n)).astype(np.float32)
    d = lambda a: torch.as_tensor(a, device=dev)
    batch = [d(x) for x in (ro, rd, t, params, cone, color, alpha)]
    loss = AlphaLoss(loss_fn="network.loss.smape", alpha_loss_fn="network.loss.mse")
    noise_std = 0.1 if fam_name == "grass_filtered" else 0.0     # config_grass_filtered_train.py:96-102: blur_idx 0, raw_noise_std 0.1
    tr = Trainer(model, max_rays=n, n_samples=S, lrate=5e-4, lrate_decay=500, perturb=True, blur_idx=fam["blur_idx"], raw_noise_std=noise_std)
    for _ in range(args.warmup):
        tr.step(*batch, loss, comm=comm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); val = tr.step(*batch, loss, comm=comm); b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    in_step = True
    if world > 1:                                            # the slowest rank's clock; do the ranks hold the same weights after the same steps?
        sdev = torch.device("cpu") if share_gpu else dev
        tmax = torch.tensor([elapsed, step_ms], dtype=torch.float64, device=sdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, step_ms = float(tmax[0]), float(tmax[1])
        w = torch.from_numpy(tr.weights()).to(sdev)
        ws = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        in_step = all(bool(torch.equal(ws[0], x)) for x in ws)
        if rank != 0:
            return
    flops_fwd = 2 * model.macs_per_sample()
    flops_needed = 2 * train_needed_macs(model)
    achieved = flops_needed * n * S / (step_ms * 1e-3) / 1e12
    achieved_3x = 3 * flops_fwd * n * S / (step_ms * 1e-3) / 1e12
    line = {"metric": "ray-samples/sec through one training step (forward + loss + backward + Adam) at 4 x 256 rays x 256 samples",
            "value": world * n * S * args.steps / elapsed, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{fam_name}_train_step: config_{fam_name}_train.py's batch ({n} all-hit rays x {S} samples = {n * S} ray-samples), ParamNerf "
                                   f"n_parameters={list(fam['n_parameters'])}, perturb=True" + (f", blur_idx={fam['blur_idx']}" if fam["blur_idx"] is not None else "")
                                   + (f", raw_noise_std={noise_std}" if noise_std else "") + ", AlphaLoss(smape, mse), Adam + ExponentialDecay(5e-4, 5e5 steps, 0.1); "
                                   "seeded weights and targets, batch resident in HBM", "rays": n, "samples_per_ray": S, "flops_per_sample_forward": flops_fwd,
                       "flops_per_sample_step": flops_needed, "loss_after": float(val.item())},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                         **(train_step_traffic() if fam_name == "carpet" else {"traffic": None, "traffic_source": None}),     # (the counter summary is the carpet step's)
                         "algorithmic_bytes": train_step_algorithmic_bytes(model, n * S),
                         "algorithmic_bytes_what": "a stored-activation step: every layer's output and the gradient at it written once (O layout) and read once by the weight "
                                                   "gradients, the encoded inputs written in both orders and read by the chain and by the weight gradients (DESIGN section 10)",
                         "kernel": "ntx_train::fwd_chain_kernel + dx_chain_kernel (the network forward and back with a block's activations in registers from layer to layer) "
                                   "+ dw_kernel (dW of every layer from the operand-order stores, persistent workgroups with an equal share each)", "kernel_ms": step_ms,
                         "what": "the FLOPs a step needs (2 MACs per weight per sample forward, the same for the weight gradients, and the gradients at the layers' "
                                 "inputs except the encoded ones': 2 x forward + forward less the pos_map / dir_map rows) over the WHOLE step's HIP-event time: "
                                 "encoders, heads, composite, loss and Adam included",
                         "frac_3x_forward": achieved_3x / F32_MFMA_PEAK_TFLOPS,
                         "frac_3x_forward_what": "the same time against 3 x the forward's FLOPs (rounds 4 and 5 quoted this convention: it counts input gradients nobody needs)"}}
    if world == 1 and parity_rays and not args.no_parity:
        # the step against the float64 restatement, outside the timed region: the first `parity_rays` rays of the batch as a batch of their own,
        # from the weights the timed steps left -- loss, predictions, every layer's gradient (the oracle follows the float32 ReLU branches)
        from oracle import nerftex_oracle as orc
        from oracle import train_oracle as tro
        t1 = time.perf_counter()
        nb = parity_rays
        spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
        tr.perturb = False
        pv, pc, pa = tr.gradients_step(*[x[:nb].contiguous() for x in batch], loss, seed=1)
        torch.cuda.synchronize()
        got = tr.gradients()
        wts = orc.split_blob(spec, tr.weights())
        z = orc.z_values(t[:nb], S, np.float32)
        M = nb * S
        masks = [(tr.activation(k, M) > 0) for k in list(range(8)) + [8, 9]]
        noise = noise_std * orc.noise_normals(nb, S, 1, dtype=np.float32).astype(np.float64) if noise_std > 0 else None
        sigma_mask = ((tr.activation(10, M).reshape(nb, S) + (0 if noise is None else noise.astype(np.float32))) > 0)
        wv, wc, wa, wg = tro.step_gradients_chunked(wts, spec, ro[:nb], rd[:nb], z, params[:nb], cone[:nb], color[:nb], alpha[:nb],
                                                    dict(kind="alpha", loss_fn="smape", alpha_loss_fn="mse"), chunk_rays=16, masks=masks, sigma_mask=sigma_mask,
                                                    noise=noise, blur_idx=fam["blur_idx"])
        flat = np.concatenate([g.ravel() for g in wg])
        worst, worst_name, p = 0.0, None, 0
        for name_, i_, o_ in orc.layer_table(spec):              # per Dense layer: kernel and bias together, relative to the layer's largest entry
            sl = slice(p, p + i_ * o_ + o_); p += i_ * o_ + o_
            e_ = float(np.abs(got[sl] - flat[sl]).max() / max(np.abs(flat[sl]).max(), 1e-30))
            if e_ > worst: worst, worst_name = e_, name_
        e_pred = float(orc.rel_linf(np.concatenate([pc.cpu().numpy(), pa.cpu().numpy()[:, None]], -1), np.concatenate([wc, wa[:, None]], -1)))
        e_loss = abs(float(pv.item()) - wv) / abs(wv)
        tr.perturb = True
        line["parity"] = {"rays": nb, "rel_loss": e_loss, "rel_linf_predictions": e_pred, "rel_linf_worst_layer_gradient": worst, "worst_layer": worst_name, "tolerance": 1e-4,
                          "ok": bool(e_loss <= 1e-4 and e_pred <= 1e-4 and worst <= 1e-4),
                          "what": f"{nb} rays x {S} samples of the batch as a step of their own (perturb off) from the weights the timed steps left: loss, [color, alpha] "
                                  "and every Dense layer's gradient (kernel and bias, relative to the layer's largest entry) against float64 torch autograd of the restated renderer and loss (oracle/train_oracle.py; "
                                  "unpinned: no TensorFlow here)", "oracle_seconds": round(time.perf_counter() - t1, 2)}
    if world == 1 and fam_name == "carpet" and data_side and not args.no_extras:
        # the LOOP around the step (train.py:60-67): batches made by nerf_tex_amd.dataset.Dataset as the config asks -- Proxy pixel sampler among the
        # proxy's hits, rays, colours gathered from resident 800 x 800 uint8 views -- and handed to the same step (tools/bench_train_loop.py)
        from nerf_tex_amd import dataset as D, util
        vrng = np.random.default_rng(0)
        views = [{"pose": D.look_at(np.asarray([np.cos(a) * 0.8, np.sin(a) * 0.8, 0.6]) * 5), "parameters": list(fam["params"]),
                  "rgba": vrng.integers(0, 256, (800, 800, 4), dtype=np.uint8)} for a in np.linspace(0, 2 * np.pi, 32, endpoint=False)]
        ds = util.instantiate({"module": "network.dataset.Dataset", "data_loader_config": {"module": "nerf_tex_amd.dataset.FromViews", "views": views, "angle": 0.63},
                               "pixel_sampler_config": {"module": "network.pixel_sampler.Proxy", "n_samples": 256}, "ray_sampler_config": {"module": "network.ray_sampler.Proxy"},
                               "proxy_config": {"module": "network.proxy.AABB", "b_0": box[0], "b_1": box[1]}, "batchsize": 4, "shuffle_buffer_size": 100, "device": dev, "seed": 0})
        it = iter(ds)
        for _ in range(max(args.warmup, 9)):                                   # every view comes up once: its image and hit list go to the device
            tr.train_step(next(it), loss)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr.train_step(next(it), loss)
        torch.cuda.synchronize()
        loop_s = (time.perf_counter() - t1) / args.steps
        line["with_data_side"] = {"ms_per_step": loop_s * 1e3, "value": n * S / loop_s, "unit": "ray-samples/s", "over_the_step_alone": loop_s / (elapsed / args.steps),
                                  "what": "the same step fed by nerf_tex_amd.dataset.Dataset (config_carpet_train.py's train_dataset_config over 32 resident 800 x 800 uint8 views: "
                                          "256 pixels a view among the proxy's hits, their rays, premultiplied colours gathered at them), batches made on the step's stream"}
    if not args.no_cpu_baseline and world == 1:
        from oracle import nerftex_oracle as orc
        from oracle import torch_cpu, train_oracle as tro
        cores = torch_cpu.effective_cpus()
        torch.set_num_threads(cores)
        spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
        w = orc.split_blob(spec, model.get_blob())
        nb = 64                                                                  # 64 rays x 256 samples of the same batch: a bounded sample
        z = orc.z_values(t[:nb], S, np.float32)
        okw = dict(kind="alpha", loss_fn="smape", alpha_loss_fn="mse")
        tro.step_gradients(w, spec, ro[:8], rd[:8], z[:8], params[:8], cone[:8], color[:8], alpha[:8], okw, dtype=torch.float32)       # warm the BLAS threads
        t1 = time.perf_counter()
        tro.step_gradients(w, spec, ro[:nb], rd[:nb], z, params[:nb], cone[:nb], color[:nb], alpha[:nb], okw, dtype=torch.float32)
        dt = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": nb * S / dt, "unit": "ray-samples/s", "cores": int(torch.get_num_threads()), "kind": "port",
                                "sample": f"{nb} rays x {S} samples of the same batch: forward + loss + torch autograd backward in float32 (oracle/train_oracle.py, no optimiser "
                                          f"step), {dt:.2f} s on {torch.get_num_threads()} threads"}
    if world > 1:
        line["allreduce_how"] = how
        line["allreduce_bytes"] = int(tr.n_weights) * 4
        line["ranks_hold_identical_weights"] = in_step
        line["config"]["workload"] += f"; data parallel over {world} GPUs: every rank its own {n} rays, gradients averaged once a step" + (
            " [NTX_BENCH_SHARE_GPU=1: all ranks on one GPU]" if share_gpu else "")
    del tr
    if emit:
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    return line


def bench_instanced_scene(args, emit: bool = True, parity_rays: int = 48):
    """`--workload carpet_instanced_scene`: one render chunk of configs/config_carpet_render.py from RAYS -- 16 384 rays of the
    config's first camera -> `ntx_instancer_model_input` (the reference's C_Instancer::GetModelInput, instancer.cpp:751-1037:
    Embree on one CPU thread there, three HIP kernels here) -> `ntx_render_instanced` on the ten buffers it leaves in
    HBM, both on one stream, nothing through the host.  The reference's meshes are LFS pointers, so the scene is synthetic in the
    config's shape: its patch box, patch_scale 0.09, step 0.002, 1024 samples, 'nearest', 48 x 48 patches on a waving sheet that is
    also the instancer mesh (`synthetic.patch_sheet`).  value = in-patch ray-samples/s over the whole step; `roofline` = the tail's
    kernel (MFMA) on the in-patch samples the instancer produced; `instancer` = its own HBM roofline (every output element once);
    `parity` = 48 seeded rays: the instancer's buffers bit for bit and the rendered RGBA within 1e-4 of the restated pipeline."""
    import torch
    from nerf_tex_amd import _lib, synthetic
    from nerf_tex_amd.dataset import look_at
    from nerf_tex_amd.instancer import Instancer
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.proxy import AABB
    from nerf_tex_amd.ray_sampler import Proxy
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("carpet_instanced_scene is a single-GPU workload")
    if args.precision != "float32":
        raise SystemExit("carpet_instanced_scene is timed at float32")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    fam = synthetic.FAMILIES["carpet"]
    emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
    model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
    model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0))
    grid, side, S, step_size, patch_scale, density_scale = 48, 128, 1024, 0.002, 0.09, 400.0      # (weights and density scale of carpet_instanced)
    n, P = side * side, model.n_params
    tr, mesh_v, mesh_f = synthetic.patch_sheet(grid)
    textures = ['', '', '', '', 'light']                      # config_carpet_render.py:86 without its image texture
    b_0, b_1 = synthetic.PATCH_BOX
    inst = Instancer(b_0, b_1, textures=textures, transformations=tr, instance_sampling_method="nearest", mesh=(mesh_v, mesh_f))
    c2w = look_at(np.asarray(fam["cam"], np.float32))
    focal = 800 / np.tan(fam["angle"] / 2) / 2
    r0 = (800 - side) // 2                                    # a centred side x side window of the config's 800 x 800 grid
    rows, cols = np.meshgrid(np.arange(r0, r0 + side), np.arange(r0, r0 + side), indexing="ij")
    loc = torch.as_tensor(np.stack([rows.ravel(), cols.ravel()], -1).astype(np.float32), device=dev)
    aabb = ([-1.7, -1.7, -.3], [1.7, 1.7, .4])
    ro, rd, t, cone = Proxy(800, 800, focal, AABB(*aabb))(loc, c2w, device=dev)
    params = torch.as_tensor(np.asarray([fam["params"]], np.float32), device=dev).repeat(n, 1).contiguous()
    e = lambda *shape, dt=torch.float32: torch.empty(shape, device=dev, dtype=dt)
    rays_d_map, pts, tt, dists = e(n, S, 3), e(n, S, 3), e(n, S), e(n, S)
    color_last, alpha_last, weight = e(n, 3), e(n), e(n, S)
    instance_id, hit, params_map = e(n, S, dt=torch.int32), e(n, dt=torch.uint8), e(n, S, P)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    color, alpha = e(n, 3), e(n)
    cone1 = cone.reshape(n).contiguous()
    stream = torch.cuda.current_stream(dev).cuda_stream
    model.reserve(0, n)
    seed = 1

    def instancer_step():
        _lib.check(_lib.lib.ntx_instancer_model_input(
            inst._h, ro.data_ptr(), rd.data_ptr(), params.data_ptr(), n, S, step_size, seed, None, rays_d_map.data_ptr(), pts.data_ptr(),
            tt.data_ptr(), dists.data_ptr(), color_last.data_ptr(), alpha_last.data_ptr(), weight.data_ptr(), instance_id.data_ptr(),
            hit.data_ptr(), params_map.data_ptr(), status.data_ptr(), stream))

    def tail_step():
        _lib.check(_lib.lib.ntx_render_instanced(
            model.ctx(0), rays_d_map.data_ptr(), pts.data_ptr(), tt.data_ptr(), dists.data_ptr(), color_last.data_ptr(),
            alpha_last.data_ptr(), weight.data_ptr(), instance_id.data_ptr(), hit.data_ptr(), params_map.data_ptr(), cone1.data_ptr(),
            n, S, -1, patch_scale, density_scale, 0, _lib.f3([1, 1, 1.]), None, None, color.data_ptr(), alpha.data_ptr(), None, stream))

    for _ in range(args.warmup):
        instancer_step(); tail_step()
    torch.cuda.synchronize()
    n_in = int((dists > 0).sum().item())
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b, c in ev:
        a.record(); instancer_step(); b.record(); tail_step(); c.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    inst_ms = float(np.mean([a.elapsed_time(b) for a, b, c in ev]))
    kernel_ms = float(np.mean([b.elapsed_time(c) for a, b, c in ev]))
    flops_per_sample = 2 * model.macs_per_sample()
    achieved = n_in * flops_per_sample / (kernel_ms * 1e-3) / 1e12
    out_bytes = n * S * 4 * (3 + 3 + 1 + 1 + 1 + 1 + P) + n * (12 + 4 + 1) + n * (24 + 4 * P)
    gbps = out_bytes / (inst_ms * 1e-3) / 1e9
    line = {
        "metric": "in-patch ray-samples/sec (rays -> patch instancer -> InstanceRenderer tail, one render chunk)",
        "value": n_in * args.steps / elapsed, "unit": "ray-samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"carpet_instanced_scene: {n} rays of config_carpet_render.py's first camera x {S} marching steps of {step_size} through "
                               f"{grid * grid} patches (box {b_0}..{b_1} x {patch_scale}) on a waving sheet of {mesh_f.shape[0]} triangles, 'nearest', "
                               f"directional light; {n_in} in-patch samples ({n_in / n:.1f} per ray); ParamNerf n_parameters={list(fam['n_parameters'])}; "
                               "ntx_instancer_model_input -> ntx_render_instanced on one stream",
                   "rays": n, "marching_samples_per_ray": S, "in_patch_samples": n_in, "hit_rays": int(hit.sum().item()), "flops_per_sample": flops_per_sample},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS,
                     "traffic": None, "kernel": "ntx::instance_kernel", "kernel_ms": kernel_ms,
                     "what": "ntx_render_instanced (ordering kernels + instance_kernel) on the instancer's own output"},
        "instancer": {"ms": inst_ms, "share_of_step": inst_ms / (inst_ms + kernel_ms), "rays_per_s": n / (inst_ms * 1e-3), "status_flag": int(status.item()),
                      "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "algorithmic_bytes": out_bytes,
                                   **instancer_traffic()},
                      "kernels": "inst_hits_kernel + inst_mesh_kernel + inst_march_kernel (DESIGN 4.5)"}}
    if not args.no_parity:
        from oracle import instancer_oracle as io
        from oracle import nerftex_oracle as orc
        t1 = time.perf_counter()
        pick = np.sort(np.random.default_rng(7).choice(n, size=parity_rays, replace=False))
        ti = torch.as_tensor(pick, device=dev)
        spec = io.make_spec(b_0, b_1, None, textures=textures, instance_sampling_method="nearest", mesh=(mesh_v, mesh_f), matrices=inst.matrices())
        h = lambda x: x[ti].cpu().numpy()
        want = io.get_model_input(spec, h(ro), h(rd), h(params), S, step_size, io.offset_uniforms(n, seed)[pick], io.choice_uniforms(n, S, seed)[pick])
        got = [h(rays_d_map), h(pts), h(tt), h(dists), h(color_last)[:, None, :], h(alpha_last)[:, None], h(weight), h(instance_id), h(hit).astype(bool), h(params_map)]
        same = all(np.array_equal(g, w) for g, w in zip(got, want))
        mspec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
        rc, ra = orc.instance_evaluate_model(orc.split_blob(mspec, model.get_blob()), mspec, *want[:8], want[8], want[9], h(cone1)[:, None], None,
                                             patch_scale, density_scale, True, False, False, (1., 1., 1.), None, dtype=np.float64)
        rgba = np.concatenate([h(color), h(alpha)[:, None]], -1)
        err = float(orc.rel_linf(rgba, np.concatenate([rc, ra[:, None]], -1)))
        line["parity"] = {"rays": parity_rays, "instancer_buffers_bit_identical": bool(same), "rel_linf_f64": err, "tolerance": 1e-4, "ok": bool(same and err <= 1e-4),
                          "what": "seeded rays of the timed chunk: the ten buffers of ntx_instancer_model_input against the restatement of "
                                  "instancer.cpp:751-1037 (oracle/instancer_oracle.py; Embree cannot be built here: unpinned), and [color, alpha] against "
                                  "that restatement followed by the float64 restatement of renderer.py:247-354",
                          "oracle_seconds": round(time.perf_counter() - t1, 2)}
    if emit:
        print(json.dumps(line), flush=True)
    return line


def launch_ranks(cmds, envs, deadline_s: float, log_dir: str, poll_s: float = 0.2, label: str = "bench.py") -> int:
    """Start one process per rank (own session each, so a rank's children die with it), rank 0's stdout passed through, every
    rank's stderr in `log_dir/rank<r>.err`.  Poll them: the first non-zero exit ends the others (SIGTERM, SIGKILL after 5 s); so
    does the deadline (exit code 124).  On failure the tail of every rank's log is echoed, labelled.  Returns the exit code."""
    import signal
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    for r, (cmd, env) in enumerate(zip(cmds, envs)):
        f = open(os.path.join(log_dir, f"rank{r}.err"), "wb")
        logs.append(f)
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL, stderr=f, start_new_session=True))

    def end_all():
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            alive = [p for p in procs if p.poll() is None]
            if not alive:
                return
            for p in alive:
                try:
                    os.killpg(p.pid, sig)                    # exactly the sessions started above
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.monotonic() + grace
            while time.monotonic() < t_end and any(p.poll() is None for p in alive):
                time.sleep(0.05)

    t0 = time.monotonic()
    rc, why = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                r, c = failed[0]
                rc, why = (abs(c) if abs(c) < 256 else 1) or 1, f"rank {r} exited with {c}; ending the other ranks"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() - t0 > deadline_s:
                rc, why = 124, f"deadline of {deadline_s:.0f} s exceeded; ending all ranks"
                break
            time.sleep(poll_s)
    finally:
        end_all()
        for f in logs:
            f.close()
    if rc != 0:
        print(f"{label}: {why}", file=sys.stderr)
        for r in range(len(procs)):
            try:
                with open(os.path.join(log_dir, f"rank{r}.err"), "rb") as f:
                    tail = f.read()[-3000:].decode("utf-8", "replace")
            except OSError:
                tail = ""
            for line in tail.splitlines()[-25:]:
                print(f"[rank {r}] {line}", file=sys.stderr)
    return rc


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, and pass rank 0's
    JSON line through (launch_ranks: bounded, siblings ended on the first failure, per-rank logs)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmds, envs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC
        cmds.append([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]); envs.append(env)
    return launch_ranks(cmds, envs, args.deadline, os.path.join(ROOT, "logs"))


def parity_block(renderer, model, family, batch, rgba, S, n_check=256, seed=7):
    """The second half of BASELINE.json's metric.  `n_check` seeded rays of the batch the timed region rendered, recomputed by the
    float64 ("truth") and float32 ("what TF-CPU float32 computes, up to summation order") restatements of renderer.py:47-213
    in oracle/ -- the checker, here and nowhere in the timed path -- and compared with the image the GPU produced:
    rel-Linf = max|out - ref| / max|ref| over [color_pred, alpha_pred] of those rays."""
    import torch
    from oracle import nerftex_oracle as orc
    from nerf_tex_amd import synthetic
    fam = synthetic.FAMILIES[family]
    spec = orc.ModelSpec(kind="ParamNerf", n_parameters=tuple(fam["n_parameters"]))
    w = orc.split_blob(spec, model.get_blob())
    n = batch["rays_o"].shape[1]
    idx = np.sort(np.random.default_rng(seed).choice(n, size=min(n_check, n), replace=False))
    ti = torch.as_tensor(idx, device=rgba.device)
    g = lambda k: batch[k][0][ti].double().cpu().numpy()
    ro, rd, t, cone = g("rays_o"), g("rays_d"), g("t"), g("cone_scale").reshape(len(idx), 1)
    params = batch["parameters"].double().cpu().numpy()
    got = rgba[ti].double().cpu().numpy()
    out = {"rays": int(len(idx)), "what": "rel-Linf over [color_pred, alpha_pred] of seeded rays of the timed image vs the numpy "
           "restatement of renderer.py:47-213 (oracle/nerftex_oracle.py; float64 = truth, float32 = the reference's arithmetic up "
           "to summation order); TensorFlow itself cannot run here: parity is pinned to the restatement, not to a TF run",
           "tolerance": 1e-4}
    t0 = time.perf_counter()
    for name, dt in (("rel_linf_f64", np.float64), ("rel_linf_f32", np.float32)):
        ref = orc.renderer_call(w, spec, ro[None].astype(dt), rd[None].astype(dt), t[None].astype(dt), params.astype(dt), cone[None].astype(dt),
                                S, blur_idx=fam["blur_idx"], dtype=dt)
        want = np.concatenate([ref["color_pred"][0], ref["alpha_pred"][0][:, None]], -1).astype(np.float64)
        out[name] = float(orc.rel_linf(got, want))
    out["oracle_seconds"] = round(time.perf_counter() - t0, 2)
    out["ok"] = bool(out["rel_linf_f32"] <= 1e-4)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="carpet", choices=sorted(WORKLOADS) + sorted(SHARDED) + ["carpet_instanced", "carpet_instanced_scene", "carpet_train_step", "grass_filtered_train_step", "fur_train_step"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="about how long the cpu_baseline sample may take (default 15 s: one full reference render_chunk where the host manages)")
    ap.add_argument("--precision", default="float32", choices=["float32", "fp16x3"],
                    help="arithmetic of the Dense layers (include/nerftex.h: ntx_precision); float32 = the reference's")
    ap.add_argument("--perturb", action="store_true", help="stratified jitter of the depths inside the kernel (the reference's default perturb=True)")
    ap.add_argument("--no-extras", action="store_true", help="skip the second figures -- fp16x3, perturb, ray setup, and `extras` (the training step and the instanced scene) -- (profiling runs: one kernel flavour per process)")
    ap.add_argument("--shard", default="rows", choices=["rows", "bands"], help="sharded workloads: pixel rows round-robin, or contiguous bands")
    ap.add_argument("--deadline", type=float, default=float(os.environ.get("NTX_BENCH_DEADLINE", "600")),
                    help="multi-rank runs: seconds after which the launcher ends all ranks / every rank's watchdog exits")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of 256 rays after the timed region")
    ap.add_argument("--scene-parity-rays", type=int, default=48, help="carpet_instanced_scene: rays of the timed chunk checked against the restated instancer + renderer (0.14 s of host time a ray)")
    ap.add_argument("--raw-noise-std", type=float, default=0.0, help="raw_noise_std of the renderer (renderer.py:190-192), drawn inside the kernel")
    ap.add_argument("--instanced-per-sample-dirs", action="store_true",
                    help="carpet_instanced: every marching sample its own direction and appearance parameters (the round-2 workload) instead of one per run")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.workload == "carpet_instanced":
        return bench_instanced(args)
    if args.workload == "carpet_instanced_scene":
        return bench_instanced_scene(args, parity_rays=args.scene_parity_rays)
    if args.workload.endswith("_train_step"):
        return bench_train_step(args)

    # The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL writes its version banner with printf
    # when a communicator is created), so from here on file descriptor 1 goes to stderr and the line is written to the
    # original stdout at the end.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from nerf_tex_amd import synthetic
    from nerf_tex_amd.dist import Comm, ShardMap, gather_image
    from nerf_tex_amd.model import ParamNerf
    from nerf_tex_amd.renderer import Renderer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Development knob (never set by the driver; the line says so): NTX_BENCH_SHARE_GPU=1 runs the N ranks on ONE GPU -- process
    # group on gloo, the RGBA gathered through host memory by the same plan -- so that everything of an N > 1 run except RCCL
    # itself (launcher, shard maps, per-rank fields, the bit-identity check of the sharded image) executes on a 1-GPU box.
    # Its timings mean nothing: the ranks take turns on the GPU.
    share_gpu = world > 1 and os.environ.get("NTX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdev = torch.device("cpu") if share_gpu else dev        # where the few scalars the ranks exchange live (gloo: host)
    comm, gather_how = None, None
    if share_gpu:
        import faulthandler
        faulthandler.dump_traceback_later(args.deadline, exit=True)
        dist.init_process_group("gloo")
        gather_how = "torch.distributed (gloo) through host memory: NTX_BENCH_SHARE_GPU=1, all ranks on one GPU -- timings are not a measurement"
    elif world > 1:
        # a rank stuck in a collective (a peer died, ncclCommInitRank never completes ...) must not hold the node: after the
        # deadline this rank dumps its stacks to stderr and exits, whatever the main thread is blocked in
        import faulthandler
        faulthandler.dump_traceback_later(args.deadline, exit=True)
        dist.init_process_group("nccl", device_id=dev)       # nccl backend == RCCL on ROCm: barrier + max-over-ranks timing
        # the data path's own communicator, behind the C ABI.  Comm() agrees among the ranks BEFORE anyone enters
        # ncclCommInitRank and raises on every rank alike (dist.CommUnavailable) when one of them cannot go ahead; an error out
        # of ncclCommInitRank itself is agreed on below.  Either way every rank falls back to the same exchange through
        # torch.distributed on the same RCCL, and the line says so.
        err = None
        try:
            comm = Comm(local_rank)
        except Exception as e:                               # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        flag = torch.tensor([0 if err is None else 1], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            if comm is not None:
                comm.close()
            comm = None
            gather_how = "torch.distributed gather/send/recv of the same plan (fallback: no ntx_comm on some rank" + (f"; here: {err})" if err else ")")
            print("bench.py: " + gather_how, file=sys.stderr)
        else:
            gather_how = f"ntx_gather_image (RCCL through the C ABI, {comm.library})"

    sharded = args.workload in SHARDED
    family, H, W, S, cfg_idx = (SHARDED if sharded else WORKLOADS)[args.workload]
    fam = synthetic.FAMILIES[family]
    emb = lambda n: {"module": "network.model.FourierFeatures", "n_freq_bands": n}
    model = ParamNerf(emb(10), emb(4), emb(4), list(fam["n_parameters"]))["model"]
    model.set_blob(synthetic.synthetic_weights(model.layer_table(), seed=0))
    mk = lambda prec, perturb: Renderer(model=model, n_samples=S, perturb=perturb, blur_idx=fam["blur_idx"], check_numerics=False,
                                        precision=prec, raw_noise_std=args.raw_noise_std)
    renderer = mk(args.precision, args.perturb)
    params = torch.as_tensor(np.asarray([fam["params"]], np.float32), device=dev)

    def camera_rays(shard_map, r):
        """the true camera of the config: this rank's pixel set generated on the device (ntx_generate_rays_strided)"""
        from nerf_tex_amd.dataset import look_at
        from nerf_tex_amd.pixel_sampler import Full
        from nerf_tex_amd.proxy import AABB
        from nerf_tex_amd.ray_sampler import Proxy
        focal = W / np.tan(fam["angle"] / 2) / 2                                  # dataset.py:229
        sampler = Proxy(H, W, focal, AABB(fam["b_0"], fam["b_1"]))
        ro, rd, t, cone = sampler(Full(H, W, shard=(shard_map, r))(), look_at(fam["cam"]), device=dev)
        # ray_index: jitter / noise are keyed by the PIXEL, so the sharded image equals the 1-GPU image also under --perturb
        return dict(rays_o=ro[None], rays_d=rd[None], t=t[None], cone_scale=cone[None], parameters=params, ray_index=shard_map.ray_index(r))

    if sharded:
        shard = ShardMap(H * W, world, W if args.shard == "rows" else None)
        batch = camera_rays(shard, rank)
        n_rays = shard.count(rank)
        n_hit = int(torch.isfinite(batch["t"][0, :, 0]).sum().item())
    else:
        n_rays = H * W                                           # per GPU (weak scaling)
        shard = ShardMap(n_rays * world, world)                  # every rank's own band of an (H * world) x W image
        ro, rd, t, cone = synthetic.all_hit_rays(n_rays, fam["b_0"], fam["b_1"], fam["cam"], seed=1 + rank)
        d = lambda a: torch.as_tensor(a, device=dev)[None]
        batch = dict(rays_o=d(ro), rays_d=d(rd), t=d(t), cone_scale=d(cone), parameters=params, ray_index=shard.ray_index(rank))
        n_hit = n_rays

    mkev = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev0, ev1, ev2 = mkev(), mkev(), mkev()

    def step(i=None, r=renderer, b=batch, gather=True):
        if i is not None:
            ev0[i].record()
        out = r(**b, seed=1234)                              # one seed for the image; the ray index map tells the shards apart
        if i is not None:
            ev1[i].record()                                  # same stream the kernel was launched on
        rgba = torch.cat([out["color_pred"][0], out["alpha_pred"][0][:, None]], -1)
        if world > 1 and gather and share_gpu:
            g_ = gather_image(rgba.cpu(), shard)             # (development knob: the same plan through host memory)
            rgba = g_.to(dev) if g_ is not None else None
        elif world > 1 and gather:
            rgba = gather_image(rgba, shard, comm=comm)      # the one collective: RGBA -> rank 0 (ntx_gather_image)
        if i is not None:
            ev2[i].record()
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
    hits_total = n_hit
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev1, ev2)]))   # torch.cat + the gather (+ un-shard on the root)
    per_rank = None
    if world > 1:
        tt = torch.tensor([elapsed, float(n_hit)], device=sdev, dtype=torch.float64)
        mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, hits_total = float(mx[0].item()), int(sm[1].item())
        mine = torch.tensor([kernel_ms, gather_ms, float(n_rays), float(n_hit), comm.init_seconds if comm is not None else -1.0], device=sdev, dtype=torch.float64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "kernel_ms": float(v[0]), "gather_ms": float(v[1]), "rays": int(v[2]), "hits": int(v[3]),
                     "comm_init_s": (float(v[4]) if float(v[4]) >= 0 else None)} for r, v in enumerate(every)]      # ncclCommInitRank of ntx_comm_create, per rank

    # rank 0 re-times its own shard ALONE (no gather, the other ranks wait at the barrier below): what one GPU takes for the same
    # rays when its 7 neighbours are idle -- the reference point of the efficiency figure in the line
    alone_ms = None
    if world > 1 and rank == 0:
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        step(gather=False); torch.cuda.synchronize()
        a0.record()
        for _ in range(args.steps):
            step(gather=False)
        a1.record(); torch.cuda.synchronize()
        alone_ms = a0.elapsed_time(a1) / args.steps

    # sharded image vs the same image rendered by ONE GPU (rank 0 alone, outside the timed region): bit-identical?
    identical, whole_ms = None, None
    if sharded and world > 1 and rank == 0:
        whole = camera_rays(ShardMap(H * W, 1), 0)
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record()
        o1 = renderer(**whole, seed=1234)
        w1.record(); torch.cuda.synchronize()
        whole_ms = w0.elapsed_time(w1)                       # the whole image on ONE GPU: the strong-scaling reference
        ref = torch.cat([o1["color_pred"][0], o1["alpha_pred"][0][:, None]], -1)
        identical = bool(torch.equal(ref, img))              # also under --perturb: the generators are keyed by the pixel
    elif world > 1 and rank == 0:
        # weak scaling: the gathered image is the N ranks' bands one after the other; rank 0 renders every band ALONE (the peers' rays are
        # seeded by their rank) and compares -- what arrived over RCCL against what one GPU computes for the same rays, bit for bit
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record()
        bands = []
        for r in range(world):
            ro_, rd_, t_, cone_ = synthetic.all_hit_rays(n_rays, fam["b_0"], fam["b_1"], fam["cam"], seed=1 + r)
            d_ = lambda a: torch.as_tensor(a, device=dev)[None]
            o1 = renderer(rays_o=d_(ro_), rays_d=d_(rd_), t=d_(t_), cone_scale=d_(cone_), parameters=params, ray_index=shard.ray_index(r), seed=1234)
            bands.append(torch.cat([o1["color_pred"][0], o1["alpha_pred"][0][:, None]], -1))
        w1.record(); torch.cuda.synchronize()
        whole_ms = w0.elapsed_time(w1)                       # (includes making the peers' rays on the host: not a timing reference)
        identical = bool(img is not None and torch.equal(torch.cat(bands, 0), img))

    # second figures on the same inputs, outside the timed region (rank 0, N = 1), clearly labelled, never `value`:
    # the opt-in fp16x3 precision, and the reference's default perturb=True (stratified jitter inside the kernel)
    def timed(r2):
        o2 = r2(**batch, seed=99)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(args.steps):
            o2 = r2(**batch, seed=99)
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / args.steps, torch.cat([o2["color_pred"][0], o2["alpha_pred"][0][:, None]], -1)

    alt = jit = None
    if world == 1 and args.precision == "float32" and not args.perturb and not args.no_extras:
        ms2, rgba2 = timed(mk("fp16x3", False))
        alt = {"precision": "fp16x3 (3-term split of weights and activations into IEEE halves on v_mfma_f32_32x32x16_f16, f32 accumulate)",
               "value": n_hit * S / (ms2 * 1e-3), "unit": "ray-samples/s", "kernel_ms": ms2,
               "rel_linf_vs_float32_kernel": float((rgba2 - img).abs().max() / img.abs().max()),
               # canonical FLOPs (2 * MACs per ray-sample) over the 16-bit dense peak; the kernel issues 3 half-precision MFMA products
               # per canonical MAC, so the matrix pipe is 3x busier than this fraction
               "canonical_frac_of_f16_peak": n_hit * S * 2 * model.macs_per_sample() / (ms2 * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
               "mfma_products_per_mac": 3}
        ms3, _ = timed(mk("float32", True))
        jit = {"what": "perturb=True (renderer.py:106-111, the reference's default): stratified jitter drawn inside the kernel "
                       "(Philox4x32-10 per depth), no [N,S] depth tensor", "value": n_hit * S / (ms3 * 1e-3),
               "unit": "ray-samples/s", "kernel_ms": ms3}

    # SURVEY 8d "both with and without ray setup": the same step with ray generation inside the timed loop (rank 0; for the
    # all-hit workloads the generated camera grid is the config's own camera and is NOT what is rendered -- the rendered rays are
    # the synthetic all-hit set -- so this prices ntx_generate_rays next to the render, nothing else)
    setup = None
    if rank == 0 and not args.no_extras:
        from nerf_tex_amd.dataset import look_at
        from nerf_tex_amd.pixel_sampler import Full
        from nerf_tex_amd.proxy import AABB
        from nerf_tex_amd.ray_sampler import Proxy
        focal = W / np.tan(fam["angle"] / 2) / 2
        sampler = Proxy(H, W, focal, AABB(fam["b_0"], fam["b_1"]))
        c2w = look_at(fam["cam"])
        pix = Full(H, W, shard=(shard, rank))() if sharded else Full(H, W)()

        def step_with_setup():
            ro, rd, tt_, cone_ = sampler(pix, c2w, device=dev)
            b2 = dict(batch, rays_o=ro[None], rays_d=rd[None], t=tt_[None], cone_scale=cone_[None]) if sharded else batch
            return renderer(**b2, seed=1234)

        step_with_setup(); torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(args.steps):
            step_with_setup()
        s1.record(); torch.cuda.synchronize()
        ms_setup = s0.elapsed_time(s1) / args.steps
        setup = {"value": n_hit * S / (ms_setup * 1e-3), "unit": "ray-samples/s", "ms": ms_setup, "ms_without": kernel_ms,
                 "what": (f"this rank's step with its rays regenerated on the device every step: ntx_generate_rays_strided ({pix[1]} pixels of the "
                          f"{H}x{W} camera grid: pixel_sampler.Full + rays_from_camera + Proxy/AABB, ray_sampler.py:32-48) + the render of those rays"
                          if sharded else
                          f"ntx_generate_rays of the {H}x{W} grid of the config's camera (pixel_sampler.Full + rays_from_camera + Proxy/AABB, "
                          f"ray_sampler.py:32-48) + the render of the workload's all-hit rays, per step")}

    parity = None
    if rank == 0 and not args.no_parity:
        local = step(gather=False)                           # this rank's own rays as the timed region rendered them
        parity = parity_block(renderer, model, family, batch, local, S) if not (args.perturb or args.raw_noise_std > 0) else None

    if rank == 0:
        flops_per_sample = 2 * model.macs_per_sample()
        # roofline of the dominant kernel on THIS rank: canonical FLOPs (2 * MACs per ray-sample; split-precision
        # multiplicity does not count) against the dense peak of the issued MFMA dtype
        achieved = n_hit * S * flops_per_sample / (kernel_ms * 1e-3) / 1e12
        peak = F32_MFMA_PEAK_TFLOPS if args.precision == "float32" else F16_MFMA_PEAK_TFLOPS
        traffic, traffic_src, traffic_prof = measured_traffic(args.workload, args.precision)
        if sharded:
            what = (f"{args.workload}: ONE {H}x{W}x{S} image of the {family} config's camera (BASELINE configs[{cfg_idx}]), rays generated on "
                    f"the device, {'pixel rows dealt round-robin' if args.shard == 'rows' else 'contiguous bands'} over {world} GPU(s), "
                    f"{hits_total} of {H * W} rays hit the proxy")
        else:
            what = (f"{args.workload} {H}x{W}x{S}: {n_rays} all-hit rays x {S} samples per GPU (BASELINE configs[{cfg_idx}])")
        line = {
            "metric": "ray-samples/sec (MLP+composite) at 800x800x64",
            "value": hits_total * S * args.steps / elapsed,
            "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "float32" else "fp16x3 (f32 accumulate)", "data": "synthetic",
            "config": {"workload": what + f", ParamNerf n_parameters={list(fam['n_parameters'])}, seeded glorot weights, "
                                   f"inputs resident in HBM, fused PE+MLP+composite"
                                   + (", perturb=True (in-kernel jitter)" if args.perturb else "")
                                   + (f", raw_noise_std={args.raw_noise_std} (in-kernel N(0,1) per sample)" if args.raw_noise_std > 0 else "")
                                   + (f", + gather of RGBA to rank 0: {gather_how}" if world > 1 else ""),
                       "rays_per_gpu": n_rays, "hit_rays_total": hits_total, "samples_per_ray": S, "flops_per_sample": flops_per_sample},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes per ntx_render_rays call, all kernels (HBM side, rocprofv3 PMC)", "traffic_source": traffic_src,
                         "traffic_profile_head": (traffic_prof or {}).get("git_head"), "traffic_profile_current": (traffic_prof or {}).get("current"),
                         "algorithmic_bytes": n_rays * 4 * (3 + 3 + 2 + 1 + 4) + 4 * model.n_params,
                         "kernel": "ntx::render_kernel" if args.precision == "float32" else "ntx::render_kernel_x3",
                         "kernel_ms": kernel_ms},
        }
        if per_rank is not None:
            kms = [p_["kernel_ms"] for p_ in per_rank]
            line["per_rank"] = per_rank
            line["gather_bytes"] = int(sum(p_["rays"] for p_ in per_rank[1:]) * 16)      # RGBA float32 of every peer -> rank 0
            line["gather_how"] = gather_how
            line["rccl_version"] = comm.version if comm is not None else None        # ncclGetVersion of the librccl behind ntx_comm (22606 = 2.26.6)
            line["imbalance"] = max(kms) / (sum(kms) / len(kms))                          # max / mean kernel_ms over the ranks
            line["rank0_alone_ms"] = alone_ms
            if sharded:
                # strong scaling: one GPU's time for the whole image / (N x the job's time per step)
                line["whole_image_1gpu_ms"] = whole_ms
                line["efficiency_vs_1gpu"] = whole_ms / (world * elapsed / args.steps * 1e3) if whole_ms else None
            else:
                # weak scaling: every rank has rank 0's work; ideal = the job takes what rank 0 takes alone
                line["efficiency_vs_rank0_alone"] = alone_ms / (elapsed / args.steps * 1e3)
        if world > 1:
            assert identical is not None, "an N > 1 run must compare its gathered image with one GPU's"
        if identical is not None:
            line["sharded_image_bit_identical_to_1gpu"] = identical
        if parity is not None:
            line["parity"] = parity
        if setup is not None:
            line["with_ray_setup"] = setup
        if alt is not None:
            line["fp16x3"] = alt
        if jit is not None:
            line["perturb"] = jit
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(family, S, args.cpu_baseline_seconds)
        if world == 1 and not args.no_extras and args.precision == "float32" and args.workload == "carpet":
            # the two other headline figures on the driver's record, after the timed region and outside `value`: the training step
            # (network/train.py:61-67) and the path the shipped render configs run (renderer.py:247-354 behind the patch instancer) -- 10 / 5 steps
            # of `--workload carpet_train_step` / `carpet_instanced_scene` with a small oracle check; `--no-extras` skips them
            t1 = time.perf_counter()
            ns = argparse.Namespace(**vars(args)); ns.no_cpu_baseline, ns.no_parity = True, False
            ns.workload, ns.steps, ns.warmup = "carpet_train_step", 10, 5      # (8 ms a step: the first launches behind another workload run at a lower clock)
            lt = bench_train_step(ns, emit=False, data_side=False, parity_rays=64)
            ns.workload, ns.steps, ns.warmup = "carpet_instanced_scene", 5, 2
            li = bench_instanced_scene(ns, emit=False, parity_rays=16)
            line["extras"] = {
                "train_step": {"ms": lt["roofline"]["kernel_ms"], "value": lt["value"], "unit": lt["unit"], "frac": lt["roofline"]["frac"], "frac_what": lt["roofline"]["what"],
                               "frac_3x_forward": lt["roofline"]["frac_3x_forward"], "parity": {k: v for k, v in lt["parity"].items() if k != "what"},
                               "workload": lt["config"]["workload"]},
                "instanced_scene": {"ms": li["ms_per_step"], "value": li["value"], "unit": li["unit"], "kernel_ms": li["roofline"]["kernel_ms"], "frac": li["roofline"]["frac"],
                                    "frac_what": li["roofline"]["what"], "instancer_ms": li["instancer"]["ms"], "in_patch_samples": li["config"]["in_patch_samples"],
                                    "parity": {k: v for k, v in li["parity"].items() if k != "what"}, "workload": li["config"]["workload"]},
                "seconds": None, "what": "`--workload carpet_train_step` (10 steps) and `--workload carpet_instanced_scene` (5 steps) after the timed region; never part of `value`"}
            line["extras"]["seconds"] = round(time.perf_counter() - t1, 1)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()                                       # rank 0 may still be rendering its reference image
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
