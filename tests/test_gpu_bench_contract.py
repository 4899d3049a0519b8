"""bench.py's one-line JSON contract on the GPU box (short run).  `-m gpu`."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    d = _run("--cpu-baseline-seconds", "2")
    c = d["cpu_baseline"]                               # the reference's CPU path beside it: a bounded sample, threads and BLAS backend stated
    assert c["kind"] == "port" and c["unit"] == "ray-samples/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["cpu_count"] >= c["cores"] and "BLAS_INFO" in c["blas"] and d["value"] > 50 * c["value"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "800x800x64" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.5 < r["frac"] < 1.0 and d["value"] > 5e7
    # value and the roofline are computed from the same launches
    assert abs(d["value"] * d["config"]["flops_per_sample"] / 1e12 - r["achieved"]) / r["achieved"] < 0.02
    x3 = d["fp16x3"]                                    # the opt-in precision, reported beside the headline, never as `value`
    assert x3["value"] > d["value"] and x3["rel_linf_vs_float32_kernel"] < 1e-4
    jit = d["perturb"]                                  # the reference's default perturb=True, jitter drawn inside the kernel
    assert jit["value"] > 0.97 * d["value"]
    p = d["parity"]                                     # the second half of the metric: rel-Linf of the timed image vs the restatement
    assert p["rays"] == 256 and p["ok"] is True and p["rel_linf_f32"] <= 1e-4 and p["tolerance"] == 1e-4 and p["rel_linf_f64"] < 5e-4
    s_ = d["with_ray_setup"]                            # with and without ray setup
    # (one timed step each: launch-to-launch spread is a few tenths of a percent, ray generation costs ~0.1 ms of 364)
    assert s_["unit"] == "ray-samples/s" and 0.97 * d["value"] < s_["value"] <= 1.02 * d["value"] and s_["ms"] >= s_["ms_without"] * 0.99
    # the two other headline figures ride on the driver's line, outside `value`: the training step and the instanced scene
    x = d["extras"]
    t_, i_ = x["train_step"], x["instanced_scene"]
    assert 5.0 < t_["ms"] < 12.0 and 0.5 < t_["frac"] < 1.0 and t_["frac_3x_forward"] > t_["frac"] and t_["parity"]["ok"] is True and t_["parity"]["rays"] == 64
    assert 20.0 < i_["ms"] < 80.0 and 0.7 < i_["frac"] < 1.0 and i_["parity"]["instancer_buffers_bit_identical"] is True and i_["parity"]["ok"] is True
    assert x["seconds"] < 30


def test_sharded_workload_line_at_one_gpu():
    """BASELINE configs[3] through the bench at N = 1: rays generated on the device per step in `with_ray_setup`, parity block on the
    true camera's rays (hits and misses), the multi-rank fields absent."""
    d = _run("--workload", "fur_sharded", "--no-cpu-baseline")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and "per_rank" not in d and "gather_bytes" not in d
    assert d["parity"]["ok"] is True and d["parity"]["rel_linf_f32"] <= 1e-4
    assert "ntx_generate_rays_strided" in d["with_ray_setup"]["what"] and d["with_ray_setup"]["value"] > 0.95 * d["value"]


def test_the_largest_single_image_at_one_gpu():
    """BASELINE configs[4] as ONE 1600 x 1600 x 128 image on one GPU (2.56 M rays, 327.68 M samples, ~3 s a step): the workload that needs
    `ntx_reserve` beyond the default ray capacity on a fresh context.  One timed step, no warm-up."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras",
                          "--workload", "grass_filtered_sharded"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["steps"] == 1 and d["warmup"] == 0
    assert d["config"]["hit_rays_total"] == 2_560_000 and d["config"]["samples_per_ray"] == 128 and "1600x1600x128" in d["config"]["workload"]
    assert d["parity"]["ok"] is True and d["parity"]["rel_linf_f32"] <= 1e-4
    assert d["roofline"]["frac"] >= 0.95, d["roofline"]                        # cold: the one step includes the first launch
    assert "extras" not in d and "fp16x3" not in d and "with_ray_setup" not in d      # --no-extras: one kernel flavour in the process
    assert "traffic_profile_head" in d["roofline"] and "traffic_profile_current" in d["roofline"]


def test_instanced_scene_line():
    """`--workload carpet_instanced_scene`: rays -> patch instancer -> InstanceRenderer tail, one chunk: the tail's MFMA roofline on the
    samples the instancer produced, the instancer's own HBM roofline, and the parity block (buffers bit for bit, RGBA <= 1e-4)."""
    d = _run("--workload", "carpet_instanced_scene", "--scene-parity-rays", "12")
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["unit"] == "ray-samples/s" and "model" not in d["config"]
    c, r, i = d["config"], d["roofline"], d["instancer"]
    assert c["rays"] == 16384 and c["marching_samples_per_ray"] == 1024 and c["hit_rays"] == 16384 and c["in_patch_samples"] > 3_000_000
    assert r["bound"] == "mfma" and 0.8 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    ir = i["roofline"]
    assert ir["bound"] == "hbm" and ir["peak"] == 8000.0 and abs(ir["frac"] - ir["achieved"] / ir["peak"]) < 1e-9 and 0.1 < ir["frac"] < 1.0
    assert abs(ir["achieved"] - ir["algorithmic_bytes"] / (i["ms"] * 1e-3) / 1e9) / ir["achieved"] < 1e-6 and ir["traffic"] < 1.1 * ir["algorithmic_bytes"]
    assert i["status_flag"] == 0 and i["share_of_step"] < 0.05 and i["ms"] < 2.0
    p_ = d["parity"]
    assert p_["ok"] is True and p_["instancer_buffers_bit_identical"] is True and p_["rel_linf_f64"] <= 1e-4 and p_["rays"] == 12


MULTI_RANK_FIELDS = ("per_rank", "gather_bytes", "gather_how", "imbalance", "rank0_alone_ms")


def test_multi_rank_line_schema():
    """The fields an N > 1 line carries (per-rank kernel / gather times, gather bytes, imbalance, efficiency against rank 0
    alone), checked on the 2-GPU run when the box has two GPUs; on a 1-GPU box the launcher must give up cleanly instead of hanging."""
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--deadline", "240",
                              "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode != 0 and "[rank " in out.stderr      # rank 1 has no GPU: its failure ends rank 0 too, labelled tails echoed
        return
    d = _run("--gpus", "2", "--no-cpu-baseline")
    for k in MULTI_RANK_FIELDS + ("efficiency_vs_rank0_alone",):
        assert k in d, k
    assert len(d["per_rank"]) == 2 and all(set(p) >= {"rank", "kernel_ms", "gather_ms", "rays", "hits"} for p in d["per_rank"])
    assert d["gather_bytes"] == 640000 * 16 and 1.0 <= d["imbalance"] < 1.1 and 0.8 < d["efficiency_vs_rank0_alone"] <= 1.05


@pytest.mark.parametrize("workload,shard", [("fur_sharded", "rows"), ("fur_sharded", "bands"), ("carpet", "rows")])
def test_two_ranks_sharing_the_gpu_run_the_whole_multi_rank_bench(workload, shard):
    """Everything of an N = 2 run except RCCL itself, on a 1-GPU box (development knob NTX_BENCH_SHARE_GPU: both ranks on GPU 0,
    process group on gloo, the gather plan executed through host memory): the self-launcher, the shard maps, every N > 1 field of
    the line, and -- the point -- the sharded image gathered from two ranks is bit-identical to the image one GPU renders."""
    env = dict(os.environ, NTX_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--deadline", "500",
                          "--no-cpu-baseline", "--workload", workload, "--shard", shard], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in MULTI_RANK_FIELDS:
        assert k in d, k
    assert d["n_gpus"] == 2 and "NTX_BENCH_SHARE_GPU" in d["gather_how"] and "NTX_BENCH_SHARE_GPU" in d["config"]["workload"]
    pr = d["per_rank"]
    assert len(pr) == 2 and [p["rank"] for p in pr] == [0, 1] and all(p["kernel_ms"] > 0 for p in pr)
    assert sum(p["hits"] for p in pr) == d["config"]["hit_rays_total"] and d["gather_bytes"] == pr[1]["rays"] * 16
    assert d["parity"]["ok"] is True                        # rank 0's own shard against the restatement
    if workload == "fur_sharded":
        assert d["scaling"] == "strong" and sum(p["rays"] for p in pr) == 640000
        assert d["sharded_image_bit_identical_to_1gpu"] is True
        assert d["whole_image_1gpu_ms"] > 0 and d["efficiency_vs_1gpu"] > 0
        if shard == "rows":                                 # rows dealt round-robin balance what the proxy culls
            assert abs(pr[0]["hits"] - pr[1]["hits"]) < 0.02 * d["config"]["hit_rays_total"]
    else:
        assert d["scaling"] == "weak" and all(p["rays"] == 640000 for p in pr) and d["efficiency_vs_rank0_alone"] > 0
        assert d["sharded_image_bit_identical_to_1gpu"] is True      # the two bands as gathered against rank 0 rendering both alone
    assert "rccl_version" in d and all("comm_init_s" in p for p in pr)


def test_train_step_line_at_one_and_at_two_ranks():
    """`--workload carpet_train_step`: the N = 1 line (roofline on 3 x forward FLOPs, the step's measured traffic quoted, a torch-CPU
    baseline beside it) and the data-parallel N = 2 line with everything but RCCL itself (NTX_BENCH_SHARE_GPU: both ranks on GPU 0, the
    gradient mean through gloo): launcher, per-rank batches, max-over-ranks clock, and the ranks holding identical weights after the steps."""
    d = _run("--workload", "carpet_train_step")
    assert d["n_gpus"] == 1 and d["unit"] == "ray-samples/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and "carpet_train_step" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.4 < r["frac"] < 1.0
    # the fraction is on the FLOPs a step needs: 2 x forward + forward less the encoded inputs' rows; the 3 x forward figure beside it
    assert d["config"]["flops_per_sample_step"] == 2 * (2 * 680832 + 623232) and d["config"]["flops_per_sample_forward"] == 2 * 680832
    assert abs(d["config"]["flops_per_sample_step"] * 262144 / (r["kernel_ms"] * 1e-3) / 1e12 - r["achieved"]) / r["achieved"] < 1e-6
    assert abs(r["frac_3x_forward"] / r["frac"] - 3 * 680832 / (2 * 680832 + 623232)) < 1e-6
    assert d["parity"]["ok"] is True and d["parity"]["rays"] == 64
    assert r["traffic"] is None or r["traffic"] > 1e10
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["value"] > 50 * d["cpu_baseline"]["value"]
    env = dict(os.environ, NTX_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--deadline", "500", "--workload", "carpet_train_step"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and "NTX_BENCH_SHARE_GPU" in d2["allreduce_how"] and "NTX_BENCH_SHARE_GPU" in d2["config"]["workload"]
    assert d2["allreduce_bytes"] == 683524 * 4 and d2["ranks_hold_identical_weights"] is True and d2["value"] > 0 and "cpu_baseline" not in d2
