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
    d = _run("--no-cpu-baseline")
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


def test_cpu_baseline_block():
    d = _run("--workload", "fur")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "ray-samples/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["cpu_count"] >= c["cores"] and "BLAS_INFO" in c["blas"]       # threads and BLAS backend stated
