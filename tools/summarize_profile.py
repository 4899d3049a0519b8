#!/usr/bin/env python3
"""gpurun_out/<prefix>_{kt,pmc1..5}/ (rocprofv3 CSV output) -> profiles/<round>/<name>_*.{csv,json}.

    python tools/summarize_profile.py prof2 r02 bench_carpet_v7 [kernel-substring, default "render_kernel<"]

Copies the kernel-stats CSV of the --kernel-trace --stats pass and reduces the separate --pmc passes to per-launch means
for the dominant kernel, with the derived quantities DESIGN.md quotes.  HBM-side traffic is also summed over ALL kernels of
the library (`ntx::` in the name: the hit compaction next to the render kernel) as bytes per ntx_render_rays call."""
import collections, csv, glob, json, os, shutil, sys

prefix, rnd, name = sys.argv[1:4]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "render_kernel<"   # "render_kernel_x3<" for the fp16x3 precision
O, P = "gpurun_out", os.path.join("profiles", rnd)
os.makedirs(P, exist_ok=True)
shutil.copy(glob.glob(f"{O}/{prefix}_kt/*kernel_stats.csv")[0], f"{P}/{name}_kernel_stats.csv")
summ, info, per_kernel = {}, {}, collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{O}/{prefix}_pmc*/*counter_collection.csv")):
    allrows = list(csv.DictReader(open(f)))
    rows = [r for r in allrows if KERNEL in r["Kernel_Name"]]
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summ[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    if rows:
        info = {k: rows[0][k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                       "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for r in allrows:
        if "ntx::" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            per_kernel[r["Kernel_Name"].split("<")[0].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
g = lambda k: summ[k]["mean_per_launch"]
stats = [r for r in csv.DictReader(open(f"{P}/{name}_kernel_stats.csv")) if KERNEL in r["Name"]][0]
avg_s = float(stats["AverageNs"]) * 1e-9
n_simd = 256 * 4
# every ntx:: kernel runs once per call: bytes per call = sum over kernels of their per-launch means
call_rd = sum(2 * 1024 * sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) for v in per_kernel.values() if v.get("FETCH_SIZE"))
call_wr = sum(1024 * sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) for v in per_kernel.values() if v.get("WRITE_SIZE"))
derived = {
    "kernel_avg_ms_kernel_trace": avg_s * 1e3,
    "shader_clock_GHz": g("GRBM_GUI_ACTIVE") / 8 / avg_s / 1e9 if "GRBM_GUI_ACTIVE" in summ else None,
    # SQ_VALU_MFMA_BUSY_CYCLES counts 64 cycles per v_mfma_f32_32x32x2_f32 (32 per v_mfma_f32_32x32x16_f16), summed over SIMDs
    "mfma_util": g("SQ_VALU_MFMA_BUSY_CYCLES") / (n_simd * g("GRBM_GUI_ACTIVE") / 8) if "GRBM_GUI_ACTIVE" in summ else None,
    "frac_wave_cycles_in_s_waitcnt": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if "SQ_WAIT_ANY" in summ else None,
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half the bytes of wide coalesced reads (x2, MI355X_MICROARCH.md)
    "hbm_side_read_bytes_corrected": 2 * 1024 * g("FETCH_SIZE") if "FETCH_SIZE" in summ else None,
    "hbm_side_write_bytes_uncalibrated": 1024 * g("WRITE_SIZE") if "WRITE_SIZE" in summ else None,
    "call_hbm_side_read_bytes_corrected": call_rd if per_kernel else None,
    "call_hbm_side_write_bytes_uncalibrated": call_wr if per_kernel else None,
    "call_kernels": sorted(per_kernel),
    "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if "TCC_HIT_sum" in summ else None,
}
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import kernel_sources_sha16, git_head
# the tree the numbers belong to: bench.py quotes `traffic` from here and says whether the kernels have changed since
derived["kernel_sources_sha16"] = os.environ.get("NTX_PROFILE_SOURCES") or kernel_sources_sha16()
derived["git_head"] = os.environ.get("NTX_PROFILE_HEAD") or git_head()
json.dump({"dispatch": info, "counters": summ, "derived": derived}, open(f"{P}/{name}_pmc_summary.json", "w"), indent=1)
print(json.dumps(derived, indent=1))
