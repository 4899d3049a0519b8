#!/usr/bin/env python3
"""Reduce the rocprofv3 passes of tools/dev/r4_*.sh (kt_<tag>, pmc_*_<tag> under a directory) to one JSON per tag:
per ntx_inst:: kernel the trace's times and the per-launch means of every counter."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]
for tag in sys.argv[2:]:
    out = {"command": f"python tools/bench_instancer.py --no-render --steps 10 <the variant's flags: tools/dev/r4_profiles.sh, tag {tag}>", "kernels": {}}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from source_hash import kernel_sources_sha16, git_head
        out["kernel_sources_sha16"] = os.environ.get("NTX_PROFILE_SOURCES") or kernel_sources_sha16(); out["git_head"] = os.environ.get("NTX_PROFILE_HEAD") or git_head()
    except Exception:
        pass
    for f in glob.glob(f"{O}/kt_{tag}/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "ntx_inst::" in r["Name"]:
                out["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
        os.system(f"cp {f} {O}/instancer_{tag}_kernel_stats.csv")
    for f in glob.glob(f"{O}/pmc_*_{tag}/*counter_collection.csv"):
        agg = collections.defaultdict(list); info = {}
        for r in csv.DictReader(open(f)):
            if "ntx_inst::" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0]
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                info[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count") if x in r}
        for (k, c), v in agg.items():
            out["kernels"].setdefault(k, {}).setdefault("counters_mean_per_launch", {})[c] = sum(v) / len(v)
            out["kernels"][k]["dispatch"] = info[k]
    for k, v in out["kernels"].items():
        c = v.get("counters_mean_per_launch", {})
        if "WRITE_SIZE" in c: v["hbm_side_write_bytes"] = c["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in c: v["hbm_side_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2   # gfx950 reports half the bytes of wide reads (MI355X_MICROARCH.md)
    for k, v in out["kernels"].items():                                   # derived: how busy the vector ALUs were, how many lanes worked
        c = v.get("counters_mean_per_launch", {})
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            v["valu_issue_busy"] = c["SQ_INSTS_VALU"] / 1024 * 4 / (c["GRBM_GUI_ACTIVE"] / 8)       # wave64 VALU instruction = 4 cycles of one of 1024 SIMDs
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            v["valu_lanes_active"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
    json.dump(out, open(f"{O}/instancer_{tag}_pmc_summary.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(tag, k, v.get("avg_us"), json.dumps(v.get("counters_mean_per_launch", {})), v.get("dispatch"))
