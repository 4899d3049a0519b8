#!/bin/bash
# GPU box: the instancer's profile campaign (round 3, ABI v4): bench lines on the carpet-shaped scene and its variants, the
# per-kernel times (rocprofv3 --kernel-trace --stats) and the HBM-side counters in their own passes -> gpurun_out/r3inst
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3inst; mkdir -p $O; cd $R
B="timeout 300 python tools/bench_instancer.py"
$B 2>/dev/null | grep "^{" > $O/instancer_bench.jsonl
{ $B --grid 64 --no-render; $B --method random --no-render; $B --method nearest_blend --no-render; $B --no-mesh --no-render;
  $B --shadows 128 --no-render; $B --shadows 100000 --no-render; $B --rays 65536 --no-render; $B --rays 4096 --no-render; $B --samples 256 --step-size 0.008 --no-render; } 2>/dev/null | grep "^{" > $O/instancer_variants.jsonl
P="python tools/bench_instancer.py --no-render --steps 10"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_r -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- $P > /dev/null 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
python - <<'PY'
import collections, csv, glob, json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r3inst")
out = {"command": "python tools/bench_instancer.py --no-render --steps 10", "kernels": {}}
for r in csv.DictReader(open(glob.glob(f"{O}/kt/*kernel_stats.csv")[0])):
    if "ntx_inst::" in r["Name"]:
        out["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
for f in glob.glob(f"{O}/pmc_*/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    info = {}
    for r in csv.DictReader(open(f)):
        if "ntx_inst::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            info[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count")}
    for (k, c), v in agg.items():
        out["kernels"].setdefault(k, {}).setdefault("counters_mean_per_launch", {})[c] = sum(v) / len(v)
        out["kernels"][k]["dispatch"] = info[k]
for k, v in out["kernels"].items():
    c = v.get("counters_mean_per_launch", {})
    if "WRITE_SIZE" in c: v["hbm_side_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c: v["hbm_side_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2    # gfx950 reports half the bytes of wide reads (MI355X_MICROARCH.md)
json.dump(out, open(f"{O}/instancer_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: {x: v.get(x) for x in ("avg_us", "hbm_side_write_bytes", "hbm_side_read_bytes_corrected")} for k, v in out["kernels"].items()}, indent=1))
PY
cp $O/kt/*kernel_stats.csv $O/instancer_kernel_stats.csv
cut -c1-400 $O/instancer_bench.jsonl; python -c "
import json
for l in open('$O/instancer_variants.jsonl'): j = json.loads(l); print(j['scene'], j['rays'], j['n_pts'], j['ms'], j['roofline']['frac'], j['status'], j.get('shadowed_samples'))"
