#!/bin/bash
# GPU box: the training step's profile set -> gpurun_out/r6trainprof (bench line, kernel trace, counters of the two MFMA kernels)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6trainprof; mkdir -p $O; cd $R
python tools/source_hash.py > $O/tree.txt
timeout 120 python bench.py --workload carpet_train_step --steps 20 --warmup 3 > $O/bench_train_step.json 2> $O/err.txt
P="python bench.py --workload carpet_train_step --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-parity"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $P > /dev/null 2>&1
cp $O/kt/*kernel_stats.csv $O/train_step_kernel_stats.csv
python tools/dev/r6_timeline.py $O/kt/kt_kernel_trace.csv > $O/train_step_timeline.txt
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc1 -o p -- $P > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -o p -- $P > /dev/null 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -o p -- $P > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -o p -- $P > /dev/null 2>&1
python - <<'PY'
import collections, csv, glob, json, os, subprocess
R = os.environ["GRAFT_REPO_ROOT"]; O = os.path.join(R, "gpurun_out", "r6trainprof")
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ntx_train" not in n: continue
        key = n.replace("void ", "").replace("ntx_train::", "").split("(")[0]
        agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
dur = {}
for r in csv.DictReader(open(f"{O}/train_step_kernel_stats.csv")):
    if "ntx_train" in r["Name"]: dur[r["Name"].replace("void ", "").replace("ntx_train::", "").split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
out = {"what": "rocprofv3 --pmc passes over `bench.py --workload carpet_train_step --steps 5 --warmup 1`: per-launch averages of the trainer's kernels",
       "tree": json.loads(open(f"{O}/tree.txt").read()) if open(f"{O}/tree.txt").read().strip().startswith("{") else open(f"{O}/tree.txt").read().strip(),
       "kernels": {}}
for k in sorted({k for k, _ in agg}):
    c = {n: sum(v) / len(v) for (kk, n), v in agg.items() if kk == k}
    e = {"launches_per_5_steps": dur.get(k, (None, None))[0], "average_us": dur.get(k, (None, None))[1], "counters": c}
    if c.get("SQ_INSTS_MFMA"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        e["mfma_busy_fraction_of_kernel_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
        e["mean_wave_lifetime_fraction_of_kernel_cycles"] = c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"] / cyc
        e["valu_instructions_per_mfma_beside_it"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    if "FETCH_SIZE" in c: e["hbm_read_MB"] = c["FETCH_SIZE"] * 1024 * 2 / 1e6          # KB; x 2: gfx950 tallies 128-byte requests at 64 (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in c: e["hbm_written_MB"] = c["WRITE_SIZE"] * 1024 / 1e6
    out["kernels"][k] = e
json.dump(out, open(f"{O}/train_step_pmc_summary.json", "w"), indent=1)
d = json.load(open(f"{O}/bench_train_step.json")); print(d["ms_per_step"], d["roofline"]["frac"], d["value"], d.get("cpu_baseline", {}).get("value"))
for k, e in out["kernels"].items():
    if "mfma_busy_fraction_of_kernel_cycles" in e: print(k, round(e["average_us"], 1), "us; MFMA busy", round(e["mfma_busy_fraction_of_kernel_cycles"], 3), "; read", round(e.get("hbm_read_MB", 0)), "MB; written", round(e.get("hbm_written_MB", 0)), "MB")
PY
