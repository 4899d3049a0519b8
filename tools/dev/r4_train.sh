#!/bin/bash
# GPU box: the training step: bench line + kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4train; mkdir -p $O; cd $R
timeout 600 python bench.py --workload carpet_train_step --steps 10 --warmup 2 > $O/bench_train_step.json 2> $O/err.txt; tail -3 $O/err.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --workload carpet_train_step --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $O/kt/*kernel_stats.csv $O/train_step_kernel_stats.csv
python - <<'PY'
import csv, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4train")
d = json.load(open(f"{O}/bench_train_step.json")); print(d["ms_per_step"], d["roofline"]["frac"], d["value"], d.get("cpu_baseline", {}).get("value"))
tot = 0
for r in csv.DictReader(open(f"{O}/train_step_kernel_stats.csv")):
    if "ntx" in r["Name"]: print(r["Name"][:70].ljust(72), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
