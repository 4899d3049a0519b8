#!/bin/bash
# build container, after `gpurun -- bash tools/dev/r4_final.sh`: gpurun_out/ -> profiles/r04/ (the files the docs and bench.py quote)
set -e
cd "$(dirname "$0")/../.."
P=profiles/r04; O=gpurun_out
python tools/summarize_profile.py ph_carpet r04 bench_carpet_v16 > /dev/null
python tools/summarize_profile.py ph_instanced_scene r04 bench_instanced_scene "instance_kernel<" > /dev/null
cp $O/r4lines/bench_*.json $P/
cp $O/r4inst/instancer_*_kernel_stats.csv $O/r4inst/instancer_*_pmc_summary.json $O/r4inst/instancer_bench.jsonl $O/r4inst/instancer_variants.jsonl $O/r4inst/instancer_scaling.jsonl $P/
cp $O/r4trainprof/bench_train_step.json $O/r4trainprof/train_step_kernel_stats.csv $O/r4trainprof/train_step_pmc_summary.json $O/r4trainprof/train_step_timeline.txt $P/
python tools/source_hash.py > $P/tree.txt
python - <<'PY'
import glob, json
for f in sorted(glob.glob("profiles/r04/bench_*.json")):
    if "pmc_summary" in f: continue
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f.split("/")[-1].ljust(40), "ms %.2f" % d["ms_per_step"], "frac %.4f" % r["frac"], "traffic current:", r.get("traffic_profile_current"))
PY
