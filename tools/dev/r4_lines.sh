#!/bin/bash
# GPU box: the bench lines that quote a committed PMC summary, printed AFTER those summaries were taken on this very tree
R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/r4lines2; mkdir -p $L; cd $R
timeout 400 python bench.py --steps 5 --warmup 1 --workload carpet > $L/bench_carpet.json 2>/dev/null
timeout 300 python bench.py > $L/bench_carpet_default.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced_scene > $L/bench_instanced_scene.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_train_step > $L/bench_train_step.json 2>/dev/null
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4lines2/*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(os.path.basename(f), d["ms_per_step"], r["frac"], r.get("traffic"), r.get("traffic_profile_current"), (d.get("instancer") or {}).get("roofline", {}).get("traffic_profile_current"))
PY
