#!/bin/bash
# GPU box: every bench line, printed AFTER the counter summaries it quotes were taken on this very tree and committed under profiles/r06/
R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/r6lines; mkdir -p $L; cd $R
for w in carpet grass fur grass_filtered; do timeout 400 python bench.py --steps 5 --warmup 1 --workload $w > $L/bench_$w.json 2>/dev/null; done
timeout 300 python bench.py > $L/bench_carpet_default.json 2>/dev/null
for w in fur_sharded grass_filtered_sharded; do timeout 400 python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline > $L/bench_$w.json 2>/dev/null; done
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced > $L/bench_instanced_float32.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced_scene > $L/bench_instanced_scene.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_train_step > $L/bench_train_step.json 2>/dev/null
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r6lines/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(os.path.basename(f).ljust(38), "ms %.3f" % d["ms_per_step"], "frac %.4f" % r["frac"], "traffic", r.get("traffic"), "current:", r.get("traffic_profile_current"), (d.get("instancer") or {}).get("roofline", {}).get("traffic_profile_current"))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
