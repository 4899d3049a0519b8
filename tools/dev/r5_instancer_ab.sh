#!/bin/bash
# GPU box: the instancer -> tail hand-off, dense ten buffers against NTX_OPT_INSTANCER_SPARSE -> gpurun_out/r5inst
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5inst; mkdir -p $O; cd $R
python tools/source_hash.py > $O/tree.txt
timeout 120 python tools/bench_instancer.py --dense-render > $O/dense.jsonl 2> $O/err_dense.txt
timeout 120 python tools/bench_instancer.py --sparse > $O/sparse.jsonl 2> $O/err_sparse.txt
for m in dense sparse; do
  F=""; [ $m = sparse ] && F="--sparse"
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${m}_$c -o p -- python tools/bench_instancer.py --no-render --steps 3 $F > /dev/null 2>&1
  done
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -o kt -- python tools/bench_instancer.py --no-render --steps 3 $F > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5inst")
out = {"what": "ntx_instancer_model_input on the bench scene (48 x 48 patches, 16384 rays x 1024 steps): dense ten buffers vs NTX_OPT_INSTANCER_SPARSE; per launch of the march kernel",
       "tree": open(f"{O}/tree.txt").read().strip()}
for m in ("dense", "sparse"):
    e = {}
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        vals = [float(r["Counter_Value"]) for f in glob.glob(f"{O}/pmc_{m}_{c}/*counter_collection.csv") for r in csv.DictReader(open(f)) if "inst_march" in r["Kernel_Name"]]
        if vals: e[c + "_KB"] = sum(vals) / len(vals)
    if "WRITE_SIZE_KB" in e: e["hbm_written_MB"] = e["WRITE_SIZE_KB"] * 1024 / 1e6
    if "FETCH_SIZE_KB" in e: e["hbm_read_MB"] = e["FETCH_SIZE_KB"] * 1024 * 2 / 1e6
    for f in glob.glob(f"{O}/kt_{m}/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "inst_march" in r["Name"]: e["march_kernel_us"] = float(r["AverageNs"]) / 1e3
    e["lines"] = [json.loads(l) for l in open(f"{O}/{m}.jsonl") if l.startswith("{")][:2]
    out[m] = e
json.dump(out, open(f"{O}/instancer_handoff_ab.json", "w"), indent=1)
for m in ("dense", "sparse"): print(m, {k: v for k, v in out[m].items() if k != "lines"}, [(l.get("ms"), l.get("ms_instancer")) for l in out[m]["lines"]])
PY
