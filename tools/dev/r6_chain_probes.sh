#!/bin/bash
# GPU box: per-kernel times of the training step for development builds (tools/dev/r6_variant.sh): r6_chain_probes.sh NAME [NAME ...] ("shipped" = the in-tree library)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probes; mkdir -p $O; cd $R
P="python bench.py --workload carpet_train_step --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-parity"
for V in "$@"; do
  L=""; [ $V != shipped ] && L="$R/build_dev/libntx_$V.so"
  NERFTEX_LIB=$L timeout 120 python bench.py --workload carpet_train_step --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', 'ms_per_step', round(d['ms_per_step'],4))"
  rm -rf $O/kt_$V
  NERFTEX_LIB=$L timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$V -o kt -- $P > /dev/null 2>&1
  python - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("$O/kt_$V/**/*kernel_stats.csv", recursive=True)[0])):
    if any(k in r["Name"] for k in ("fwd_chain", "dx_chain", "dw_kernel")): print("  $V", r["Name"].split("(")[0][-28:], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
PY
done
