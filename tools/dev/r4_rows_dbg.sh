#!/bin/bash
# GPU box: the rows kernel with parts switched off (NERFTEX_ROWS_DEBUG bits: 1 no stagger, 2 no X stream, 4 no weight stream, 8 no epilogue)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4rows; mkdir -p $O; cd $R
for D in ${DBG_LIST:-0}; do
  NERFTEX_ROWS_PHASE=${D#*p} NERFTEX_ROWS_DEBUG=${D%p*} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$D -o kt -- python bench.py --workload carpet_train_step --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - $D <<'PY'
import csv, sys, os
D = sys.argv[1]
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4rows")
tot = n = 0
for r in csv.DictReader(open(f"{O}/kt{D}/kt_kernel_stats.csv")):
    if "rows_kernel<8" in r["Name"]:
        print("debug", D, r["Name"][16:34], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
print("debug", D, "all rows_kernel<8, *>: average", round(tot / n / 1e3, 1), "us")
PY
done
