#!/bin/bash
# GPU box: the weight-gradient kernel with its job's waves meeting every 16 (shipped), 8 or 4 blocks -- kernel time and HBM-side read bytes.
# The variants are built in the container into build_dev/libntx_s{4,8}.so (train_chain_5.o with -DNTX_DW_SYNC=...).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dwsync; mkdir -p $O; cd $R
P="python bench.py --workload carpet_train_step --steps 5 --warmup 1 --no-cpu-baseline"
for V in shipped s8 s4; do
  L=""; [ $V != shipped ] && L="$R/build_dev/libntx_$V.so"
  NERFTEX_LIB=$L timeout 120 python bench.py --workload carpet_train_step --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', 'ms_per_step', round(d['ms_per_step'],4))"
  NERFTEX_LIB=$L timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$V -o kt -- $P > /dev/null 2>&1
  NERFTEX_LIB=$L timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$V -o p -- $P > /dev/null 2>&1
  python - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("$O/kt_$V/*kernel_stats.csv")[0])):
    if "dw_kernel" in r["Name"]: print("$V", "dw_kernel us", round(float(r["AverageNs"]) / 1e3, 1))
v = [float(r["Counter_Value"]) for f in glob.glob("$O/pmc_$V/*counter_collection.csv") for r in csv.DictReader(open(f)) if "dw_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$V", "dw_kernel HBM-side read GB", round(sum(v) / len(v) * 1024 * 2 / 1e9, 3))
PY
done
