#!/bin/bash
# GPU box: counters of the trainer's rows_kernel inside one training step -> gpurun_out/r4rowspmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4rowspmc; mkdir -p $O; cd $R
P="python bench.py --workload carpet_train_step --steps 2 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $O/pmc3 -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc4 -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5 -o p -- $P > /dev/null 2>&1
python - <<'PY'
import collections, csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4rowspmc")
agg = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rows_kernel<8" in r["Kernel_Name"] or "gemm_kernel" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])
            agg[key].append(float(r["Counter_Value"]))
for k in sorted(agg): print(k, len(agg[k]), sum(agg[k]) / len(agg[k]))
for r in csv.DictReader(open(glob.glob(f"{O}/kt/*kernel_stats.csv")[0])):
    if "rows_kernel" in r["Name"] or "gemm_kernel" in r["Name"]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3)
PY
