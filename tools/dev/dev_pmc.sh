#!/bin/bash
# dev helper (GPU box): PMC counters of the render kernel for the current build
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_dev; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision ${PREC:-float32}"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_IFETCH --output-format csv -d $O/b -o b -- $B > $O/b.log 2>&1
python - <<'PY'
import csv,collections,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_dev"
for f in sorted(glob.glob(O+"/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("render_kernel_x3<" if os.environ.get("PREC") == "fp16x3" else "render_kernel<") in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print("PMC %-28s %.5g"%(k,sum(v)/len(v)))
PY
