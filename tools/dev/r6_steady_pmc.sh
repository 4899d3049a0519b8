#!/bin/bash
# GPU box: counters of the training step's three matrix-core kernels launch by launch over 30 steps (steady state, not the first launches after idle):
# clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix pipe busy, VALU.   r6_steady_pmc.sh [NAME]  (NAME: a build_dev library, default the shipped one)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=${1:-shipped}; O=$R/gpurun_out/steady_$V; rm -rf $O; mkdir -p $O; cd $R
L=""; [ $V != shipped ] && L="$R/build_dev/libntx_$V.so"
P="python bench.py --workload ${2:-carpet_train_step} --steps 30 --warmup 3 --no-cpu-baseline --no-extras"
NERFTEX_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $O/p1 -o p -- $P > /dev/null 2>&1
python - <<PY
import csv, glob, collections
O = "$O"
tr = {r["Dispatch_Id"]: r for f in glob.glob(O + "/p1/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))}
cnt = collections.defaultdict(dict)
for f in glob.glob(O + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); cnt[r["Dispatch_Id"]]["name"] = r["Kernel_Name"]
for key in ("fwd_chain", "dx_chain", "dw_kernel", "render_kernel", "instance_kernel"):
    rows = []
    for d, c in sorted(cnt.items(), key=lambda kv: int(kv[0])):
        if key not in c["name"] or d not in tr: continue
        us = (int(tr[d]["End_Timestamp"]) - int(tr[d]["Start_Timestamp"])) / 1e3
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        rows.append((us, cyc / us / 1e3, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["SQ_INSTS_MFMA"] * 64 / 1024 / cyc, (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / max(c["SQ_INSTS_MFMA"], 1), c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc if "SQ_ACTIVE_INST_VALU" in c else 0))
    if not rows: continue
    print(key, "launches", len(rows))
    for r in rows[::3]: print("   us %.1f  GHz %.3f  mfma_busy %.3f  mfma_issue_cycles/cycles %.3f  valu/mfma %.3f  active_valu %.3f" % r)
    tail = rows[len(rows) // 2:]
    print("   second half mean: us %.1f GHz %.3f busy %.3f issue %.3f" % tuple(sum(r[i] for r in tail) / len(tail) for i in range(4)))
PY
