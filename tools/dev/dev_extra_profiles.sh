#!/bin/bash
# GPU box: power traces (all four render cases again with the card auto-detected, and the instanced kernels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; rm -rf $O/power2; mkdir -p $O/power2
for p in fp16x3 float32; do for wt in glorot zero; do
  timeout 120 python tools/power_trace.py $O/power2/power_${p}_${wt}.csv --period 0.01 -- python tools/dev/dev_loop.py $p 6 $wt >> $O/power2/power.log 2>&1
done; done
for p in float32 fp16x3; do
  timeout 200 python tools/power_trace.py $O/power2/power_instanced_$p.csv --period 0.01 -- python bench.py --workload carpet_instanced --precision $p --steps 700 --warmup 5 >> $O/power2/power.log 2>&1
done
grep -E "LOOP|POWER_TRACE" $O/power2/power.log
