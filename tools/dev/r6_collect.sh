#!/bin/bash
# build container, after `gpurun -- bash tools/dev/r6_final.sh`: gpurun_out/ -> profiles/r06/ (the files the docs and bench.py quote)
set -e
cd "$(dirname "$0")/../.."
P=profiles/r06; O=gpurun_out; mkdir -p $P
for w in carpet grass fur grass_filtered fur_sharded grass_filtered_sharded; do python tools/summarize_profile.py pg_$w r06 bench_${w}_v17 > /dev/null; done
python tools/summarize_profile.py pg_instanced r06 bench_instanced_v17 "instance_kernel<" > /dev/null
python tools/summarize_profile.py pg_instanced_scene r06 bench_instanced_scene "instance_kernel<" > /dev/null
cp $O/r6trainprof/train_step_kernel_stats.csv $O/r6trainprof/train_step_pmc_summary.json $O/r6trainprof/train_step_timeline.txt $P/
cp $O/r6inst/instancer_base_kernel_stats.csv $O/r6inst/instancer_base_pmc_summary.json $P/ 2>/dev/null || true
cp $O/r6_tree.txt $P/tree.txt
ls $P
