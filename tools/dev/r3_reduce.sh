#!/bin/bash
# here (after tools/dev/r3_final.sh on the GPU box): gpurun_out/pg_* and gpurun_out/r3final -> profiles/r03 (build v14)
cd /root/repo
for w in carpet grass fur grass_filtered fur_sharded grass_filtered_sharded; do
  python tools/summarize_profile.py pg_$w r03 bench_${w}_v14 > /dev/null && echo $w
done
python tools/summarize_profile.py pg_instanced r03 bench_instanced_v14 "instance_kernel<" > /dev/null && echo instanced
python tools/summarize_profile.py pg_carpet_flex r03 bench_carpet_flex_v14 > /dev/null && echo flex
cp gpurun_out/r3final/*.json gpurun_out/r3final/*.jsonl profiles/r03/
