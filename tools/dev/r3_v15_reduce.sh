#!/bin/bash
# here (after tools/dev/r3_v15.sh on the GPU box): gpurun_out/ph_* and gpurun_out/r3v15 -> profiles/r03 (build v15)
cd /root/repo
python tools/summarize_profile.py ph_instanced r03 bench_instanced_v15 "instance_kernel<" > /dev/null && echo instanced
python tools/summarize_profile.py ph_instanced_single r03 bench_instanced_single_ray_claims_v15 "instance_kernel<" > /dev/null && echo single
cp gpurun_out/ph_carpet_kt/*kernel_stats.csv profiles/r03/bench_carpet_v15_kernel_stats.csv
cp gpurun_out/r3v15/*.json gpurun_out/r3v15/*.jsonl profiles/r03/
