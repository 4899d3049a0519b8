#!/bin/bash
# GPU box: the round's profile campaign -- bench lines + rocprofv3 passes of every BASELINE workload at both precisions
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/r2lines
cd $R
for w in carpet grass fur grass_filtered; do
  timeout 400 python bench.py --steps 5 --warmup 1 --workload $w > $O/r2lines/bench_$w.json 2>/dev/null
  timeout 300 python bench.py --steps 5 --warmup 1 --workload $w --precision fp16x3 --no-cpu-baseline > $O/r2lines/benchx3_$w.json 2>/dev/null
done
for w in fur_sharded grass_filtered_sharded; do
  timeout 300 python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline > $O/r2lines/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --steps 5 --warmup 1 --perturb --no-cpu-baseline > $O/r2lines/bench_carpet_perturb.json 2>/dev/null
for p in float32 fp16x3; do
  timeout 300 python bench.py --steps 10 --warmup 2 --workload carpet_instanced --precision $p > $O/r2lines/bench_instanced_$p.json 2>/dev/null
done
timeout 600 python tools/bench_small_kernels.py > $O/r2lines/small_kernels.jsonl 2>/dev/null
for w in carpet grass fur grass_filtered; do
  bash tools/profile_bench.sh pf_$w --workload $w > /dev/null 2>&1
  bash tools/profile_bench.sh px_$w --workload $w --precision fp16x3 > /dev/null 2>&1
done
bash tools/profile_bench.sh pf_instanced --workload carpet_instanced > /dev/null 2>&1
bash tools/profile_bench.sh px_instanced --workload carpet_instanced --precision fp16x3 > /dev/null 2>&1
# keep the merge small: only the CSVs the summariser reads
find $O -name "*.db" -delete 2>/dev/null; find $O -path "*pf_*" -name "*agent_info.csv" -delete 2>/dev/null
ls $O | grep -c "^p[fx]_"; du -sh $O | tail -1
