#!/bin/bash
# GPU box: round 3's FINAL profile campaign (build v14: flex families, batched RGBA stores) -- bench lines, then rocprofv3 passes
# (tools/profile_bench.sh) of every BASELINE workload at float32, the instanced tail, the flex kernels on the headline workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=$O/r3final; mkdir -p $L
cd $R
for w in carpet grass fur grass_filtered; do
  timeout 400 python bench.py --steps 5 --warmup 1 --workload $w > $L/bench_$w.json 2>/dev/null
done
for w in fur_sharded grass_filtered_sharded; do
  timeout 400 python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline > $L/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --steps 5 --warmup 1 --perturb --no-cpu-baseline --no-extras > $L/bench_carpet_perturb.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 1 --perturb --raw-noise-std 0.5 --no-cpu-baseline --no-extras > $L/bench_carpet_perturb_noise.json 2>/dev/null
NERFTEX_FORCE_FLEX=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $L/bench_carpet_flex.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced > $L/bench_instanced_float32.json 2>/dev/null
NERFTEX_NO_DIR_HOIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced > $L/bench_instanced_float32_no_run_rows.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced --precision fp16x3 > $L/bench_instanced_fp16x3.json 2>/dev/null
NTX_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --workload fur_sharded --no-cpu-baseline > $L/bench_fur_sharded_2ranks_one_gpu.json 2>/dev/null
timeout 600 python tools/bench_small_kernels.py > $L/small_kernels.jsonl 2>/dev/null
for w in carpet grass fur grass_filtered fur_sharded grass_filtered_sharded; do
  bash tools/profile_bench.sh pg_$w --workload $w > /dev/null 2>&1
done
# the instanced kernel is short (19 ms): 23 launches per pass, so that the first, cold one does not carry the average
STEPS=20 WARMUP=3 bash tools/profile_bench.sh pg_instanced --workload carpet_instanced > /dev/null 2>&1
ONLY="kt pmc1 pmc3" NERFTEX_FORCE_FLEX=1 bash tools/profile_bench.sh pg_carpet_flex --workload carpet > /dev/null 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -path "*pg_*" -name "*agent_info.csv" -delete 2>/dev/null
ls $O | grep -c "^pg_"; du -sh $O | tail -1
