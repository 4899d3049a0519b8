#!/bin/bash
# GPU box: round 3's profile campaign -- bench lines + rocprofv3 passes (tools/profile_bench.sh) of every BASELINE workload (float32)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=$O/r3lines; mkdir -p $L
cd $R
for w in carpet grass fur grass_filtered; do
  timeout 400 python bench.py --steps 5 --warmup 1 --workload $w > $L/bench_$w.json 2>/dev/null
done
for w in fur_sharded grass_filtered_sharded; do
  timeout 400 python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline > $L/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --steps 5 --warmup 1 --perturb --no-cpu-baseline --no-extras > $L/bench_carpet_perturb.json 2>/dev/null
timeout 300 python bench.py --steps 5 --warmup 1 --perturb --raw-noise-std 0.5 --no-cpu-baseline --no-extras > $L/bench_carpet_perturb_noise.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced > $L/bench_instanced_float32.json 2>/dev/null
NERFTEX_NO_DIR_HOIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced > $L/bench_instanced_float32_no_run_rows.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced --instanced-per-sample-dirs > $L/bench_instanced_float32_per_sample_dirs.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced --precision fp16x3 > $L/bench_instanced_fp16x3.json 2>/dev/null
timeout 600 python tools/bench_small_kernels.py > $L/small_kernels.jsonl 2>/dev/null
for w in carpet grass fur grass_filtered fur_sharded grass_filtered_sharded; do
  bash tools/profile_bench.sh pf_$w --workload $w > /dev/null 2>&1
done
bash tools/profile_bench.sh pf_instanced --workload carpet_instanced > /dev/null 2>&1
ONLY="kt pmc1 pmc3" NERFTEX_NO_DIR_HOIST=1 bash tools/profile_bench.sh pf_instanced_norows --workload carpet_instanced > /dev/null 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -path "*pf_*" -name "*agent_info.csv" -delete 2>/dev/null
ls $O | grep -c "^pf_"; du -sh $O | tail -1
