#!/bin/bash
# GPU box: profile campaign of build v15 (instance kernel on bundles of rays; the render kernels are v14's) -- the instanced bench
# lines and their knobs on one box, rocprofv3 passes of the instanced workload, and the headline line + kernel trace as a check
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=$O/r3v15; mkdir -p $L
cd $R
I="timeout 300 python bench.py --workload carpet_instanced --steps 30 --warmup 5"
$I > $L/bench_instanced_float32.json 2>/dev/null
NERFTEX_DEBUG_RUNS=9 $I > $L/bench_instanced_float32_single_ray_claims.json 2>/dev/null
NERFTEX_DEBUG_RUNS=17 $I > $L/bench_instanced_float32_no_prefetch.json 2>/dev/null
NERFTEX_NO_DIR_HOIST=1 $I > $L/bench_instanced_float32_no_run_rows.json 2>/dev/null
$I --instanced-per-sample-dirs > $L/bench_instanced_float32_per_sample_dirs.json 2>/dev/null
$I --precision fp16x3 > $L/bench_instanced_fp16x3.json 2>/dev/null
for n in 2048 4096 8192 16384 32768 65536; do
  NTX_INSTANCED_RAYS=$n $I 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'rays': $n, 'claims': 'chunks', 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))"
  NTX_INSTANCED_RAYS=$n NERFTEX_DEBUG_RUNS=9 $I 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'rays': $n, 'claims': 'single rays', 'kernel_ms': d['roofline']['kernel_ms'], 'frac': d['roofline']['frac']}))"
done > $L/inst_chunk_sizes.jsonl
timeout 400 python bench.py --steps 5 --warmup 1 > $L/bench_carpet.json 2>/dev/null
timeout 600 python tools/bench_small_kernels.py > $L/small_kernels.jsonl 2>/dev/null
STEPS=30 WARMUP=5 bash tools/profile_bench.sh ph_instanced --workload carpet_instanced > /dev/null 2>&1
ONLY="kt pmc1 pmc3" STEPS=30 WARMUP=5 NERFTEX_DEBUG_RUNS=9 bash tools/profile_bench.sh ph_instanced_single --workload carpet_instanced > /dev/null 2>&1
ONLY="kt" bash tools/profile_bench.sh ph_carpet --workload carpet > /dev/null 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -path "*ph_*" -name "*agent_info.csv" -delete 2>/dev/null
ls $O | grep -c "^ph_"; du -sh $O | tail -1; cat $L/inst_chunk_sizes.jsonl
