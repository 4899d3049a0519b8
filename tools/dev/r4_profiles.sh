#!/bin/bash
# GPU box: round 4's profile campaign on ONE box -> gpurun_out/r4lines, gpurun_out/ph_*, gpurun_out/r4inst
#   bench lines of every BASELINE workload + the instanced ones; rocprofv3 passes of the headline and of the instanced scene;
#   the instancer alone: base / shadows / textures, kernel trace + counters each, and the scaling table 2.3 k .. 10^5 patches
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; L=$O/r4lines; I=$O/r4inst; mkdir -p $L $I; cd $R
python tools/source_hash.py > $L/tree.txt
for w in carpet grass fur grass_filtered; do
  timeout 400 python bench.py --steps 5 --warmup 1 --workload $w > $L/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py > $L/bench_carpet_default.json 2>/dev/null
for w in fur_sharded grass_filtered_sharded; do
  timeout 400 python bench.py --steps 3 --warmup 1 --workload $w --no-cpu-baseline > $L/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced > $L/bench_instanced_float32.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --workload carpet_instanced_scene > $L/bench_instanced_scene.json 2>/dev/null
bash tools/profile_bench.sh ph_carpet --workload carpet > /dev/null 2>&1
STEPS=12 WARMUP=3 bash tools/profile_bench.sh ph_instanced_scene --workload carpet_instanced_scene > /dev/null 2>&1
# the instancer alone
B="timeout 600 python tools/bench_instancer.py"
$B 2>/dev/null | grep "^{" > $I/instancer_bench.jsonl
{ $B --grid 64 --no-render; $B --method random --no-render; $B --method nearest_blend --no-render; $B --no-mesh --no-render;
  $B --shadows 128 --no-render; $B --shadows 100000 --no-render; $B --textures 256 --no-render; $B --textures 100000 --no-render;
  $B --shadows 128 --textures 256 --no-render; $B --rays 65536 --no-render; $B --rays 4096 --no-render; $B --samples 256 --step-size 0.008 --no-render; } 2>/dev/null | grep "^{" > $I/instancer_variants.jsonl
for G in 48 100 200 316; do
  for RAYS in 16384 65536; do
    $B --grid $G --scale-with-grid --rays $RAYS --steps 10 2>/dev/null | grep "^{" | head -2
    $B --grid $G --scale-with-grid --rays $RAYS --steps 5 --shadows 128 2>/dev/null | grep "^{" | head -2
    $B --grid $G --scale-with-grid --rays $RAYS --steps 5 --textures 256 --no-render 2>/dev/null | grep "^{"
  done
done > $I/instancer_scaling.jsonl
prof() {   # prof <tag> <bench_instancer args...>
  T=$1; shift
  P="python tools/bench_instancer.py --no-render --steps 10 $*"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $I/kt_$T -o kt -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $I/pmc_sq_$T -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $I/pmc_sq2_$T -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $I/pmc_sq3_$T -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $I/pmc_w_$T -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $I/pmc_r_$T -o p -- $P > /dev/null 2>&1
}
prof base
prof shadow_128 --shadows 128
prof shadow_per_step --shadows 100000
prof texture_256 --textures 256
prof shadow_texture --shadows 128 --textures 256
prof scale_1e5 --grid 316 --scale-with-grid
prof scale_1e5_shadow --grid 316 --scale-with-grid --shadows 128
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $I -name "*kernel_trace.csv" -path "*pmc*" -delete
python tools/dev/r4_reduce_inst.py $I base shadow_128 shadow_per_step texture_256 shadow_texture scale_1e5 scale_1e5_shadow | cut -c1-200
du -sh $O | tail -1
