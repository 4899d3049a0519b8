#!/bin/bash
# GPU box: round 6's profile set on the FINAL tree, one box.  rocprofv3 passes first (tools/profile_bench.sh: kernel trace + the counter
# passes each bench line quotes) for every BASELINE workload, the sharded pair, the instanced tail and scene; the training step's set
# (tools/dev/r6_train_profiles.sh).  tools/dev/r6_collect.sh reduces them into
# profiles/r06/ in the build container; tools/dev/r6_lines.sh then prints the bench lines against the summaries just committed.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python tools/source_hash.py > $O/r6_tree.txt
bash tools/profile_bench.sh pg_carpet --workload carpet > /dev/null 2>&1
for w in grass fur grass_filtered fur_sharded grass_filtered_sharded; do
  ONLY="kt pmc1 pmc2 pmc5 pmc3" bash tools/profile_bench.sh pg_$w --workload $w > /dev/null 2>&1
done
ONLY="kt pmc1 pmc2 pmc5 pmc3" STEPS=12 WARMUP=3 bash tools/profile_bench.sh pg_instanced --workload carpet_instanced > /dev/null 2>&1
ONLY="kt pmc1 pmc2 pmc5 pmc3" STEPS=12 WARMUP=3 bash tools/profile_bench.sh pg_instanced_scene --workload carpet_instanced_scene > /dev/null 2>&1
bash tools/dev/r6_train_profiles.sh > $O/r6_train_profiles.log 2>&1
# the instancer alone (base: what bench.py's instanced-scene line quotes), with counters
I=$O/r6inst
P="python tools/bench_instancer.py --no-render --steps 10 --sparse"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $I/kt_base -o kt -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $I/pmc_sq_base -o p -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $I/pmc_sq2_base -o p -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $I/pmc_sq3_base -o p -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $I/pmc_w_base -o p -- $P > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $I/pmc_r_base -o p -- $P > /dev/null 2>&1
python tools/dev/r4_reduce_inst.py $I base | cut -c1-200
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete
tail -4 $O/r6_train_profiles.log; ls $O | grep -c "^pg_"; du -sh $O | tail -1
