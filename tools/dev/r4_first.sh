#!/bin/bash
# GPU box, round 4, first campaign: where the instancer stands before this round's work --
#   (1) the shadow kernel under rocprofv3 (kernel trace + SQ / memory counters in their own passes) at 128 samples per unit and per step
#   (2) the instancer from 2.3 k to 10^5 patches (the same sheet under smaller patches), with and without shadow rays
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r4first}; mkdir -p $O; cd $R
B="timeout 600 python tools/bench_instancer.py"
for G in 48 100 200 316; do
  for RAYS in 16384 65536; do
    $B --grid $G --scale-with-grid --rays $RAYS --steps 10 2>/dev/null | grep "^{" | head -2
    $B --grid $G --scale-with-grid --rays $RAYS --steps 5 --shadows 128 --no-render 2>/dev/null | grep "^{"
  done
done > $O/${SCALING:-instancer_scaling_before}.jsonl
for N in 128 100000; do
  P="python tools/bench_instancer.py --no-render --steps 10 --shadows $N"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$N -o kt -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_$N -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_$N -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_sq3_$N -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$N -o p -- $P > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_r_$N -o p -- $P > /dev/null 2>&1
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
python tools/dev/r4_reduce_inst.py $O 128 100000
cut -c1-330 $O/${SCALING:-instancer_scaling_before}.jsonl
