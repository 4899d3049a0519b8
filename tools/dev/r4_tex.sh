#!/bin/bash
# GPU box: the texture flavour of the march kernel on the bench scene
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4tex; mkdir -p $O; cd $R
B="timeout 300 python tools/bench_instancer.py --no-render"
{ $B; $B --textures 256; $B --textures 100000; $B --textures 256 --shadows 128; } 2>/dev/null | grep "^{" > $O/tex.jsonl
P="python tools/bench_instancer.py --no-render --steps 10 --textures 256"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_tex -o kt -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_tex -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2_tex -o p -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq3_tex -o p -- $P > /dev/null 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -path "*pmc*" -delete
python tools/dev/r4_reduce_inst.py $O tex
python -c "
import json
for l in open('$O/tex.jsonl'): j = json.loads(l); print(j['scene'], j['rays'], j['ms'], j['status'])"
