#!/bin/bash
# GPU box: the whole GPU suite, then the instancer's fuzz with many seeds, plain and with the block level of the cull forced on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4soak; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/suite.txt
NTX_INSTANCER_FUZZ_SEEDS=${SEEDS:-2000} timeout 1500 python -m pytest tests/test_gpu_instancer.py -q -k fuzz 2>&1 | tail -3 > $O/soak_plain.txt
NERFTEX_INST_FORCE_BLOCKS=1 NTX_INSTANCER_FUZZ_SEEDS=${SEEDS:-2000} timeout 1500 python -m pytest tests/test_gpu_instancer.py -q 2>&1 | tail -3 > $O/soak_forced_blocks.txt
cat $O/suite.txt $O/soak_plain.txt $O/soak_forced_blocks.txt
