#!/bin/bash
# dev (GPU box): two ranks sharing the GPU through the whole multi-rank bench; flex fuzz
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q -k "sharing" 2>&1 | tail -30 > $O/share.log; tail -12 $O/share.log
ls logs 2>/dev/null | head; tail -5 logs/rank1.err 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_flex.py -m gpu -x -q 2>&1 | tail -6 > $O/flex.log; tail -3 $O/flex.log
