#!/bin/bash
# dev (GPU box): parameter branches (param_depth > 0), then the whole suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_flex.py -m gpu -x -q 2>&1 | tail -40 > $O/flex.log; tail -30 $O/flex.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/suite.log; tail -4 $O/suite.log
