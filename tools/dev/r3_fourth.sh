#!/bin/bash
# dev (GPU box): the flex family -- its tests, the whole suite, and what the layer loop costs against the straight-line kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_flex.py -m gpu -x -q 2>&1 | tail -25 > $O/flex.log; tail -8 $O/flex.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],4), d.get('parity',{}).get('rel_linf_f32'))"; }
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | line tuned
NERFTEX_FORCE_GENERIC=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | line generic
NERFTEX_NO_DIR_HOIST=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | line tuned_nohoist
NERFTEX_FORCE_FLEX=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>$O/flex_bench.err | line flex
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/suite.log; tail -4 $O/suite.log
