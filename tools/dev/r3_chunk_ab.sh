#!/bin/bash
# dev (GPU box): interleaved A/B of the instance kernel's knobs at one chunk size (default 16 384 rays); NERFTEX_DEBUG_RUNS bits:
# 0 run rows, 1 flags only, 2 groups of one batch, 3 single-ray claims, 4 no gather prefetch, 8-12 / 16-20 chunk thresholds ta / tb
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_instance.py -m gpu -x -q 2>&1 | tail -2
run() { "$@" timeout 300 python bench.py --workload carpet_instanced --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
run env A=default
run env NERFTEX_DEBUG_RUNS=17
run env NERFTEX_DEBUG_RUNS=9
run env NTX_INSTANCED_RAYS=65536
run env NTX_INSTANCED_RAYS=65536 NERFTEX_DEBUG_RUNS=17
done
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('carpet', d['ms_per_step'], d['roofline']['frac'], d.get('parity'))"
