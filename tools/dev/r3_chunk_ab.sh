#!/bin/bash
# dev (GPU box): interleaved A/B of the chunked hand-out at one chunk size (default 16 384 rays)
cd $GRAFT_REPO_ROOT
run() { "$@" timeout 300 python bench.py --workload carpet_instanced --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
run env A=default
run env NERFTEX_DEBUG_RUNS=9
run env NERFTEX_DEBUG_RUNS=$((1 + (4 << 8) + (2 << 16)))
run env NERFTEX_DEBUG_RUNS=$((1 + (8 << 8) + (2 << 16)))
run env NERFTEX_DEBUG_RUNS=$((1 + (6 << 8) + (3 << 16)))
done
