#!/bin/bash
# dev (GPU box): where the instanced kernel's time goes -- knob sweeps on one box
cd $GRAFT_REPO_ROOT
run() { "$@" timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; }
run env A=1
run env NERFTEX_NO_DIR_HOIST=1
run env NERFTEX_DEBUG_RUNS=3
run env NERFTEX_DEBUG_RUNS=5
run env NTX_INSTANCED_RAYS=65536
run env NTX_INSTANCED_RAYS=65536 NERFTEX_NO_DIR_HOIST=1
run env NTX_INSTANCED_RAYS=4096
run env A=2
