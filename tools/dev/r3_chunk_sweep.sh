#!/bin/bash
# dev (GPU box): the chunked hand-out against single rays over chunk sizes; NERFTEX_DEBUG_RUNS = 1 | 8 (single rays) | ta << 8 | tb << 16
cd $GRAFT_REPO_ROOT
run() { "$@" timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; }
for n in 2048 4096 8192 12288 16384 32768 65536; do
run env NTX_INSTANCED_RAYS=$n
run env NTX_INSTANCED_RAYS=$n NERFTEX_DEBUG_RUNS=9
done
