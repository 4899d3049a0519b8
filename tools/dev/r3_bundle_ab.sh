#!/bin/bash
# dev (GPU box): the instanced kernel with bundles of rays -- parity first, then same-box A/B and a kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_instance.py -m gpu -x -q 2>&1 | tail -5
run() { "$@" timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; }
run env A=1
run env NERFTEX_DEBUG_RUNS=9
run env NERFTEX_NO_DIR_HOIST=1
run env NTX_INSTANCED_RAYS=65536
run env NTX_INSTANCED_RAYS=65536 NERFTEX_DEBUG_RUNS=9
run env NTX_INSTANCED_RAYS=4096
run env A=2
STEPS=20 WARMUP=3 ONLY=kt bash tools/profile_bench.sh inst_bundle --workload carpet_instanced
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/inst_bundle_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ntx::' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
