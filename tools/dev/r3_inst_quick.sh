#!/bin/bash
# GPU box: the instancer's parity tests, its bench line and the per-kernel times of one build (development loop of round 3)
export TMPDIR=/tmp
T=${1:-x}
timeout 500 python -m pytest tests/test_gpu_instancer.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -12
timeout 200 python tools/bench_instancer.py --no-render 2>&1 | grep "^{" | python -c "import sys,json; [print({k:j[k] for k in ('ms','rays_per_s')}, j['roofline']['frac']) for j in map(json.loads, sys.stdin)]"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inst_prof_$T -o inst -- python tools/bench_instancer.py --no-render --steps 5 > /dev/null 2>&1
python -c "import csv,sys; [print(r[\"Name\"][10:30], round(float(r[\"AverageNs\"])/1e3,1)) for r in csv.DictReader(open(sys.argv[1])) if \"inst_\" in r[\"Name\"]]" gpurun_out/inst_prof_$T/inst_kernel_stats.csv
