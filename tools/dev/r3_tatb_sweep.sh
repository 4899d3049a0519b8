#!/bin/bash
# GPU box: hand-out thresholds ta / tb of the instance kernel (NERFTEX_DEBUG_RUNS bits 8-12 / 16-20) on both instanced bench workloads
for tatb in "6 3" "3 1" "4 2" "2 1" "8 4" "12 6" "5 2"; do
  set -- $tatb; v=$((1 + ($1 << 8) + ($2 << 16)))
  a=$(NERFTEX_DEBUG_RUNS=$v timeout 120 python bench.py --workload carpet_instanced --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['roofline']['kernel_ms'],3))")
  b=$(NERFTEX_DEBUG_RUNS=$v timeout 120 python bench.py --workload carpet_instanced_scene --steps 30 --warmup 5 --no-parity 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['roofline']['kernel_ms'],3))")
  echo "ta=$1 tb=$2 synthetic_ms=$a scene_ms=$b"
done
