#!/bin/bash
# dev helper (GPU box): fp16x3 parity (carpet) + bench for each dev build build_dev/<name>.so given as arguments
for n in "$@"; do
export NERFTEX_LIB=$GRAFT_REPO_ROOT/build_dev/$n.so
timeout 300 python -m pytest tests/test_gpu_fp16x3.py -m gpu -q -x -k "carpet or zvals" 2>&1 | tail -1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision fp16x3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $n', d['value'], d['roofline']['kernel_ms'])"
done
