#!/bin/bash
# dev helper (GPU box): fp16x3 parity + bench lines at both precisions
timeout 600 python -m pytest tests/test_gpu_fp16x3.py -m gpu -q -x 2>&1 | tail -8
for p in float32 fp16x3; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --precision $p 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $p', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
done
