#!/bin/bash
# dev helper (GPU box): quick parity (carpet only) + bench line
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(mlp_forward and ParamNerf-npar0) or (render_rays_camera and carpet) or full_size" 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
