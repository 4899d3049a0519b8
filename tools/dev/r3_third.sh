#!/bin/bash
# dev (GPU box): third pass of round 3 -- full GPU suite on the rebuilt library, instanced kernel-trace with 20 launches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/suite.log; tail -4 $O/suite.log
STEPS=20 WARMUP=3 ONLY="kt" bash tools/profile_bench.sh pi_instanced --workload carpet_instanced > /dev/null 2>&1
grep instance_kernel gpurun_out/pi_instanced_kt/kt_kernel_stats.csv | cut -c1-160
timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('INST', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"
timeout 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('carpet', round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],4), d['parity']['rel_linf_f32'])"
find gpurun_out -name "*.db" -delete 2>/dev/null
