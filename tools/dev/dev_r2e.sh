#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_instance.py tests/test_gpu_mip.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.log
for n in 16384 65536; do for p in float32 fp16x3; do
  NTX_INSTANCED_RAYS=$n timeout 300 python bench.py --steps 10 --warmup 2 --workload carpet_instanced --precision $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT n=$n $p', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['ms_per_step'])" >> $O/inst.log 2>&1
done; done
cat $O/tests.log $O/inst.log
