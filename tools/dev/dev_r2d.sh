#!/bin/bash
# dev (GPU box): full GPU suite on the in-tree build + instanced scaling experiment
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gputests.log
for n in 16384 65536; do for p in float32 fp16x3; do
  NTX_INSTANCED_RAYS=$n timeout 300 python bench.py --steps 10 --warmup 2 --workload carpet_instanced --precision $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT n=$n $p', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" >> $O/inst.log 2>&1
done; done
timeout 300 python bench.py --steps 5 --warmup 1 > $O/bench_default.json 2>$O/bench_default.err
cat $O/gputests.log $O/inst.log; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['roofline']['frac'], d['fp16x3']['kernel_ms'], d['perturb']['kernel_ms']); print(d['cpu_baseline'])"
