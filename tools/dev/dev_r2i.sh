#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/gputests.log
cat $O/gputests.log
for w in carpet grass_filtered; do
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $w', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['fp16x3']['kernel_ms'], d['perturb']['kernel_ms'])"
done
timeout 300 python bench.py --steps 10 --warmup 2 --workload carpet_instanced 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT inst', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
