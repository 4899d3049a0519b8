#!/bin/bash
# dev (GPU box): first GPU pass of round 3 -- new tests first, then the whole suite, then bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_instance.py -m gpu -x -q 2>&1 | tail -25 > $O/inst.log; cat $O/inst.log | tail -8
for w in "" "--instanced-per-sample-dirs"; do
  timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 $w 2>$O/inst_bench.err | tee -a $O/inst_bench.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('INST $w', d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
NERFTEX_NO_DIR_HOIST=1 timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>>$O/inst_bench.err | tee -a $O/inst_bench.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('INST nohoist', d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>>$O/inst_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('INST again', d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_mip.py tests/test_gpu_fp16x3.py -m gpu -q 2>&1 | tail -15 > $O/rest.log; tail -6 $O/rest.log
