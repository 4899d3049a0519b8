#!/bin/bash
# dev (GPU box): instanced kernels with packed tails vs the in-tree build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O
export NERFTEX_LIB=$R/build_dev/r2/tails.so
timeout 900 python -m pytest tests/test_gpu_instance.py tests/test_gpu_mip.py -m gpu -x -q -k "carpet or npar0 or packed or lockstep or plain or mip" 2>&1 | tail -15 > $O/tests.log
for lib in default tails; do
  if [ $lib = default ]; then unset NERFTEX_LIB; else export NERFTEX_LIB=$R/build_dev/r2/$lib.so; fi
  for p in float32 fp16x3; do
    timeout 300 python bench.py --steps 10 --warmup 2 --workload carpet_instanced --precision $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $lib $p', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" >> $O/inst.log 2>&1
  done
done
cat $O/tests.log $O/inst.log
