#!/bin/bash
# dev (GPU box): round-2 verification run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O
( nproc; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)" ) > $O/cpuinfo.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputests.log
for lib in default h16; do
  if [ $lib = default ]; then unset NERFTEX_LIB; else export NERFTEX_LIB=$R/build_dev/r2/$lib.so; fi
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RESULT $lib', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['fp16x3']['kernel_ms'], d['perturb']['kernel_ms'])" >> $O/iters.log 2>&1
done
unset NERFTEX_LIB
for w in fur_sharded grass_filtered_sharded; do timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $w > $O/bench_$w.json 2>$O/bench_$w.err; done
for p in fp16x3 float32; do for wt in glorot zero; do
  timeout 120 python tools/power_trace.py $O/power_${p}_${wt}.csv --period 0.01 -- python tools/dev/dev_loop.py $p 6 $wt >> $O/power.log 2>&1
done; done
cat $O/cpuinfo.txt; cat $O/gputests.log; cat $O/iters.log; tail -c 600 $O/bench_fur_sharded.json; tail -c 600 $O/bench_grass_filtered_sharded.json; grep -E "LOOP|POWER_TRACE" $O/power.log
