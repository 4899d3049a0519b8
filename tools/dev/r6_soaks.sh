#!/bin/bash
# GPU box: the three 600-case soaks of the training step on the round's kernels (tools/dev/soak_train.py) and the training-step lines of the other
# two shipped families -> gpurun_out/r6_soak/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_soak; mkdir -p $O; cd $R
python tools/source_hash.py > $O/tree.txt
for w in fur grass_filtered; do timeout 300 python bench.py --steps 20 --warmup 3 --workload ${w}_train_step > $O/bench_train_step_$w.json 2>/dev/null; done
for S in 1 2 3; do
  timeout 1500 python tools/dev/soak_train.py --cases 600 --seed $S > $O/soak_train_seed$S.txt 2>&1
  tail -n 1 $O/soak_train_seed$S.txt
done
