#!/bin/bash
# GPU box: the three 600-case soaks of the training step on the round's kernels (tools/dev/soak_train.py) -> gpurun_out/r6_soak/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_soak; mkdir -p $O; cd $R
python tools/source_hash.py > $O/tree.txt
for S in 1 2 3; do
  timeout 1500 python tools/dev/soak_train.py --cases 600 --seed $S > $O/soak_train_seed$S.txt 2>&1
  tail -n 1 $O/soak_train_seed$S.txt
done
