#!/bin/bash
# GPU box: the round's profile set on the FINAL tree, one box: tools/dev/r4_profiles.sh (render lines, headline + instanced-scene passes, the
# instancer alone with counters, scaling table) and tools/dev/r4_train_profiles.sh (the training step).  Collected by tools/dev/r4_collect.sh.
R=$GRAFT_REPO_ROOT; cd $R
bash tools/dev/r4_profiles.sh > gpurun_out/r4_profiles.log 2>&1
bash tools/dev/r4_train_profiles.sh > gpurun_out/r4_train_profiles.log 2>&1
tail -3 gpurun_out/r4_profiles.log; tail -6 gpurun_out/r4_train_profiles.log
