#!/bin/bash
# dev (GPU box): where does the fp16x3 kernel wait?  timing with the stage barrier / the stage vmcnt wait removed (results wrong)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
for lib in default nobar novm; do
  if [ $lib = default ]; then unset NERFTEX_LIB; else export NERFTEX_LIB=$R/build_dev/r2/$lib.so; fi
  for wt in glorot zero; do
    echo -n "LIB=$lib " >> $O/wait.log
    timeout 120 python tools/dev/dev_loop.py fp16x3 3 $wt 2>/dev/null | grep LOOP >> $O/wait.log
  done
done
cat $O/wait.log
