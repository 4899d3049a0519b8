#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log
cat $O/gputests.log
rm -rf $R/gpurun_out/r2lines $R/gpurun_out/pf_* $R/gpurun_out/px_*
bash tools/dev/dev_profiles.sh
