#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/gputests.log
cat $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/dev/dev_profiles.sh
