#!/bin/bash
# build container: profiles/r04 on the tree as it is now -- the campaign on one GPU box, collected; then the bench lines that quote a
# counter summary, printed against the summaries just taken (so that their traffic_profile_current is true).  ~12 GPU-minutes.
set -e
cd "$(dirname "$0")/../.."
/usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/dev/r4_final.sh' > /tmp/r4_final.log 2>&1
bash tools/dev/r4_collect.sh | tail -11
/usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/dev/r4_lines.sh' 2>&1 | grep -v "amdgpu\|gpurun" | tail -5
cp gpurun_out/r4lines2/*.json profiles/r04/
