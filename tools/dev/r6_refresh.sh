#!/bin/bash
# build container: profiles/r06 on the tree as it is now (commit first: the summaries record HEAD) -- the campaign on one GPU box, collected;
# then every bench line printed against the summaries just taken, so that each traffic_profile_current is true.  ~25 GPU-minutes.
set -e
cd "$(dirname "$0")/../.."
H=$(git rev-parse --short=12 HEAD)
/usr/local/graft/bin/gpurun --timeout 3000 -- "NTX_PROFILE_HEAD=$H bash tools/dev/r6_final.sh" > /tmp/r6_final.log 2>&1 || true
tail -12 /tmp/r6_final.log
NTX_PROFILE_HEAD=$H bash tools/dev/r6_collect.sh | tail -3
/usr/local/graft/bin/gpurun --timeout 1500 -- "NTX_PROFILE_HEAD=$H bash tools/dev/r6_lines.sh" 2>&1 | grep -v "amdgpu\|gpurun\] merged" | tail -14
cp gpurun_out/r6lines/*.json profiles/r06/
