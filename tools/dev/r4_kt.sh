#!/bin/bash
# GPU box: kernel trace of bench_instancer with the given args -> gpurun_out/r4kt/<tag>_kernel_stats.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4kt; mkdir -p $O; cd $R
TAG=$1; shift
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$TAG -o kt -- python tools/bench_instancer.py --no-render --steps 10 "$@" > $O/$TAG.log 2>&1
cp $O/kt_$TAG/*kernel_stats.csv $O/${TAG}_kernel_stats.csv
grep ntx_inst $O/${TAG}_kernel_stats.csv | cut -d, -f1-6 | cut -c1-160
grep "^{" $O/$TAG.log | cut -c1-200
