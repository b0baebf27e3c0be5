#!/bin/bash
# the on-the-fly refit loop (cfg5 in shape) under the stage timers and under a rocprofv3 kernel trace
# usage (GPU box, repo root): bash tools/prof_cfg5.sh <tag>
set -e
TAG=${1:-r04}
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG
mkdir -p $O
python3 tools/cfg5_refit_profile.py --cprofile > $O/${TAG}_cfg5_refit_stages.log 2>&1
P=$PWD/gpurun_out/prof_${TAG}cfg5
mkdir -p $P
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o cfg5 -- python3 $OLDPWD/tools/cfg5_loop.py > $O/${TAG}_cfg5_loop_under_rocprof.log 2>&1
cd $OLDPWD
find $P -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_cfg5_refit_kernel_stats.csv
python3 tools/trace_by_grid.py $(find $P -name "*kernel_trace.csv" | head -1) $O/${TAG}_cfg5_refit_kernel_stats_by_grid.csv
rm -rf $P
