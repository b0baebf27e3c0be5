#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench run; output under gpurun_out/prof_<tag>/
# usage (on the GPU box, from the repo root): bash tools/prof_bench.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_stdout.log 2>&1
cd $OLDPWD
find $OUT -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv || true
head -30 $OUT/kernel_stats.csv
