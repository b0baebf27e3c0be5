#!/bin/bash
# rocprofv3 kernel-trace summary of an arbitrary python tool; output under gpurun_out/prof_<tag>/
# usage (GPU box, repo root): bash tools/prof_cmd.sh <tag> tools/<script>.py [args]
set -e
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $SCRIPT "$@" > $OUT/stdout.log 2>&1 || { tail -20 $OUT/stdout.log; exit 1; }
cd $ROOT
find $OUT -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv || true
head -25 $OUT/kernel_stats.csv
