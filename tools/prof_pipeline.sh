#!/bin/bash
# device timeline of the batches-in-flight loop: rocprofv3 kernel trace of a short bench run + tools/overlap_timeline.py
# usage (GPU box, repo root): bash tools/prof_pipeline.sh <tag> [bench args]
set -e
TAG=${1:-pipe}; shift || true
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $ROOT/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-pair-route --sustain 0 "$@" > $OUT/bench_stdout.log 2>&1
cd $ROOT
python tools/overlap_timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 8
