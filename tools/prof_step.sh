#!/bin/bash
# device timeline of the headline step: rocprofv3 kernel trace of a short bench run + tools/step_gaps.py
# usage (GPU box, repo root): bash tools/prof_step.sh <tag>
set -e
TAG=${1:-step}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pair-route > $OUT/bench_stdout.log 2>&1
cd $ROOT
python tools/step_gaps.py $(find $OUT -name "*kernel_trace.csv" | head -1) 3
