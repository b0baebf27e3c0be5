#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_potrf.sh <tag> <n> [first_step last_step]
set -e
TAG=$1; N=$2
bash tools/prof_cmd.sh $TAG tools/potrf_check.py $N > /dev/null
python tools/potrf_timeline.py $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1) ${3:-0} ${4:-2}
