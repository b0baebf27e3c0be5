#!/bin/bash
# second half of a round's collection (the first, tools/collect_round.sh, fills a gpurun call on its own): the pipeline depth
# sweep, the stream -> hardware-queue maps, the single-structure RBF call (timeline, host profile, few-column variance kernel),
# the force-force block's traffic against its block -> XCD mapping.
# usage (GPU box, repo root): bash tools/collect_round_b.sh r04
R=${1:-r04}
O=gpurun_out/$R
mkdir -p $O
bash tools/depth_sweep.sh $R > /dev/null 2>&1
echo "depth sweep done"; tail -3 $O/${R}_depth_sweep.log
python tools/stream_queues.py 10 > $O/${R}_stream_queues.log 2>&1
GPU_MAX_HW_QUEUES=8 python tools/stream_queues.py 10 >> $O/${R}_stream_queues.log 2>&1
for d in 2 3 4 6; do python tools/session_queues.py $d 2>&1 | tail -$((d + 1)); done > $O/${R}_session_queues.log
python tools/variance_few.py > $O/${R}_variance_few.log 2>&1
python tools/rbf_single_profile.py > $O/${R}_rbf_single_profile.log 2>&1
bash tools/prof_rbf_single.sh $R > /dev/null 2>&1
echo "single-structure call done"; head -3 $O/${R}_rbf_single_profile.log | cut -c1-300
bash tools/kff_traffic.sh $R rbf > /dev/null 2>&1
echo "kff traffic done"; cat $O/${R}_kff_traffic.log
python tools/gemm_k.py > $O/${R}_gemm_k.log 2>&1
python tools/potrf_check.py 8300 16500 24250 > $O/${R}_potrf_check_large.log 2>&1
python tools/session_stress.py 20000 4 > $O/${R}_session_stress.log 2>&1
echo "all done"
