#!/bin/bash
# the headline step against the number of batches in flight: steady state (600 steps) and the driver's short region (20 steps)
# usage (GPU box): bash tools/depth_sweep.sh <tag>  ->  gpurun_out/<tag>/<tag>_depth_sweep.log
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
LOG=$O/${TAG}_depth_sweep.log
echo "bench.py --no-pair-route --no-cpu-baseline --depth D: ms per step (structures/s); kernel HIP-event ms per step inside the timed region" > $LOG
for steps in 600 20; do
  for d in 1 2 3 4 5 6; do
    for rep in 1 2; do
      timeout -k 10 200 python3 bench.py --gpus 1 --steps $steps --warmup $((steps == 20 ? 5 : 20)) --no-pair-route --no-cpu-baseline --sustain 0.2 --depth $d > $O/_sweep.json 2> $O/_sweep.err || { echo "depth $d failed" >> $LOG; continue; }
      python3 - $O/_sweep.json $steps $d >> $LOG <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"steps {int(sys.argv[2]):4d} depth {sys.argv[3]}: {d['ms_per_step']:.4f} ms  ({d['value']:.0f} structures/s)  kernels {d['roofline']['kernel_ms_per_step']}")
PY
    done
  done
done
rm -f $O/_sweep.json $O/_sweep.err
cat $LOG
