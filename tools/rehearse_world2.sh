#!/bin/bash
# two ranks on the one GPU of the box over gloo (RCCL refuses two ranks per device): the N > 1 code paths of bench.py,
# predict (weak scaling) and fit (strong scaling, incl. the block-cyclic Cholesky with its threshold lowered)
# usage (GPU box, repo root): bash tools/rehearse_world2.sh <outdir>
set -e
O=${1:-gpurun_out/world2}
mkdir -p $O
export GPR_DIST_BACKEND=gloo
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 "${@:2}"; }
run 29511 --steps 20 --warmup 3 --no-cpu-baseline > $O/predict_dot.json 2> $O/predict_dot.err
run 29512 --mode fit --steps 5 --warmup 1 > $O/fit_dot.json 2> $O/fit_dot.err
run 29513 --mode fit --kernel RBF --steps 3 --warmup 1 > $O/fit_rbf.json 2> $O/fit_rbf.err
GPR_DIST_CHOL_MIN_N=1024 run 29514 --mode fit --kernel RBF --steps 3 --warmup 1 > $O/fit_rbf_cyclic.json 2> $O/fit_rbf_cyclic.err
for f in predict_dot fit_dot fit_rbf fit_rbf_cyclic; do python - $O/$f.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], j["n_gpus"], round(j["value"], 3), j["unit"], j["config"].get("parallelism"), j["config"].get("fit_ms"))
PY
done
