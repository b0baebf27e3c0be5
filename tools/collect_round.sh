#!/bin/bash
# everything profiles/ holds for one round, in one GPU-box call: bench lines, rocprofv3 kernel-trace summaries (+ the split
# by launch shape), device timelines, PMC traffic passes, per-kernel micro-benchmarks, the two-rank rehearsal.
# usage (GPU box, repo root): bash tools/collect_round.sh r04
set -e
R=${1:-r04}
O=gpurun_out/$R
mkdir -p $O
python bench.py > $O/${R}_dot_bench.json 2> $O/${R}_dot_bench.err
echo "bench dot done"; tail -c 300 $O/${R}_dot_bench.json
python bench.py --depth 1 --no-cpu-baseline --no-pair-route > $O/${R}_dot_bench_sync.json 2> $O/${R}_dot_bench_sync.err
python bench.py --kernel RBF --steps 20 --warmup 3 > $O/${R}_rbf_bench.json 2> $O/${R}_rbf_bench.err
echo "bench rbf done"
python bench.py --mode fit --steps 10 --warmup 2 --no-cpu-baseline > $O/${R}_fit_dot_bench.json 2> $O/${R}_fit_dot.err
python bench.py --mode fit --kernel RBF --steps 5 --warmup 1 --no-cpu-baseline > $O/${R}_fit_rbf_bench.json 2> $O/${R}_fit_rbf.err
python bench.py --n-train 8000 --steps 50 --warmup 5 --no-cpu-baseline --no-pair-route > $O/${R}_cfg4_dot_bench.json 2> $O/${R}_cfg4_dot.err
python bench.py --n-train 500 --batch 1 --steps 100 --warmup 5 --no-cpu-baseline --no-pair-route > $O/${R}_cfg2_dot_bench.json 2> $O/${R}_cfg2_dot.err
# BASELINE.json configs[1] as it is stated: RBF, N_train = 500, single-shot fit() + predict() (batch 1, synchronous)
python bench.py --kernel RBF --n-train 500 --batch 1 --depth 1 --steps 50 --warmup 5 > $O/${R}_cfg2_rbf_bench.json 2> $O/${R}_cfg2_rbf.err
python bench.py --kernel RBF --n-train 500 --mode fit --steps 10 --warmup 2 --no-cpu-baseline > $O/${R}_cfg2_fit_rbf_bench.json 2> $O/${R}_cfg2_fit_rbf.err
echo "fit / cfg2 / cfg4 done"
python tools/potrf_check.py 1516 6063 24250 > $O/${R}_potrf_check.log 2>&1
python tools/trsv_check.py > $O/${R}_trsv_check.log 2>&1 || true
python tools/tail_bench.py > $O/${R}_tail_bench.log 2>&1
GPR_TAIL_FORM=1 python tools/tail_bench.py >> $O/${R}_tail_bench.log 2>&1
GPR_TAIL_FORM=1 GPR_TAIL_NO_GLDS=1 python tools/tail_bench.py >> $O/${R}_tail_bench.log 2>&1
python tools/nbr_bench.py > $O/${R}_nbr_bench.log 2>&1
GPR_NBR_NO_FUSED=1 python tools/nbr_bench.py >> $O/${R}_nbr_bench.log 2>&1
python tools/pipeline_split.py 3 > $O/${R}_pipeline_split.log 2>&1
python tools/single_latency.py 1 500 > $O/${R}_single_latency.log 2>&1
python tools/single_latency.py 8 >> $O/${R}_single_latency.log 2>&1
python tools/cfg5_loop.py > $O/${R}_cfg5_loop.log 2>&1
python tools/cfg5_neb.py 50 2 > $O/${R}_cfg5_neb.log 2>&1
bash tools/prof_cfg5.sh ${R}cfg5 > /dev/null 2>&1 && cp gpurun_out/${R}cfg5/${R}cfg5_cfg5_refit_stages.log $O/${R}_cfg5_refit_stages.log && cp gpurun_out/${R}cfg5/${R}cfg5_cfg5_refit_kernel_stats.csv $O/${R}_cfg5_refit_kernel_stats.csv && cp gpurun_out/${R}cfg5/${R}cfg5_cfg5_refit_kernel_stats_by_grid.csv $O/${R}_cfg5_refit_kernel_stats_by_grid.csv
# the band under a kernel trace: what a refit is made of once the training set has a few thousand rows
( export TMPDIR=/tmp; P=$PWD/gpurun_out/prof_${R}neb; mkdir -p $P; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $P -o neb -- python3 $OLDPWD/tools/cfg5_neb.py 25 1 > $OLDPWD/$O/${R}_cfg5_neb_under_rocprof.log 2>&1; cd $OLDPWD; find $P -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $O/${R}_cfg5_neb_kernel_stats.csv; rm -rf $P ) || true
python tools/so3_only.py > $O/${R}_so3_only.log 2>&1
python tools/so3_scan.py > $O/${R}_so3_scan.log 2>&1
bash tools/so3_phases.sh > $O/${R}_so3_phases.log 2>&1 || true
bash tools/pmc_so3_phases.sh > $O/${R}_so3_pmc_phases.log 2>&1 || true
bash tools/pmc_so3_wait.sh > $O/${R}_so3_pmc_wait.log 2>&1 || true
python tools/fit_opt_time.py > $O/${R}_fit_opt_time.log 2>&1 || true
echo "micro done"
bash tools/prof_bench.sh ${R}dot > /dev/null
bash tools/prof_bench.sh ${R}rbf --kernel RBF > /dev/null
for k in dot rbf; do
  P=gpurun_out/prof_${R}$k
  cp $P/kernel_stats.csv $O/${R}_${k}_kernel_stats.csv
  cp $P/bench_stdout.log $O/${R}_${k}_bench_under_rocprof.log
  python tools/trace_by_grid.py $(find $P -name "*kernel_trace.csv" | head -1) $O/${R}_${k}_kernel_stats_by_grid.csv
done
bash tools/prof_pipeline.sh ${R}pipe --depth 4 > $O/${R}_overlap_depth4.txt 2>&1
bash tools/prof_pipeline.sh ${R}sync --depth 1 > $O/${R}_overlap_sync.txt 2>&1
bash tools/prof_potrf.sh ${R}pt 6063 0 3 > $O/${R}_potrf_timeline.txt 2>&1
echo "traces done"
bash tools/prof_bench.sh ${R}fit --mode fit > /dev/null
cp gpurun_out/prof_${R}fit/kernel_stats.csv $O/${R}_fit_dot_kernel_stats.csv
bash tools/pmc_bench.sh ${R}dot > $O/${R}_dot_pmc.log 2>&1
bash tools/pmc_bench.sh ${R}rbf --kernel RBF > $O/${R}_rbf_pmc.log 2>&1
cp gpurun_out/pmcb_${R}dot/traffic_summary.csv $O/${R}_dot_pmc_traffic_summary.csv
cp gpurun_out/pmcb_${R}rbf/traffic_summary.csv $O/${R}_rbf_pmc_traffic_summary.csv
rm -rf gpurun_out/prof_${R}dot gpurun_out/prof_${R}rbf gpurun_out/prof_${R}fit gpurun_out/prof_${R}pipe gpurun_out/prof_${R}sync gpurun_out/prof_${R}pt gpurun_out/pmcb_${R}dot/FETCH_SIZE gpurun_out/pmcb_${R}dot/WRITE_SIZE gpurun_out/pmcb_${R}rbf/FETCH_SIZE gpurun_out/pmcb_${R}rbf/WRITE_SIZE
echo "pmc done"
export GPR_DIST_BACKEND=gloo
python bench.py --gpus 2 --steps 50 --warmup 5 --no-pair-route --no-cpu-baseline > $O/${R}_world2_gloo_predict_dot.json 2> $O/${R}_world2_predict.err
python bench.py --gpus 2 --mode fit --steps 5 --warmup 1 > $O/${R}_world2_gloo_fit_dot.json 2> $O/${R}_world2_fit_dot.err
python bench.py --gpus 2 --mode fit --kernel RBF --steps 3 --warmup 1 > $O/${R}_world2_gloo_fit_rbf.json 2> $O/${R}_world2_fit_rbf.err
GPR_DIST_CHOL_MIN_N=1024 python bench.py --gpus 2 --mode fit --kernel RBF --steps 3 --warmup 1 > $O/${R}_world2_gloo_fit_rbf_cyclic.json 2> $O/${R}_world2_fit_rbf_cyclic.err
GPR_DIST_LOOKAHEAD=0 GPR_DIST_CHOL_MIN_N=1024 python bench.py --gpus 2 --mode fit --kernel RBF --steps 3 --warmup 1 > $O/${R}_world2_gloo_fit_rbf_cyclic_one_lane.json 2> $O/${R}_world2_fit_rbf_cyclic_one_lane.err
echo "all done"
