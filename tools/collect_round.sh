#!/bin/bash
# everything profiles/ holds for one round, in one GPU-box call: bench lines, rocprofv3 kernel-trace summaries (+ the split
# by launch shape), PMC traffic passes.   usage (GPU box, repo root): bash tools/collect_round.sh r02
set -e
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
python bench.py > $O/${R}_dot_bench.json 2> $O/${R}_dot_bench.err
echo "bench dot done"; tail -c 400 $O/${R}_dot_bench.json
python bench.py --kernel RBF --steps 20 --warmup 3 > $O/${R}_rbf_bench.json 2> $O/${R}_rbf_bench.err
echo "bench rbf done"
python bench.py --mode fit --steps 10 --warmup 2 --no-cpu-baseline > $O/${R}_fit_dot_bench.json 2> $O/${R}_fit_dot.err
python bench.py --mode fit --kernel RBF --steps 5 --warmup 1 --no-cpu-baseline > $O/${R}_fit_rbf_bench.json 2> $O/${R}_fit_rbf.err
echo "fit done"
python tools/potrf_check.py 1516 6063 > $O/${R}_potrf_check.log 2>&1
bash tools/prof_bench.sh ${R}dot > /dev/null
bash tools/prof_bench.sh ${R}rbf --kernel RBF > /dev/null
for k in dot rbf; do
  P=gpurun_out/prof_${R}$k
  cp $P/kernel_stats.csv $O/${R}_${k}_kernel_stats.csv
  cp $P/bench_stdout.log $O/${R}_${k}_bench_under_rocprof.log
  python tools/trace_by_grid.py $(find $P -name "*kernel_trace.csv" | head -1) $O/${R}_${k}_kernel_stats_by_grid.csv
done
echo "traces done"
bash tools/prof_bench.sh ${R}fit --mode fit > /dev/null
cp gpurun_out/prof_${R}fit/kernel_stats.csv $O/${R}_fit_dot_kernel_stats.csv
python tools/trace_by_grid.py $(find gpurun_out/prof_${R}fit -name "*kernel_trace.csv" | head -1) $O/${R}_fit_dot_kernel_stats_by_grid.csv
bash tools/pmc_bench.sh ${R}dot > $O/${R}_dot_pmc.log 2>&1
bash tools/pmc_bench.sh ${R}rbf --kernel RBF > $O/${R}_rbf_pmc.log 2>&1
cp gpurun_out/pmcb_${R}dot/traffic_summary.csv $O/${R}_dot_pmc_traffic_summary.csv
cp gpurun_out/pmcb_${R}rbf/traffic_summary.csv $O/${R}_rbf_pmc_traffic_summary.csv
rm -rf gpurun_out/prof_${R}dot gpurun_out/prof_${R}rbf gpurun_out/prof_${R}fit gpurun_out/pmcb_${R}dot/FETCH_SIZE gpurun_out/pmcb_${R}dot/WRITE_SIZE gpurun_out/pmcb_${R}rbf/FETCH_SIZE gpurun_out/pmcb_${R}rbf/WRITE_SIZE
echo "all done"
