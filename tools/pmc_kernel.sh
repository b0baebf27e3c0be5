#!/bin/bash
# SQ counters of one kernel of the default bench; usage on the GPU box: bash tools/pmc_kernel.sh <kernel substring> [tag]
set -e
export TMPDIR=/tmp
KERN=${1:-ws_tail}
TAG=${2:-$KERN}
OUT=$PWD/gpurun_out/pmck_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pair-route > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pair-route > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pair-route > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; exit 1; }
cd $REPO
python3 - $OUT $KERN <<'PY'
import csv, sys, glob, collections
out, kern = sys.argv[1], sys.argv[2]
for sub in ("a", "b", "c"):
    fs = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if kern not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in agg: print(f"{k:32s} {agg[k]/cnt[k]:16.0f} per launch ({cnt[k]} launches)")
PY
