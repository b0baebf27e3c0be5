#!/bin/bash
# SQ counters of the descriptor kernel (tools/so3_only.py); usage on the GPU box: bash tools/pmc_so3.sh <tag>
set -e
export TMPDIR=/tmp
TAG=${1:-so3}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o pmc -- python3 $REPO/tools/so3_only.py > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/b -o pmc -- python3 $REPO/tools/so3_only.py > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for sub in ("a", "b"):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if "so3_centre" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in agg: print(f"{k:24s} {agg[k]/cnt[k]:16.0f} per launch ({cnt[k]} launches)")
PY
