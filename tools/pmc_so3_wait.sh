#!/bin/bash
# where the wavefronts of the descriptor kernel spend their cycles: parked (s_waitcnt / barrier), issue-stalled, issuing; LDS
# and memory pipe activity; per wavefront, batch launches only.  usage (GPU box): bash tools/pmc_so3_wait.sh
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_so3w
REPO=$PWD
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -o pmc -- python3 $REPO/tools/so3_only.py > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/b -o pmc -- python3 $REPO/tools/so3_only.py > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for kern in ("so3_fixed", "so3_centre"):
    for sub in ("a", "b"):
        fs = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(fs[0])):
            if kern not in r["Kernel_Name"] or int(r["Grid_Size"]) < 100000: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        if not cnt: continue
        w = agg["SQ_WAVES"] / max(cnt["SQ_WAVES"], 1)
        print(f"{kern:10s} [{sub}] " + "  ".join(f"{k[3:]}={agg[k]/cnt[k]/max(w,1):.0f}" for k in sorted(agg) if k != "SQ_WAVES") + f"  (per wavefront; waves={w:.0f})")
PY
rm -rf $OUT
