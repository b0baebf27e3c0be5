#!/bin/bash
# VALU / LDS / SALU instruction counts of the descriptor kernel per phase: the kernel ends early with GPR_SO3_DBG = 1..4
# (tables | + c_tot, x, T | + R/S/dP | + row accumulation | 0 = whole kernel); usage (GPU box): bash tools/pmc_so3_phases.sh
# needs the tools build of the library (make -C gpr_calculator_amd/csrc ablation): the production library has no early exits
set -e
export TMPDIR=/tmp
export GPR_HIP_LIB=$PWD/gpr_calculator_amd/libgprhip_ablation.so
OUT=$PWD/gpurun_out/pmc_so3ph
REPO=$PWD
for d in 1 2 3 4 0; do
  mkdir -p $OUT/$d
  cd /tmp
  GPR_SO3_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/$d -o pmc -- python3 $REPO/tools/so3_only.py > $OUT/$d.log 2>&1 || { tail -5 $OUT/$d.log; exit 1; }
  cd $REPO
  python3 - $OUT/$d $d <<'PY'
import csv, sys, glob, collections
f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
for kern in ("so3_fixed", "so3_centre"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"] or int(r["Grid_Size"]) < 100000: continue        # the batch launches only
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    if not cnt: continue
    w = agg["SQ_WAVES"] / max(cnt["SQ_WAVES"], 1)
    print(f"dbg={sys.argv[2]} {kern:10s}: " + "  ".join(f"{k[3:]}={agg[k]/cnt[k]/max(w,1):.0f}/wave" for k in sorted(agg) if k != "SQ_WAVES") + f"  waves={w:.0f}")
PY
done
rm -rf $OUT
