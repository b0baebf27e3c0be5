#!/bin/bash
# PMC passes (separate runs, no tracing domains mixed in) for the force-force block kernel.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_${1:-r01}
mkdir -p $OUT
REPO=$PWD
cd /tmp
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python3 $REPO/tools/kff_only.py 2 dot > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $REPO
for f in $(find $OUT -name "*counter_collection.csv"); do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    if "kblock" in k:
        for c, val in v.items():
            print(f"{k:50s} {c:28s} total={val:.4e} per_launch={val/cnt[(k,c)]:.4e} launches={cnt[(k,c)]}")
PY
done
