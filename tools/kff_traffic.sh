#!/bin/bash
# force-force block kernel (tools/kff_only.py, cfg3 predict shape): time and L2 <-> fabric bytes per launch against the block -> XCD
# mapping (GPR_KBLOCK_XCD; 1 is the default) and the packed ranges taken per task (GPR_KBLOCK_MERGE; 1 is the default).  FETCH_SIZE / WRITE_SIZE in separate --pmc passes.
# usage (GPU box): bash tools/kff_traffic.sh <tag> [kind=rbf]
TAG=${1:-r04}; KIND=${2:-rbf}
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG; mkdir -p $O
LOG=$O/${TAG}_kff_traffic.log
REPO=$PWD
echo "kblock_kernel<FF> ($KIND), 2048 x 28 stream rows against 2000 x 28 resident rows, d = 30: algorithmic bytes = operands once + output once" > $LOG
for cfg in "0 1" "1 1" "1 2"; do set -- $cfg; xcd=$1; mg=$2
  export GPR_KBLOCK_XCD=$xcd GPR_KBLOCK_MERGE=$mg
  t=$(python3 tools/kff_only.py 5 $KIND 2>/dev/null | tail -1)
  line="xcd=$xcd merge=$mg: $t"
  for c in FETCH_SIZE WRITE_SIZE; do
    P=$O/_pmc_$c; rm -rf $P
    (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $P -o pmc -- python3 $REPO/tools/kff_only.py 2 $KIND > $P.log 2>&1)
    v=$(python3 - $P <<'PY'
import csv, glob, sys
tot, n = 0.0, 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kblock" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print(f"{tot / max(n, 1):.4e}")
PY
)
    line="$line  $c/launch=$v KB"
    rm -rf $P $P.log
  done
  echo "$line" | tee -a $LOG
done
