#!/bin/bash
# HBM traffic counters (separate --pmc passes, no tracing domains) of a short bench run.
# usage (GPU box, repo root): bash tools/pmc_bench.sh <tag> [bench args]
set -e
export TMPDIR=/tmp
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/pmcb_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
done
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, collections, glob, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        g = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
        k = (r["Kernel_Name"], int(g))
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg: res[k][c] = (agg[k], cnt[k])
rows = []
for k, v in res.items():
    f, nf = v.get("FETCH_SIZE", (0, 1)); w, nw = v.get("WRITE_SIZE", (0, 1))
    rows.append((f + w, k, f / max(nf, 1), w / max(nw, 1), nf))
rows.sort(reverse=True)
with open(f"{out}/traffic_summary.csv", "w") as fh:
    fh.write("kernel,grid_threads,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
    for _, (k, g), f, w, n in rows[:40]:
        fh.write(f"\"{k[:120]}\",{g},{n},{f:.1f},{w:.1f}\n")
        print(f"{k[:60]:60s} grid={g:9d} n={n:4d} fetch/launch={f:12.1f} KB write/launch={w:12.1f} KB")
PY
