#!/bin/bash
# kernel trace of weight-space fits of the cfg3 Dot model; usage on the GPU box: bash tools/ws_fit_trace.sh
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/wsfit
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/wsfit.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch, bench
from gpr_calculator_amd.parallel import Comm
gp = bench.build_model("Dot", 2000, Comm())
gp.weight_space_fit = True
for _ in range(3): gp.fit(show=False, opt=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): gp.fit(show=False, opt=False)
torch.cuda.synchronize(); print("ws fit ms", (time.perf_counter() - t0) / 5 * 1e3)
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 /tmp/wsfit.py > $OUT/out.log 2>&1
cd $R
grep "ws fit" $OUT/out.log
python3 - $OUT <<'PY'
import csv, glob, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]))))
# last fit: kernels after the last big gap
last = len(rows) - 1
start = last
while start > 0 and rows[start][0] - rows[start - 1][1] < 300000: start -= 1
seg = rows[start:]
print(len(seg), "kernels, span us", (seg[-1][1] - seg[0][0]) / 1e3, "busy", sum(b - a for a, b, _ in seg) / 1e3)
import collections
agg = collections.defaultdict(lambda: [0, 0])
for a, b, n in seg: agg[n[:90]][0] += b - a; agg[n[:90]][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]: print(f"{t / 1e3:9.1f} us x{c:4d}  {n}")
PY
