#!/bin/bash
# one structure per call through the pair-kernel route under a kernel + memory-copy trace: the device timeline of a call
# usage (GPU box): bash tools/prof_rbf_single.sh <tag>
set -e
TAG=${1:-r04}
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG; mkdir -p $O
P=$PWD/gpurun_out/prof_${TAG}single; rm -rf $P; mkdir -p $P
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $P -o single -- python3 $OLDPWD/tools/rbf_single_profile.py > $O/${TAG}_rbf_single_under_rocprof.log 2>&1
cd $OLDPWD
python3 - $P $O/${TAG}_rbf_single_timeline.txt <<'PY'
import csv, glob, sys
P, out = sys.argv[1], sys.argv[2]
ev = []
for f in glob.glob(f"{P}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(f"{P}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40]))
ev.sort()
# the last complete call: events between the last two so3 launches
idx = [i for i, e in enumerate(ev) if "so3_" in e[2] and "K " in e[2]]
a, b = idx[-2], idx[-1]
# the neighbour search precedes so3: back up to the first event after the previous call's last kernel by taking a window
t0 = ev[a][0]
with open(out, "w") as fh:
    fh.write("one GP.predict_structure call (RBF, cfg2 size): device events from one descriptor launch to the next, us since the first\n")
    for s, e, n in ev[a:b]:
        fh.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}\n")
    busy = sum(e - s for s, e, n in ev[a:b]) / 1e3
    fh.write(f"span {(ev[b][0] - t0) / 1e3:.1f} us, sum of event durations {busy:.1f} us, {b - a} events\n")
print(open(out).read())
PY
rm -rf $P
