"""Device timeline of the batches-in-flight loop from a rocprofv3 kernel trace (csv): for the last `nsteps` tail launches,
how many kernels run at once, which fraction of the wall time has 0 / 1 / 2+ kernels on the device, and the average
duration of every kernel kind inside that window.   usage: python tools/overlap_timeline.py <kernel_trace.csv> [window_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]) for r in rows]
ev.sort()
t_end = max(e for _, e, _ in ev)
# the timed loop is the densest part: take the window_ms before the 60 % mark of the tail launches
tails = [s for s, e, n in ev if n.startswith("ws_tail") and "combine" not in n]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 10e6
mid = tails[int(len(tails) * 0.5)]
lo, hi = mid - win / 2, mid + win / 2
sel = [(s, e, n) for s, e, n in ev if s >= lo and e <= hi]
pts = []
for s, e, n in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
occ = collections.Counter(); cur = 0; last = lo
for t, d in pts:
    occ[min(cur, 4)] += t - last; last = t; cur += d
occ[min(cur, 4)] += hi - last
tot = sum(occ.values())
print(f"window {win * 1e-6:.1f} ms, {len(sel)} launches, {sum(1 for s, e, n in sel if 'ws_tail_kernel' in n)} steps")
for k in sorted(occ): print(f"  {k}{'+' if k == 4 else ' '} kernels running: {100.0 * occ[k] / tot:5.1f} %")
by = collections.defaultdict(list)
for s, e, n in sel: by[n].append((e - s) * 1e-3)
nst = max(1, sum(1 for s, e, n in sel if n.startswith("ws_tail") and "combine" not in n))
print(f"  step period {win * 1e-3 / nst:.1f} us")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n[:48]:48s} n/step {len(v) / nst:4.1f}  avg {sum(v) / len(v):7.1f} us  busy/step {sum(v) / nst:7.1f} us")
