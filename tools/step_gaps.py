"""Device timeline of the last prediction steps in a rocprofv3 kernel trace of bench.py: per step (from the geometry
upload kernel to the combine launch of the tail) the span on the device, the summed kernel durations and the idle time
between kernels.  usage: python tools/step_gaps.py <kernel_trace.csv> [nsteps]"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ends = [i for i, r in enumerate(rows) if "ws_tail_combine_kernel" in r[2]]
for e in ends[-nsteps:]:
    s = e
    while s > 0 and "nbr_upload_kernel" not in rows[s][2]:
        s -= 1
    seg = rows[s:e + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(b - a for a, b, _ in seg)
    print(f"step: {len(seg)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
    if e == ends[-1]:
        prev = None
        for a, b, n in seg:
            gap = 0 if prev is None else a - prev
            print(f"   +{gap / 1e3:6.1f} us gap  {(b - a) / 1e3:7.1f} us  {n[:70]}")
            prev = b
