"""Device timeline of one Cholesky (gpr_potrf_lower) from a rocprofv3 kernel trace of tools/potrf_check.py <n>:
per 256-column step the diagonal-block, panel and update launches with their start offsets, durations and queues.
usage: python tools/potrf_timeline.py <kernel_trace.csv> [first_step last_step]"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
               for r in csv.DictReader(open(sys.argv[1]))))
short = lambda n: ("diag" if "potrf_diag256" in n else "panel" if "potrf_panel256" in n else "gemm128" if "gemm128" in n else
                   "gemm64" if "gemm_kernel" in n else "bulk" if "syrk_bulk" in n else "syrk" if "syrk_narrow" in n else "triinv" if "tri_inv64" in n else None)
# the last complete factorisation: from the last diag launch with the largest step count backwards
diags = [i for i, r in enumerate(rows) if short(r[2]) == "diag"]
ends = [i for i, r in enumerate(rows) if short(r[2]) == "triinv"]
segs = [[r for r in rows[(ends[i - 1] if i else -1) + 1:ends[i] + 1] if short(r[2])] for i in range(len(ends))]
big = max(len(s) for s in segs)
seg = [s for s in segs if len(s) == big][-1]                     # the last of the largest factorisations
t0 = seg[0][0]
print(f"factorisation: {len(seg)} launches, span {(seg[-1][1] - t0) / 1e3:.1f} us")
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 3)
step = -1
busy = {}
for a, b, n, q, wg in seg:
    k = short(n)
    busy[k] = busy.get(k, 0) + (b - a)
    if k == "diag":
        step += 1
    if lo <= step <= hi:
        print(f"  step {step:2d} {k:8s} q{q:>3s} wgs {wg:6d}  start {(a - t0) / 1e3:9.1f}  dur {(b - a) / 1e3:7.1f} us")
print({k: round(v / 1e3, 1) for k, v in busy.items()})
# the dependent chain: diag_k start to diag_{k+1} start
ds = [r for r in seg if short(r[2]) == "diag"]
gaps = [(ds[i + 1][0] - ds[i][0]) / 1e3 for i in range(len(ds) - 1)]
print("diag-to-diag us:", " ".join(f"{g:.0f}" for g in gaps))
print("diag durations us:", " ".join(f"{(b - a) / 1e3:.0f}" for a, b, *_ in ds))
