"""Per-(kernel, grid size) duration summary of a rocprofv3 kernel trace CSV: the plain --stats table
averages a kernel over launches of every shape (fit and predict, one structure and a batch); this
one keeps the shapes apart so the bench's timed-step launches can be read off directly.
usage: python tools/trace_by_grid.py <kernel_trace.csv> [out.csv]"""
import collections
import csv
import sys


def main():
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        agg[(r["Kernel_Name"], grid, int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(((sum(v), k, v) for k, v in agg.items()), reverse=True)
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(("Name", "GridThreads", "WorkgroupX", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs"))
    for tot, (name, grid, wg), v in rows:
        out.writerow((name[:140], grid, wg, len(v), tot, round(tot / len(v), 1), min(v), max(v)))


if __name__ == "__main__":
    main()
