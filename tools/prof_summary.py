"""Condense rocprofv3 --kernel-trace --stats CSV output into a short per-kernel table (committed under profiles/).

    python tools/prof_summary.py <dir with the rocprofv3 output>

Only the NEWEST kernel_stats / kernel_trace pair under the directory counts (gpurun merges the files of earlier calls into the
same place).  Kernels launched with more than one grid -- W^T X on the whole matrix and on the 16 row blocks of the once-per-run
rounding calibration (Engine.cross_rounding), sweeps on full and last column blocks -- are listed again per grid: the mean of
the mixed launches says nothing about either."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


rows = []
f = newest("*kernel_stats.csv")
if f:
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows[:40]:
    name = r["Name"][:70]
    print(f"{name:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
          f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
t = newest("*kernel_trace.csv")
if t:
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(t) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if "nnf_" not in n:
                continue
            g = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            by[n][g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    multi = {n: gs for n, gs in by.items() if len(gs) > 1}
    if multi:
        print("\n== by grid (threads x, y, z) -- kernels launched with more than one grid ==")
        print(f"{'kernel':70s} {'grid':>18s} {'calls':>7s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s}")
        for n, gs in sorted(multi.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            for g, v in sorted(gs.items(), key=lambda kv: -sum(kv[1])):
                print(f"{n[:70]:70s} {'x'.join(map(str, g)):>18s} {len(v):7d} {sum(v)/len(v):10.2f} {min(v):9.2f} {max(v):9.2f}")
