#!/usr/bin/env python3
"""Timeline of one steady-state bench iteration from a rocprofv3 --kernel-trace CSV: start / end (us, relative), duration,
stream/queue and the idle gap since the previous kernel ended (all queues merged).
    python tools/timeline.py <dir with *kernel_trace.csv> [anchor kernel substring] [iteration index from the end]"""
import csv
import glob
import sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "xht_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
i0, i1 = starts[-back - 1], starts[-back]
t0 = rows[i0][0]
last_end = t0
print(f"iteration = {(rows[i1][0] - t0) / 1e3:.1f} us ({i1 - i0} kernels)")
busy = 0
for s, e, name, q in rows[i0:i1]:
    gap = (s - last_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3} gap {gap:7.1f}  {name[:70]}")
    last_end = max(last_end, e)
