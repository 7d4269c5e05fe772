"""Condense rocprofv3 --kernel-trace --stats CSV output into a short per-kernel table (committed under profiles/)."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows[:40]:
    name = r["Name"][:70]
    print(f"{name:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
          f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
