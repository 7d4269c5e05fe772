"""Summarise a rocprofv3 --pmc CSV: mean counter value per kernel name (first 60 chars)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if "hals" not in k and "xty" not in k and "xht" not in k and "cost" not in k and "mu_" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f} max={max(v):16.1f}")
