#!/bin/bash
# HBM traffic of the stand-alone cost kernels (FETCH_SIZE / WRITE_SIZE in separate counter passes; 2*FETCH + WRITE KiB per launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/cost_traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/$C -- python3 $R/tools/probes/cost_only.py $1 > $O/$C.log 2>&1 || { tail -3 $O/$C.log; exit 1; }
done
cd $R
python - "$O" <<'PY'
import csv, glob, collections, sys, os
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(fs[-1])):
        if "nnf_cost_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:44] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    f, w = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
    print(f"{k}: launches {len(cs['FETCH_SIZE'])}  FETCH_SIZE {f:.0f} KiB (x2)  WRITE_SIZE {w:.0f} KiB  ->  {(2 * f + w) * 1024 / 1e6:.1f} MB per launch")
PY
