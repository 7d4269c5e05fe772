#!/bin/bash
# One rocprofv3 counter pass over a short bench run (counters only: never combined with tracing), summarised per kernel.
#   bash tools/pmc_pass.sh CONFIG TAG COUNTER [COUNTER ...]        -> gpurun_out/pmc_<CONFIG>_<TAG>.txt
set -o pipefail
CFG=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_${CFG}_${TAG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/raw -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu --no-fixed --no-extra --no-kernels > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
python - "$O" <<'PY' > $O.txt
import csv, glob, collections, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not k.startswith(("void nnf_", "nnf_")): continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
PY
rm -rf $O/raw
cat $O.txt | head -${HEAD:-60}
