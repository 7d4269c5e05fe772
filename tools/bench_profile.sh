#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM traffic / SQ counters for the same command.
#   bash tools/bench_profile.sh [CONFIG=B] [stats|all]
# Outputs under gpurun_out/prof_<CONFIG>/; tools/collect_profiles.py copies the summaries to profiles/.
# Counters are collected in their own runs (never combined with tracing); the program itself follows `--`.
set -o pipefail
CFG=${1:-B}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$CFG
mkdir -p $O
cd $R && timeout -k 10 500 python bench.py --config $CFG --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
# (the loop of the bench line itself: same --steps / --warmup; the legs that follow the timed region are switched off)
ARGS="--config $CFG --steps 20 --warmup 3 --no-cpu --no-fixed --no-extra --no-kernels"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1 || exit 1
if [ "$WHAT" = "all" ]; then
ARGS="--config $CFG --steps 3 --warmup 1 --no-cpu --no-fixed --no-extra --no-kernels"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM --output-format csv -d $O/sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1 || exit 1
fi
cd $R
python tools/prof_summary.py $O/stats > $O/stats_summary.txt
python - "$O" <<'PY' > $O/pmc_summary.txt
import csv, glob, collections, sys
O = sys.argv[1]
for d in ("fetch", "write", "sq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k, cs in sorted(acc.items()):
        if not k.startswith(("void nnf_", "nnf_")): continue
        print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in cs.items()})
PY
rm -f $O/*/*/*.db
head -16 $O/stats_summary.txt
