#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM traffic counters for the same command.
# Outputs under gpurun_out/; the summaries are copied to profiles/ by hand.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 500 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-fixed > $R/gpurun_out/prof_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-fixed > $R/gpurun_out/prof_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-fixed > $R/gpurun_out/prof_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/prof_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-fixed > $R/gpurun_out/prof_sq.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/prof_stats > gpurun_out/prof_stats_summary.txt
python - <<'PY' > gpurun_out/prof_pmc_summary.txt
import csv, glob, collections
for d in ("prof_fetch", "prof_write", "prof_sq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for k, cs in sorted(acc.items()):
        if not k.startswith(("void nnf_", "nnf_")): continue
        print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in cs.items()})
PY
rm -f gpurun_out/prof_*/*/*.db
head -14 gpurun_out/prof_stats_summary.txt; cat gpurun_out/prof_pmc_summary.txt | cut -c1-260
