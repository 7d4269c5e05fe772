set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for L in mfma lane; do
export NNF_HALS_FORCE=$L
O=$R/gpurun_out/pmc_sweeps_$L
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/a -- python3 $R/tools/probes/sweeps_only.py 50x100000 100x125000 > $O/a.log 2>&1 || { tail -3 $O/a.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d $O/b -- python3 $R/tools/probes/sweeps_only.py 50x100000 100x125000 > $O/b.log 2>&1 || { tail -3 $O/b.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for L in ("mfma", "lane"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_sweeps_{L}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hals" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"] and "sum" not in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print(L, k)
        for c, v in sorted(cs.items()):
            print(f"    {c:30s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
PY
