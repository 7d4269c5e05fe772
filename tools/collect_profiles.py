"""Copy the summaries written by tools/bench_profile.sh (under gpurun_out/) into profiles/ and recompute the HBM traffic
of the dominant kernel from the PMC passes (FETCH_SIZE is doubled: the gfx950 correction of MI355X_MICROARCH.md)."""
import csv, glob, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
shutil.copy(os.path.join(G, "prof_stats_summary.txt"), os.path.join(P, f"{tag}_bench_kernel_stats.txt"))
ks = sorted(glob.glob(os.path.join(G, "prof_stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "prof_pmc_summary.txt"), os.path.join(P, f"{tag}_bench_pmc_fetch_write_sq.txt"))
shutil.copy(os.path.join(G, "bench.json"), os.path.join(P, f"{tag}_bench_line.json"))
acc = collections.defaultdict(list)
for d in ("prof_fetch", "prof_write"):
    for f in glob.glob(os.path.join(G, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "nnf_xty_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
fetch = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
write = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
out = {"kernel": "nnf_xty_kernel<3,2,true>", "fetch_size_kib_raw": fetch, "write_size_kib": write,
       "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
       "note": "separate --pmc passes of `python bench.py --steps 3 --warmup 1 --no-cpu` (tools/bench_profile.sh); "
               f"FETCH_SIZE doubled per the gfx950 correction; mean over {len(acc['FETCH_SIZE'])} launches",
       "source": f"profiles/{tag}_bench_pmc_fetch_write_sq.txt"}
json.dump(out, open(os.path.join(P, f"{tag}_xty_traffic.json"), "w"), indent=1)
print(out)
