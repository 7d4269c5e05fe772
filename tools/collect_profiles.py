"""Copy the summaries written by tools/bench_profile.sh (gpurun_out/prof_<CONFIG>/) into profiles/ and recompute the HBM
traffic of each configuration's main kernels from the PMC passes (FETCH_SIZE doubled: the gfx950 correction of
MI355X_MICROARCH.md; WRITE_SIZE as reported; both in KiB).

    python tools/collect_profiles.py r02 B C D
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
configs = sys.argv[2:] or ["B", "C", "D"]
MAIN = {"B": ["nnf_xty_kernel", "nnf_xht_kernel", "nnf_cost_kernel", "nnf_hals_kernel", "nnf_hals_wave_kernel", "nnf_gram_cost_kernel",
              "nnf_gram_kernel"],
        "C": ["nnf_mu_left_kernel", "nnf_mu_right_kernel", "nnf_cost_kernel"],
        "D": ["nnf_mttkrp_rows_kernel", "nnf_xht_lds_kernel", "nnf_hals_wave_kernel", "nnf_gram_cost_kernel", "nnf_mu_left_kernel"],
        "E": ["nnf_xty_kernel", "nnf_xht_kernel", "nnf_hals_mfma_kernel", "nnf_hals_wave_kernel", "nnf_gram_cost_kernel", "nnf_gram_kernel"]}
SHAPES = {"B": [100000, 2000, 50], "C": [100000, 2000, 50], "D": [500, 500, 30], "E": [1000000, 4000, 100]}


def _meta(cfg):
    """What the collection belongs to: commit, shape, and the -D switches of the library it ran (bench.py attaches the traffic
    figures only to a run of the same shape and build)."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = "?"
    return {"commit": commit, "shape": SHAPES.get(cfg), "build_flags": bench.library_build_flags()}
for cfg in configs:
    O = os.path.join(G, f"prof_{cfg}")
    if not os.path.isdir(O):
        print("missing", O)
        continue
    # (regenerated here from the raw CSVs: the newest pass only, with the per-grid split of tools/prof_summary.py)
    import subprocess
    with open(os.path.join(P, f"{tag}_{cfg}_kernel_stats.txt"), "w") as fh:
        fh.write(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), os.path.join(O, "stats")],
                                capture_output=True, text=True).stdout)
    ks = sorted(glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(P, f"{tag}_{cfg}_kernel_stats.csv"))
    shutil.copy(os.path.join(O, "pmc_summary.txt"), os.path.join(P, f"{tag}_{cfg}_pmc_fetch_write_sq.txt"))
    line = [ln for ln in open(os.path.join(O, "bench.json")) if ln.startswith("{")]
    if line:
        with open(os.path.join(P, f"{tag}_{cfg}_bench_line.json"), "w") as fh:
            fh.write(line[-1])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    grids = {}
    for d in ("fetch", "write", "sq"):
        # (gpurun merges into gpurun_out/: files of earlier calls stay -- only the newest pass of each kind counts)
        for f in sorted(glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            rows = list(csv.DictReader(open(f)))
            # a kernel launched with several grids (W^T X on the matrix and on the row blocks of the once-per-run rounding
            # calibration; sweeps on full and last column blocks): only the launches of its LARGEST grid count
            for r in rows:
                for k in MAIN.get(cfg, []):
                    if k in r["Kernel_Name"]:
                        grids[k] = max(grids.get(k, 0), int(r["Grid_Size"]))
            for r in rows:
                for k in MAIN.get(cfg, []):
                    if k in r["Kernel_Name"] and int(r["Grid_Size"]) == grids[k]:
                        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        if not cs.get("FETCH_SIZE") or not cs.get("WRITE_SIZE"):
            continue
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        e = {"fetch_size_kib_raw": fetch, "write_size_kib": write, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
             "launches": len(cs["FETCH_SIZE"]), "grid_threads": grids.get(k)}
        if cs.get("SQ_VALU_MFMA_BUSY_CYCLES") and cs.get("SQ_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines (~ kernel duration each)
            e["mfma_busy_frac"] = (sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])) / \
                (32 * sum(cs["SQ_BUSY_CYCLES"]) / len(cs["SQ_BUSY_CYCLES"]))
        out[k] = e
    out["_meta"] = _meta(cfg)
    out["_note"] = ("separate --pmc passes of `python3 bench.py --config %s --steps 3 --warmup 1 --no-cpu --no-fixed --no-extra "
                    "--no-kernels` (tools/bench_profile.sh); FETCH_SIZE doubled per the gfx950 correction; means over the "
                    "launches of each kernel" % cfg)
    json.dump(out, open(os.path.join(P, f"{tag}_{cfg}_traffic.json"), "w"), indent=1)
    print(cfg, {k: v.get("hbm_bytes_per_launch") for k, v in out.items() if isinstance(v, dict)})
