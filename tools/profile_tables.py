"""Markdown tables for DESIGN.md section 7 / README.md / profiles/README.md, generated from the committed files under profiles/
(so that every quoted figure equals a tracked file):   python tools/profile_tables.py [r04]"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def line(cfg):
    f = os.path.join(P, f"{tag}_{cfg}_bench_line.json")
    return json.load(open(f)) if os.path.exists(f) else None


def stats(cfg):
    f = os.path.join(P, f"{tag}_{cfg}_kernel_stats.txt")
    rows = {}
    if os.path.exists(f):
        for ln in open(f).read().splitlines()[1:]:
            m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
            if m:
                rows[m.group(1).strip()] = dict(calls=int(m.group(2)), avg=float(m.group(4)), mn=float(m.group(5)), mx=float(m.group(6)))
    return rows


def find(rows, key):
    for k, v in rows.items():
        if key in k:
            return k, v
    return None, None


print(f"### Headline lines (`profiles/{tag}_<cfg>_bench_line.json`: `python bench.py --config <cfg> --steps 20 --warmup 3`, one box, one gpurun call)\n")
print("| config | iterations/s | ms per iteration | CPU baseline (same inputs, host of the same box) | dominant kernel: in-loop events | of peak |")
print("|---|---|---|---|---|---|")
for cfg in "BCDE":
    d = line(cfg)
    if not d:
        continue
    r = d["roofline"]
    cb = d.get("cpu_baseline") or {}
    unit = "MFMA" if r["bound"] == "mfma" else "HBM"
    print(f"| {cfg} | {d['value']:.1f} | {d['ms_per_step']:.3f} | {cb.get('value', float('nan')):.3g} it/s on {cb.get('cores', '?')} threads | "
          f"`{r['kernel'].split(' ')[0]}` {r['launch_ms'] * 1e3:.1f} µs | {r['frac']:.3f} {unit} |")
d = line("B")
if d:
    print()
    for k in ("steady_state", "fixed_work", "nondeterministic"):
        if k in d:
            e = d[k]
            print(f"* B `{k}`: {e['iterations_per_s']:.1f} iterations/s ({e['ms_per_step']:.3f} ms)" + (f", {e['inner_sweeps_mean']:.1f} inner sweeps per iteration" if 'inner_sweeps_mean' in e else ""))
    for n, e in (d.get("extra_configs") or {}).items():
        if "iterations_per_s" in e:
            print(f"* B line, `extra_configs.{n}`: {e['iterations_per_s']:.1f} iterations/s ({e['ms_per_step']:.3f} ms)")
be = os.path.join(P, f"{tag}_E_block_bench_line.json")
if os.path.exists(be):
    e = json.load(open(be))
    print(f"* one rank's 125000-row block of E under the forced sharded protocol (1-rank RCCL group, `{tag}_E_block_*`): {e['value']:.1f} iterations/s ({e['ms_per_step']:.3f} ms)")

KEYS = {"B": [("nnf_xty_kernel", 2.0e10, "TF"), ("nnf_xht_kernel", 2.0e10, "TF"), ("nnf_hals_kernel<50", None, None), ("nnf_hals_wave_kernel", None, None),
              ("nnf_gram_cost_kernel", None, None)],
        "C": [("nnf_mu_left_kernel<3, 2, 3", 4.0e10, "TF"), ("nnf_mu_right_kernel", 4.0e10, "TF"), ("nnf_mu_left_kernel<3, 2, 1", 4.0e10, "TF")],
        "D": [("nnf_mttkrp_rows_kernel", 500.2e6, "GB"), ("nnf_xht_lds_kernel", 530.2e6, "GB"), ("nnf_hals_wave_kernel", None, None), ("nnf_gram_cost_kernel", None, None)],
        "E": [("nnf_hals_mfma_kernel", None, None), ("nnf_xht_kernel", 8.0e11, "TF"), ("nnf_xty_kernel", 8.0e11, "TF"), ("nnf_hals_wave_kernel", None, None),
              ("nnf_cost_kernel", 8.0e11, "TF")]}
for cfg in "BCDE":
    rows = stats(cfg)
    tf = os.path.join(P, f"{tag}_{cfg}_traffic.json")
    traffic = json.load(open(tf)) if os.path.exists(tf) else {}
    if not rows:
        continue
    print(f"\n### Config {cfg}: `rocprofv3 --kernel-trace --stats` of the loop (`{tag}_{cfg}_kernel_stats.txt`), PMC traffic (`{tag}_{cfg}_traffic.json`)\n")
    print("| kernel | calls | mean µs | min – max | algorithmic | of peak (mean) | HBM traffic (PMC) | MFMA busy |")
    print("|---|---|---|---|---|---|---|---|")
    for key, algo, unit in KEYS[cfg]:
        k, v = find(rows, key)
        if not v:
            continue
        tr = next((t for tk, t in traffic.items() if isinstance(t, dict) and tk in k), {})
        frac = ""
        alg = ""
        if algo and unit == "TF" and not (cfg in "BE" and "xty" in key and v["calls"] > 14):
            frac = f"{algo / (v['avg'] * 1e-6) / 1e12 / PEAK_TF:.3f} MFMA"
            alg = f"{algo:.1e} flop"
        elif algo and unit == "GB":
            frac = f"{algo / (v['avg'] * 1e-6) / 1e9 / PEAK_GBS:.3f} HBM"
            alg = f"{algo / 1e6:.1f} MB"
        elif algo:
            alg = f"{algo:.1e} flop (mean includes the short calibration launches)"
        hb = f"{tr['hbm_bytes_per_launch'] / 1e6:.1f} MB" if tr.get("hbm_bytes_per_launch") else ""
        mb = f"{tr['mfma_busy_frac']:.2f}" if tr.get("mfma_busy_frac") else ""
        print(f"| `{k[:60]}` | {v['calls']} | {v['avg']:.1f} | {v['mn']:.1f} – {v['mx']:.1f} | {alg} | {frac} | {hb} | {mb} |")
