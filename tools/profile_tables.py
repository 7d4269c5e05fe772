"""Markdown tables for DESIGN.md section 7 / README.md / profiles/README.md, generated from the committed files under profiles/
(so that every quoted figure equals a tracked file):

    python tools/profile_tables.py [r04]            print them
    python tools/profile_tables.py r04 --write      rewrite the blocks between the `<!-- BEGIN/END GENERATED profile_tables -->`
                                                    markers of DESIGN.md, profiles/README.md (all tables) and README.md (one paragraph)
"""
import builtins, csv, json, os, re, sys
_out = []


def print(*a, **k):          # (collect instead of writing: the same text is printed or spliced into the documents)
    _out.append(" ".join(str(x) for x in a))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r04"
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def line(cfg):
    f = os.path.join(P, f"{tag}_{cfg}_bench_line.json")
    return json.load(open(f)) if os.path.exists(f) else None


def stats(cfg):
    """Rows of the kernel table; a kernel launched with several grids is represented by the launches of its largest grid (the
    "by grid" section of tools/prof_summary.py)."""
    f = os.path.join(P, f"{tag}_{cfg}_kernel_stats.txt")
    rows, big = {}, {}
    if not os.path.exists(f):
        return rows
    split = False
    for ln in open(f).read().splitlines()[1:]:
        if ln.startswith("== by grid"):
            split = True
            continue
        if not split:
            m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
            if m:
                rows[m.group(1).strip()] = dict(calls=int(m.group(2)), avg=float(m.group(4)), mn=float(m.group(5)), mx=float(m.group(6)))
        else:
            m = re.match(r"(.+?)\s+(\d+)x(\d+)x(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
            if m:
                name, threads = m.group(1).strip(), int(m.group(2)) * int(m.group(3)) * int(m.group(4))
                if threads > big.get(name, (0,))[0]:
                    big[name] = (threads, dict(calls=int(m.group(5)), avg=float(m.group(6)), mn=float(m.group(7)), mx=float(m.group(8)), grid=threads))
    for name, (_, row) in big.items():
        rows[name] = row
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
        if algo and unit == "TF":
            frac = f"{algo / (v['avg'] * 1e-6) / 1e12 / PEAK_TF:.3f} MFMA"
            alg = f"{algo:.1e} flop"
        elif algo and unit == "GB":
            frac = f"{algo / (v['avg'] * 1e-6) / 1e9 / PEAK_GBS:.3f} HBM"
            alg = f"{algo / 1e6:.1f} MB"
        hb = f"{tr['hbm_bytes_per_launch'] / 1e6:.1f} MB" if tr.get("hbm_bytes_per_launch") else ""
        mb = f"{tr['mfma_busy_frac']:.2f}" if tr.get("mfma_busy_frac") else ""
        gl = " (largest grid)" if "grid" in v else ""
        print(f"| `{k[:60]}`{gl} | {v['calls']} | {v['avg']:.1f} | {v['mn']:.1f} – {v['mx']:.1f} | {alg} | {frac} | {hb} | {mb} |")


def readme_paragraph():
    b, c, dd, e = line("B"), line("C"), line("D"), line("E")
    eb = json.load(open(be)) if os.path.exists(be) else None
    t = []
    if b:
        cb = b.get("cpu_baseline") or {}
        t.append(f"config B (100000×2000, rank 50) HALS {b['value']:.0f} outer iterations/s ({b['steady_state']['iterations_per_s']:.0f} in the settled "
                 f"window, {b['fixed_work']['iterations_per_s']:.0f} with 10 sweeps per solve; NumPy restatement of the reference on the same box's host, "
                 f"same inputs: {cb.get('value', float('nan')):.2f} on {cb.get('cores', '?')} threads); WᵀX {b['roofline']['launch_ms'] * 1e3:.1f} µs in the loop = "
                 f"{b['roofline']['frac']:.3f} of the fp32 MFMA peak")
    if c:
        t.append(f"MU β=1 {c['value']:.0f}")
    if dd:
        t.append(f"NTF 500³ rank 30 HALS {dd['value']:.0f}")
    if e:
        x = (b or {}).get("extra_configs", {}).get("E", {})
        t.append(f"the 10⁶×4000 rank-100 problem of config E on ONE GPU {e['value']:.1f} over 20 iterations"
                 + (f" ({x['iterations_per_s']:.1f} over the default line's short leg, before the cost leaves the Gram identity)" if x else ""))
    if eb:
        t.append(f"one rank's 125000-row block of it under the sharded protocol {eb['value']:.0f} ({eb['ms_per_step']:.2f} ms per iteration)")
    return "; ".join(t) + "."


text = "\n".join(_out)
if "--write" in sys.argv:
    B, E = "<!-- BEGIN GENERATED profile_tables -->", "<!-- END GENERATED profile_tables -->"
    for name, body in (("DESIGN.md", text), (os.path.join("profiles", "README.md"), text), ("README.md", readme_paragraph())):
        f = os.path.join(ROOT, name)
        doc = open(f).read()
        if B not in doc or E not in doc:
            builtins.print("no markers in", name)
            continue
        a, z = doc.index(B) + len(B), doc.index(E)
        open(f, "w").write(doc[:a] + f"\n(`python tools/profile_tables.py {tag} --write`)\n\n" + body + "\n" + doc[z:])
        builtins.print("rewrote", name)
else:
    builtins.print(text)
