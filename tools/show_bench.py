#!/usr/bin/env python3
"""Prints the headline numbers of a bench.py JSON line: value, ms per step and every roofline entry (kernel, fraction, launch time)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(f"{d['config']['workload'][:60]}: {d['value']:.1f} {d['unit']}  ({d['ms_per_step']:.3f} ms/step, n_gpus {d['n_gpus']})")
for r in ([d["roofline"]] if d.get("roofline") else []) + list(d.get("roofline_more") or []):
    print(f"  {r['kernel'][:70]:70s} frac {r['frac']:.3f}  {r['launch_ms'] * 1e3:8.1f} us  [{r['bound']}]")
for k, v in (d.get("extra_configs") or {}).items():
    print(f"  extra {k}: {v.get('value', v.get('iterations_per_s'))} iterations/s ({v.get('ms_per_step')} ms/step)")
