"""Microseconds per blind HALS sweep (nnf_hals_sweeps_f32) at a few ranks / column counts; NNF_HALS_TWOCOL=0|1 in the environment
selects the one- or two-columns-per-lane resident kernel at ranks 80..104."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
cases = [(int(a), int(b)) for a, b in (x.split("x") for x in sys.argv[1:])] or [(100, 125000), (100, 131072), (96, 125000), (80, 125000), (100, 70000)]
for r, m in cases:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(300, r, device="cuda", generator=g)
    G = (A.t() @ A).contiguous()
    cross = (A.t() @ (A @ torch.rand(r, 2000, device="cuda", generator=g)))[:, torch.arange(m, device="cuda") % 2000].contiguous()
    F0 = torch.rand(r, m, device="cuda", generator=g)
    F = F0.clone()
    eng.hals_sweeps(cross, G, F, 3)
    torch.cuda.synchronize()
    ns = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    F = F0.clone()
    e0.record()
    nd = eng.hals_sweeps(cross, G, F, ns)
    e1.record()
    torch.cuda.synchronize()
    print(f"r={r:4d} cols={m:8d}: {e0.elapsed_time(e1) * 1e3 / ns:8.2f} us per sweep   nd[-1]={float(nd[-1]):.6e}  sum={float(F.double().sum()):.9e}", flush=True)
