"""Frobenius cost pass (nnf_cost_kernel<FROB>) launch time by rank at 100000 x 2000: the LDS footprint (U fragments + one or two V
images) decides how many workgroups share a CU.  python tools/probes/cost_rank_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
m, n = 100000, 2000
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.rand(m, n, device="cuda", generator=g)
for r in (48, 50, 52, 56, 64, 72, 76, 80, 100, 104, 112, 128):
    Ut = torch.rand(r, m, device="cuda", generator=g)
    V = torch.rand(r, n, device="cuda", generator=g)
    for _ in range(2): eng.frob_resid(X, Ut, V)
    a = min(eng.time_kernel("cost", lambda: eng.frob_resid(X, Ut, V)) for _ in range(2))
    print(f"rank {r:3d}: cost pass {1e3 * a:7.1f} us", flush=True)
