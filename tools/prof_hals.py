"""HALS-only workload for rocprofv3 counter passes: U-side (r x m) and V-side (r x n) fixed-count sweeps + solves."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine
m, n, r = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (100000, 2000, 50)
eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand(m, r, device="cuda", generator=g) @ torch.rand(r, n, device="cuda", generator=g)
Ut = torch.rand(r, m, device="cuda", generator=g)
V = torch.rand(r, n, device="cuda", generator=g)
UtM, UtU = eng.xht(X, V), eng.gram(V)
VtM, VtV = eng.xty(X, Ut), eng.gram(Ut)
for _ in range(3):
    F = Ut.clone(); eng.hals_sweeps(UtM, UtU, F, 20)
    F = Ut.clone(); eng.hals_solve(UtM, UtU, F, 20, delta=0.0)
    F = V.clone(); eng.hals_solve(VtM, VtV, F, 50, delta=0.0)
torch.cuda.synchronize()
print("done")
