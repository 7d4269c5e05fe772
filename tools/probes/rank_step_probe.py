"""W^T X / X H^T launch time by rank around the leftover-rank instantiations (16q+1..4 run q MFMA tiles + 2 or 4 ranks on the
VALU pipe, 16q+5.. run q+1 tiles): is the (q, 4) form faster than the padded (q+1, 0) one?  python tools/probes/rank_step_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
m, n = 100000, 2000
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.rand(m, n, device="cuda", generator=g)
for r in (18, 20, 21, 34, 36, 37, 50, 52, 53, 66, 68, 69, 80, 100, 112, 128):
    Ut = torch.rand(r, m, device="cuda", generator=g)
    V = torch.rand(r, n, device="cuda", generator=g)
    for _ in range(2): eng.xty(X, Ut); eng.xht(X, V)
    a = min(eng.time_kernel("xty", lambda: eng.xty(X, Ut)) for _ in range(2))
    b = min(eng.time_kernel("xht", lambda: eng.xht(X, V)) for _ in range(2))
    print(f"rank {r:3d}: W^T X {1e3 * a:7.1f} us   X H^T {1e3 * b:7.1f} us", flush=True)
