"""Streaming-contraction workload for rocprofv3 passes: xty / xht / cost at config B (or given shape)."""
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
for _ in range(5):
    eng.xty(X, Ut); eng.xht(X, V); eng.frob_resid(X, Ut, V)
torch.cuda.synchronize()
print("done")
