"""The stand-alone cost passes at config B's shape (100000 x 2000, rank 50) and nothing else: the subject of `rocprofv3 --pmc`
passes on nnf_cost_kernel (HBM traffic against the 820 MB it needs: X once + the factors)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402

eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
m, n, r = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "100000x2000x50").split("x"))
X = torch.rand(m, n, device="cuda", generator=g) + 0.05
Ut = torch.rand(r, m, device="cuda", generator=g)
V = torch.rand(r, n, device="cuda", generator=g)
for _ in range(4):
    eng.frob_resid(X, Ut, V)
    eng.betadiv(X, Ut, V, 1)
torch.cuda.synchronize()
