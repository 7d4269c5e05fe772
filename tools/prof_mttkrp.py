"""MTTKRP / CP-cost workload for rocprofv3 passes (config D: 500^3, R = 30)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
I = J = K = 500; R = 30
T = torch.rand(I, J, K, device="cuda", generator=g)
Ft = [torch.rand(R, s, device="cuda", generator=g) for s in (I, J, K)]
for _ in range(4):
    for mode in range(3):
        eng.mttkrp3(T, Ft, mode)
    eng.cp3_betadiv(T, Ft, 2)
torch.cuda.synchronize()
print("done")
