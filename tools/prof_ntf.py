"""NTF HALS workload for rocprofv3 (config D: 500^3, R = 30, 6 iterations)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.ntf import compute_ntf
g = torch.Generator(device="cuda").manual_seed(0)
I = J = K = 500; R = 30
A, B, C = (torch.rand(s, R, device="cuda", generator=g) for s in (I, J, K))
T = torch.einsum('ir,jr,kr->ijk', A, B, C) + 1e-2 * torch.rand(I, J, K, device="cuda", generator=g)
F0 = [torch.rand(s, R, device="cuda", generator=g) for s in (I, J, K)]
compute_ntf(T, R, F0, n_iter_max=6, tol=0, alpha=math.inf, sparsity_coefficients=[None] * 3, normalize=[False] * 3)
torch.cuda.synchronize()
print("done")
