"""NTD workload for rocprofv3 (300^3, ranks 20): a few HALS iterations then a few MU iterations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.ntd import compute_ntd
g = torch.Generator(device="cuda").manual_seed(0)
I = J = K = 300; rk = [20, 20, 20]
Fs = [torch.rand(s, q, device="cuda", generator=g) for s, q in zip((I, J, K), rk)]
T = torch.einsum('abc,ia,jb,kc->ijk', torch.rand(*rk, device="cuda", generator=g), *Fs) + 1e-2 * torch.rand(I, J, K, device="cuda", generator=g)
F0 = [torch.rand(s, q, device="cuda", generator=g) for s, q in zip((I, J, K), rk)]
C0 = torch.rand(*rk, device="cuda", generator=g)
kw = dict(sparsity_coefficients=[None] * 4, normalize=[False] * 4, tol=0, deterministic=True)
rule = sys.argv[1] if len(sys.argv) > 1 else "hals"
compute_ntd(T, rk, C0, F0, n_iter_max=4, update_rule=rule, beta=2 if rule == "hals" else 1, **kw)
torch.cuda.synchronize()
print("done")
