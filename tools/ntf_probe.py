"""NTF config D through compute_ntf, N iterations; NNF_NTF_OVERLAP=0 disables the overlapped cost (tuning probe)."""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.ntf import compute_ntf
g = torch.Generator(device="cuda").manual_seed(0)
I = 500; R = 30; N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
A, B, C = (torch.rand(I, R, device="cuda", generator=g) for _ in range(3))
T = torch.einsum('ir,jr,kr->ijk', A, B, C) + 1e-2 * torch.rand(I, I, I, device="cuda", generator=g)
F0 = [torch.rand(I, R, device="cuda", generator=g) for _ in range(3)]
kw = dict(tol=0, alpha=math.inf, sparsity_coefficients=[None] * 3, normalize=[False] * 3)
compute_ntf(T, R, F0, n_iter_max=3, **kw)
torch.cuda.synchronize(); t0 = time.time()
_, costs, toc = compute_ntf(T, R, F0, n_iter_max=N, return_costs=True, **kw)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"NTF HALS D: {dt/N*1e3:.3f} ms/iter ({N} iterations) -> {N/dt:.1f} it/s; cost {costs[-1]:.4e}", flush=True)
