"""NTD HALS at 300^3, ranks 20^3 (not a BASELINE config): ms per iteration and where it goes.  NNF_NTD_PG_MULTI=0 selects the
one-workgroup core update."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.ntd import compute_ntd
g = torch.Generator(device="cuda").manual_seed(0)
I = J = K = 300; rk = [20, 20, 20]
Fs = [torch.rand(s, q, device="cuda", generator=g) for s, q in zip((I, J, K), rk)]
G0 = torch.rand(*rk, device="cuda", generator=g)
T = torch.einsum('abc,ia,jb,kc->ijk', G0, *Fs) + 1e-2 * torch.rand(I, J, K, device="cuda", generator=g)
F0 = [torch.rand(s, q, device="cuda", generator=g) for s, q in zip((I, J, K), rk)]
C0 = torch.rand(*rk, device="cuda", generator=g)
kw = dict(sparsity_coefficients=[None] * 4, normalize=[False] * 4, tol=0, deterministic=True)
compute_ntd(T, rk, C0, F0, n_iter_max=2, update_rule="hals", **kw)
pg = []
torch.cuda.synchronize(); t0 = time.time()
_, _, costs, _ = compute_ntd(T, rk, C0, F0, n_iter_max=10, update_rule="hals", return_costs=True, pg_log=pg, **kw)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"PG_MULTI={os.environ.get('NNF_NTD_PG_MULTI', '1')}: NTD hals 300^3 ranks 20: {dt*100:.3f} ms/iter; pg steps {pg}; cost {costs[-1]:.4e}")
