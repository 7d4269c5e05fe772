"""Timing of the MU (config C) and NTF (config D) paths.  Usage: python tools/perf_probe2.py"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
m, n, r = 100000, 2000, 50
X = torch.rand(m, r, device="cuda", generator=g) @ torch.rand(r, n, device="cuda", generator=g) + 1e-2
Ut = torch.rand(r, m, device="cuda", generator=g) + 0.01
V = torch.rand(r, n, device="cuda", generator=g) + 0.01
fl = 2.0 * r * m * n
for beta in (1, 2, 0.5):
    for name, fn, k in (("mu_left", lambda: eng.mu_left(X, Ut, V, beta), 2), ("mu_right", lambda: eng.mu_right(X, Ut, V, beta), 2),
                        ("betadiv", lambda: eng.betadiv(X, Ut, V, beta), 1)):
        ms = timeit(fn)
        print(f"beta={beta:<4} {name:9s} {ms*1e3:8.1f} us  {k*fl/ms/1e9:7.2f} TF/s(useful)  {m*n*4/ms/1e6:7.1f} GB/s")
from nn_fac_amd.nmf import compute_nmf
U0 = torch.rand(m, r, device="cuda", generator=g); V0 = torch.rand(r, n, device="cuda", generator=g)
compute_nmf(X, r, U0, V0, n_iter_max=2, tol=0, update_rule="mu", beta=1)
torch.cuda.synchronize(); t0 = time.time()
_, _, costs, _ = compute_nmf(X, r, U0, V0, n_iter_max=10, tol=0, update_rule="mu", beta=1, return_costs=True)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"NMF MU beta=1 (config C): {dt*100:.3f} ms/iter -> {10/dt:.1f} it/s; costs {costs[0]:.4e} -> {costs[-1]:.4e}")
del X, Ut, V, U0, V0
# ---- config D: 500^3 rank 30 NTF
I = J = K = 500; R = 30
A, B, C = (torch.rand(s, R, device="cuda", generator=g) for s in (I, J, K))
T = torch.einsum('ir,jr,kr->ijk', A, B, C) + 1e-2 * torch.rand(I, J, K, device="cuda", generator=g)
Ft = [torch.rand(R, s, device="cuda", generator=g) for s in (I, J, K)]
for mode in range(3):
    ms = timeit(lambda: eng.mttkrp3(T, Ft, mode))
    print(f"mttkrp mode {mode}: {ms*1e3:8.1f} us  {2.0*I*J*K*R/ms/1e9:7.2f} TF/s  {I*J*K*4/ms/1e6:7.1f} GB/s")
ms = timeit(lambda: eng.cp3_betadiv(T, Ft, 2))
print(f"cp3 cost      : {ms*1e3:8.1f} us  {I*J*K*4/ms/1e6:7.1f} GB/s")
from nn_fac_amd.ntf import compute_ntf
F0 = [f.t() for f in Ft]
compute_ntf(T, R, F0, n_iter_max=2, tol=0, alpha=math.inf, sparsity_coefficients=[None]*3, normalize=[False]*3)
sw = []
torch.cuda.synchronize(); t0 = time.time()
_, costs, _ = compute_ntf(T, R, F0, n_iter_max=10, tol=0, alpha=math.inf, return_costs=True, sparsity_coefficients=[None]*3,
                          normalize=[False]*3, sweep_log=sw)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"NTF HALS (config D): {dt*100:.3f} ms/iter -> {10/dt:.1f} it/s; sweeps {sw[-3:]}; costs {costs[0]:.4e} -> {costs[-1]:.4e}")
del T, A, B, C
# ---- NTD: 300x300x300, ranks (20, 20, 20)
from nn_fac_amd.ntd import compute_ntd
I = J = K = 300; rk = [20, 20, 20]
Fs = [torch.rand(s, q, device="cuda", generator=g) for s, q in zip((I, J, K), rk)]
G0 = torch.rand(*rk, device="cuda", generator=g)
T = torch.einsum('abc,ia,jb,kc->ijk', G0, *Fs) + 1e-2 * torch.rand(I, J, K, device="cuda", generator=g)
F0 = [torch.rand(s, q, device="cuda", generator=g) for s, q in zip((I, J, K), rk)]
C0 = torch.rand(*rk, device="cuda", generator=g)
kw = dict(sparsity_coefficients=[None] * 4, normalize=[False] * 4, tol=0, deterministic=True)
for rule, beta in (("hals", 2), ("mu", 1)):
    compute_ntd(T, rk, C0, F0, n_iter_max=2, update_rule=rule, beta=beta, **kw)
    torch.cuda.synchronize(); t0 = time.time()
    _, _, costs, _ = compute_ntd(T, rk, C0, F0, n_iter_max=10, update_rule=rule, beta=beta, return_costs=True, **kw)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"NTD {rule} beta={beta} (300^3, ranks 20): {dt*100:.3f} ms/iter -> {10/dt:.1f} it/s; costs {costs[0]:.4e} -> {costs[-1]:.4e}")
