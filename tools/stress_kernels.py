"""Randomised kernel-level sweep: the contraction / MU / cost entry points on strided, offset views of random shapes (row counts
around the tiling thresholds included) against fp64 evaluations on the device.  python tools/stress_kernels.py [seed] [cases]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nn_fac_amd.engine import get_engine
eng = get_engine()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.RandomState(seed)
g = torch.Generator(device="cuda").manual_seed(seed)
def rel(a, b): return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))
bad = 0
for c in range(cases):
    r = int(rng.choice([1, 2, 4, 5, 17, 18, 20, 33, 34, 36, 49, 50, 52, 64, 65, 66, 68, 100, 128]))
    m = int(rng.choice([1, 3, 64, 255, 1000, 4097, 65535, 65537, 98303, 98305, 100000, 131071, 131073, 200001]))
    n = int(rng.choice([1, 2, 3, 4, 7, 63, 64, 65, 127, 130, 257]))
    if m * n > 3e7:
        n = max(1, int(3e7 // m))
    pad, off = int(rng.choice([0, 0, 1, 3, 4])), int(rng.choice([0, 0, 1, 2]))
    big = torch.rand(m, n + pad + off, device="cuda", generator=g) + 0.05
    X = big[:, off:off + n]
    Ut = torch.rand(r, m, device="cuda", generator=g) + 0.05
    V = torch.rand(r, n, device="cuda", generator=g) + 0.05
    X64, U64, V64 = X.double(), Ut.double().t(), V.double()
    K = U64 @ V64
    try:
        checks = [("xty", eng.xty(X, Ut), U64.t() @ X64, 1e-5), ("xht", eng.xht(X, V), V64 @ X64.t(), 1e-5),
                  ("frob", eng.frob_resid(X, Ut, V), ((X64 - K) ** 2).sum().reshape(1), 1e-5)]
        beta = float(rng.choice([0.5, 1.0, 2.0, 3.0]))
        if beta == 1.0:
            wl = torch.clamp(U64 * ((X64 / K) @ V64.t() / V64.sum(dim=1)), min=1e-12)
            wr = torch.clamp(V64 * (U64.t() @ (X64 / K) / U64.sum(dim=0)[:, None]), min=1e-12)
        elif beta == 2.0:
            wl = torch.clamp(U64 * (X64 @ V64.t()) / (K @ V64.t()), min=1e-12)
            wr = torch.clamp(V64 * (U64.t() @ X64) / (U64.t() @ K), min=1e-12)
        else:
            gam = 1 / (2 - beta) if beta < 1 else (1 / (beta - 1) if beta > 2 else 1.0)
            wl = torch.clamp(U64 * ((K ** (beta - 2) * X64) @ V64.t() / (K ** (beta - 1) @ V64.t())) ** gam, min=1e-12)
            wr = torch.clamp(V64 * (U64.t() @ (K ** (beta - 2) * X64) / (U64.t() @ K ** (beta - 1))) ** gam, min=1e-12)
        checks += [(f"mu_left b{beta}", eng.mu_left(X, Ut, V, beta).t(), wl, 3e-5), (f"mu_right b{beta}", eng.mu_right(X, Ut, V, beta), wr, 3e-5)]
        for name, got, want, tol in checks:
            e = rel(got, want)
            if not (e < tol):
                bad += 1
                print("CASE", c, (m, n, r, pad, off), name, f"rel {e:.2e}", flush=True)
    except BaseException as ex:   # noqa: BLE001
        if isinstance(ex, KeyboardInterrupt):
            raise
        bad += 1
        print("CASE", c, (m, n, r, pad, off), "raised", type(ex).__name__, ex, flush=True)
    del big, X, Ut, V, X64, U64, V64, K
print(f"stress_kernels seed {seed}: {cases} cases, {bad} flagged")
