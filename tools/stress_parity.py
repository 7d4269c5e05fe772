"""Randomised end-to-end parity sweep (HALS and MU drivers vs the CPU oracle) over odd shapes and ranks.  Test
infrastructure: imports oracle/.  python tools/stress_parity.py [seed] [cases]"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.nmf import compute_nmf
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.RandomState(seed)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
bad = 0
for c in range(cases):
    r = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 32, 33, 48, 50, 63, 64, 65, 100, 127, 128]))
    m = int(rng.choice([r, r + 1, 64, 97, 255, 256, 257, 700, 1500]))
    n = int(rng.choice([r, r + 3, 16, 61, 64, 130, 257, 600]))
    m, n = max(m, r), max(n, r)
    rule, beta = [("hals", 2), ("mu", 1), ("mu", 2), ("mu", 0.5), ("mu", 0), ("mu", 3)][rng.randint(6)]
    if rule == "mu" and r > 64 and beta != 2 and m * n > 300000:
        continue
    X = (rng.rand(m, r) @ rng.rand(r, n) + 1e-2 * rng.rand(m, n)).astype(np.float32)
    if rng.rand() < 0.3 and rule == "hals":
        X[rng.rand(m, n) < 0.3] = 0.0
    U0, V0 = rng.rand(m, r).astype(np.float32) + 0.01, rng.rand(r, n).astype(np.float32) + 0.01
    sp = [None, None] if rule == "mu" or rng.rand() < 0.6 else [float(rng.rand() * 0.1), float(rng.rand() * 0.1)]
    nz = [False, bool(rng.rand() < 0.3)] if rule == "hals" else [False, False]
    kw = dict(n_iter_max=4, tol=0, update_rule=rule, beta=beta, sparsity_coefficients=sp, normalize=nz, return_costs=True,
              deterministic=True)
    sw, swo = [], []
    try:
        U, V, costs, _ = compute_nmf(X, r, U0, V0, sweep_log=sw, **kw)
        Uo, Vo, co, _ = orc.compute_nmf(X.astype(np.float64), r, U0.astype(np.float64), V0.astype(np.float64), sweeps=swo, **kw)
    except Exception as e:   # noqa: BLE001
        print("CASE", c, (m, n, r, rule, beta, sp, nz), "raised", type(e).__name__, e); bad += 1; continue
    eu, ev = rel(U, Uo), rel(V, Vo)
    ec = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(costs, co))
    tol_f = 2e-3 if rule == "hals" else 2e-4
    flag = (not np.all(np.isfinite(costs))) or eu > tol_f or ev > tol_f or ec > 2e-3 or (rule == "hals" and sw != swo)
    if flag:
        bad += 1
        print("CASE", c, (m, n, r, rule, beta, sp, nz), f"relU {eu:.1e} relV {ev:.1e} cost {ec:.1e} sweeps {sw} vs {swo}")
print(f"stress seed {seed}: {cases} cases, {bad} flagged")
