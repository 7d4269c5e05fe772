"""Randomised end-to-end parity sweep (HALS and MU drivers vs the CPU oracle) over odd shapes and ranks.  Test
infrastructure: imports oracle/.  python tools/stress_parity.py [seed] [cases]

Tolerances are DESIGN.md section 4's: HALS factors rel_fro <= 5e-4, cost <= 1e-3, inner sweep counts equal; MU
factors <= 2e-4 (general beta over odd shapes), cost <= 1e-3.  Two documented exceptions, both decided from the ORACLE's own
numbers, never from the device's:

* threshold noise -- the inner loop stops at the first sweep with eps < delta * eps0 (nnls.py:156).  When the fp64 oracle's
  own eps / (delta * eps0) at the sweep where the counts part is within THRESH_BAND of 1, the fp32 history of the operands
  (a state that differs by ~2e-4 from the oracle's after a few outer iterations moves that ratio by about as much) decides
  on which side of the threshold the sweep falls: the counts of that solve may then differ by one (seed 81, case 43:
  ratio 0.99968 in the oracle, 1.00002 on the device -- with the oracle run on the device's own operands the ratio is
  1.00010 and the counts agree, tools/probes/seed81_probe.py).  From that solve on the two runs are one sweep apart, so the
  rest of the case is compared with the looser bound LOOSE (one sweep of a solve that still moves by 1 % of its first
  sweep) and later counts are not compared.
* ill-conditioned Grams -- when the 2-norm condition number of the ORACLE's final U^T U or V V^T exceeds KAPPA (1e5; the
  shapes the 5e-4 was calibrated on sit at 1e2 ... 1e4), the fp32 rounding of the Gram / cross terms (~1e-7 relative) is
  amplified into the factors by that number: factors are compared with LOOSE, the cost -- which stays well determined --
  keeps its bound.  Typical: rank = the smaller dimension (seed 0 case 55), rank 128 of a 256 x 257 matrix of exactly that
  rank (seed 81 case 37: kappa 3e6, relV 1.4e-3 at equal sweep counts, cost 8.5e-6).  Beyond 1e7 the bound is 5e-3 (seeds 5 /
  6: kappa 1.3e8 / 1.8e8 give relV 2.1e-3 / 2.2e-3 with the wave-per-column V-side solve and 1.8e-3 / 1.7e-3 with the
  four-lanes-per-column one, at equal sweep counts, costs to 5e-5).
* exact fits -- when the oracle's first-sweep sum of squared steps of a solve is exactly 0 (a 1 x 1 problem after its first
  iteration: seed 5 case 59) the reference runs to maxiter (`eps >= delta * 0`), while fp32 leaves rounding noise in eps0 and
  the rule stops after two sweeps: counts are not compared from that solve on; costs are compared with an absolute floor of
  1e-9 ||X||^2 (a cost that IS rounding noise has no relative accuracy).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.nmf import compute_nmf
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
verbose = len(sys.argv) > 3
TOL_HALS, TOL_MU, TOL_COST, LOOSE, THRESH_BAND, DELTA, KAPPA = 5e-4, 2e-4, 1e-3, 2e-3, 2e-3, 0.01, 1e5
rng = np.random.RandomState(seed)
def rel(a, b): return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
def kappa(G):
    sv = np.linalg.svd(G, compute_uv=False)
    return float(sv[0] / max(sv[-1], 1e-300))

_orig, solve_logs = orc.hals_nnls_acc, []
def _logged(*a, **kw):                       # every inner solve of the oracle leaves its per-sweep sums of squared steps
    log = []
    kw["sweep_log"] = log
    out = _orig(*a, **kw)
    solve_logs.append(log)
    return out
orc.hals_nnls_acc = _logged

bad = notes = 0
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
    solve_logs.clear()
    try:
        U, V, costs, _ = compute_nmf(X, r, U0, V0, sweep_log=sw, **kw)
        Uo, Vo, co, _ = orc.compute_nmf(X.astype(np.float64), r, U0.astype(np.float64), V0.astype(np.float64), sweeps=swo, **kw)
    except Exception as e:   # noqa: BLE001
        print("CASE", c, (m, n, r, rule, beta, sp, nz), "raised", type(e).__name__, e); bad += 1; continue
    eu, ev = rel(U, Uo), rel(V, Vo)
    ec = max(abs(a - b) / max(abs(b), 1e-9 * float(np.sum(X.astype(np.float64) ** 2))) for a, b in zip(costs, co))
    tol_f, tol_c, why = (TOL_HALS if rule == "hals" else TOL_MU), TOL_COST, ""
    counts_ok = True
    if rule == "hals":
        kap = max(kappa(Uo.T @ Uo), kappa(Vo @ Vo.T))
        if kap > KAPPA:
            tol_f, why = (LOOSE if kap <= 1e7 else 5e-3), f"ill-conditioned: kappa {kap:.1e}"
        exact = next((i for i, lg in enumerate(solve_logs) if lg and lg[0] == 0.0), None)
        if exact is not None and sw[:exact] == swo[:exact]:
            sw, swo = sw[:exact], swo[:exact]
            why = (why + "; " if why else "") + f"exact fit from solve {exact} on (eps0 = 0 in fp64): counts not compared"
        if sw != swo:
            j = next(i for i, (a, b) in enumerate(zip(sw, swo)) if a != b)
            log, s = solve_logs[j], min(sw[j], swo[j])          # the oracle's sums of that solve; s = the earlier stop
            ratio = log[s - 1] / (DELTA * log[0]) if 1 <= s <= len(log) and log[0] > 0 else float("inf")
            if abs(sw[j] - swo[j]) == 1 and abs(ratio - 1.0) < THRESH_BAND:
                tol_f, tol_c = LOOSE, 2 * LOOSE
                why = f"threshold noise at solve {j}: oracle eps/(delta eps0) = {ratio:.6f} at sweep {s}"
            else:
                counts_ok = False
    flag = (not np.all(np.isfinite(costs))) or eu > tol_f or ev > tol_f or ec > tol_c or not counts_ok
    if flag:
        bad += 1
        print("CASE", c, (m, n, r, rule, beta, sp, nz), f"relU {eu:.1e} relV {ev:.1e} cost {ec:.1e} sweeps {sw} vs {swo}", why)
    elif why and (sw != swo or max(eu, ev) > TOL_HALS):
        notes += 1
        print("NOTE", c, (m, n, r, rule, beta), f"relU {eu:.1e} relV {ev:.1e} cost {ec:.1e} [{why}] sweeps {sw} vs {swo}")
    elif verbose:
        print("ok  ", c, (m, n, r, rule, beta), f"relU {eu:.1e} relV {ev:.1e} cost {ec:.1e}")
print(f"stress seed {seed}: {cases} cases, {bad} flagged ({notes} within a documented exception)")
