"""Randomised parity sweep of the NTF / NTD drivers and the single-call NNLS entry points vs the CPU oracle over odd shapes
and ranks.  Test infrastructure: imports oracle/.  python tools/stress_tensor.py [seed] [cases]"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.ntf import compute_ntf
from nn_fac_amd.ntd import compute_ntd
from nn_fac_amd.update_rules.nnls import hals_nnls_acc, hals_coupling_nnls_acc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.RandomState(seed)
def rel(a, b): return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))
bad = 0
def flag(tag, cfg, msg):
    global bad
    bad += 1
    print(tag, cfg, msg, flush=True)
for c in range(cases):
    kind = ["ntf", "ntd", "nnls", "coupling"][rng.randint(4)]
    try:
        if kind == "ntf":
            shape = [int(rng.choice([2, 5, 16, 17, 33, 64, 70])) for _ in range(3)]
            if rng.rand() < 0.3:
                shape[rng.randint(3)] = 1                      # at most one mode of length 1
            shape = tuple(shape)
            R = int(rng.choice([1, 2, 3, 7, 16, 17, 30, 64]))
            R = min(R, 2 * min(s_ for s_ in shape if s_ > 1))  # far more components than rows: nothing is identifiable
            rule, beta = [("hals", 2), ("mu", 1), ("mu", 2), ("mu", 0.5)][rng.randint(4)]
            T, F0 = orc.synth_ntf(shape, R, seed=c, dtype=np.float32)
            kw = dict(n_iter_max=3, tol=0, update_rule=rule, beta=beta, alpha=math.inf, sparsity_coefficients=[None] * 3,
                      normalize=[False] * 3, return_costs=True)
            F, costs, _ = compute_ntf(T, R, F0, **kw)
            Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], **kw)
            e = max(rel(a, b) for a, b in zip(F, Fo))
            ec = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(costs, co))   # normalised costs: 1e-6 = an exact fit
            if not np.all(np.isfinite(costs)) or e > 2e-3 or ec > 5e-3:
                flag("NTF", (shape, R, rule, beta), f"rel {e:.1e} cost {ec:.1e}")
        elif kind == "ntd":
            shape = tuple(int(rng.choice([2, 5, 16, 17, 33, 40])) for _ in range(3))
            ranks = tuple(int(min(s, rng.choice([1, 2, 3, 5, 9, 16]))) for s in shape)
            rule, beta = [("hals", 2), ("mu", 1), ("mu", 2)][rng.randint(3)]
            F = [rng.rand(s, q) for s, q in zip(shape, ranks)]
            T = (np.einsum('abc,ia,jb,kc->ijk', rng.rand(*ranks), *F) + 1e-2 * rng.rand(*shape)).astype(np.float32)
            C0 = (rng.rand(*ranks) + 0.01).astype(np.float32)
            F0 = [(rng.rand(s, q) + 0.01).astype(np.float32) for s, q in zip(shape, ranks)]
            kw = dict(n_iter_max=3, tol=0, update_rule=rule, beta=beta, sparsity_coefficients=[None] * 4, normalize=[False] * 4,
                      return_costs=True, deterministic=True)
            Cg, Fg, costs, _ = compute_ntd(T, ranks, C0, F0, **kw)
            Co, Fo, co, _ = orc.compute_ntd(T.astype(np.float64), ranks, C0.astype(np.float64), [f.astype(np.float64) for f in F0], **kw)
            e = max([rel(Cg, Co)] + [rel(a, b) for a, b in zip(Fg, Fo)])
            ec = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(costs, co))
            if not np.all(np.isfinite(costs)) or e > 5e-3 or ec > 5e-3:
                flag("NTD", (shape, ranks, rule, beta), f"rel {e:.1e} cost {ec:.1e}")
        else:
            r = int(rng.choice([1, 2, 3, 16, 17, 32, 33, 50, 64, 65, 100, 128]))
            n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 257, 1000, 40000]))
            U = rng.rand(r + 20, r)
            M = U @ rng.rand(r, n) + 1e-2 * rng.rand(r + 20, n)
            UtU, UtM, V0 = U.T @ U, U.T @ M, rng.rand(r, n)
            kw = dict(maxiter=int(rng.choice([1, 2, 9, 40])), delta=float(rng.choice([0.0, 0.01, 0.3])), alpha=math.inf,
                      normalize=bool(rng.rand() < 0.25))
            if kind == "nnls":
                if rng.rand() < 0.4:
                    kw["sparsity_coefficient"] = float(rng.rand() * 0.2)
                if rng.rand() < 0.3:
                    kw["nonzero"] = True
                    UtM = UtM - 0.6 * np.abs(UtM).max() * (rng.rand(r, 1) < 0.3)    # some rows are driven to zero
                V, eps, cnt, _ = hals_nnls_acc(UtM, UtU, V0, **kw)
                Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM, UtU, V0, **kw)
            else:
                Vt, mu = rng.rand(r, n), float(rng.choice([0.0, 0.3, 10.0]))
                V, eps, cnt, _ = hals_coupling_nnls_acc(UtM, UtU, V0, Vt, mu, **kw)
                Vo, epso, cnto, _ = orc.hals_coupling_nnls_acc(UtM, UtU, V0, Vt, mu, **kw)
            e = rel(V, Vo)
            if cnt != cnto or e > 5e-4 or not np.isfinite(eps):
                flag(kind.upper(), (r, n, kw), f"rel {e:.1e} cnt {cnt} vs {cnto} eps {eps:.3e} vs {epso:.3e}")
    except BaseException as ex:   # noqa: BLE001  (the package's exceptions derive from BaseException, like the reference's)
        if isinstance(ex, KeyboardInterrupt):
            raise
        flag(kind.upper(), c, f"raised {type(ex).__name__}: {ex}")
print(f"stress_tensor seed {seed}: {cases} cases, {bad} flagged")
