"""A degenerate NTF shape flagged by tools/stress_tensor.py -- (17, 1, 64), rank 34 = twice the shortest real mode -- over 40
data seeds, next to rank 32 (full MFMA tiles) and rank 30: how the factor error against the fp64 oracle after 3 iterations is
distributed (conditioning vs a defect of the leftover-rank form of the fused cost + partial pass)."""
import math, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.ntf import compute_ntf
def rel(a, b): return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300))
shape = (17, 1, 64)
for R in (34, 33, 32, 30, 18, 17):
    errs, cerr, same = [], [], 0
    for c in range(40):
        T, F0 = orc.synth_ntf(shape, R, seed=c, dtype=np.float32)
        kw = dict(n_iter_max=3, tol=0, update_rule="hals", beta=2, alpha=math.inf, sparsity_coefficients=[None] * 3,
                  normalize=[False] * 3, return_costs=True)
        sw, swo = [], []
        F, costs, _ = compute_ntf(T, R, F0, sweep_log=sw, **kw)
        Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], sweeps=swo, **kw)
        errs.append(max(rel(a, b) for a, b in zip(F, Fo)))
        cerr.append(max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(costs, co)))
        same += int(list(sw) == list(swo))
    e = np.array(errs)
    print(f"R={R:3d}: factor rel err median {np.median(e):.1e} max {e.max():.1e}  (> 2e-3: {(e > 2e-3).sum()}/40)  cost err max {max(cerr):.1e}  "
          f"sweep counts equal {same}/40", flush=True)
