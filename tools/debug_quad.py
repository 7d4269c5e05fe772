"""Dev tool: compare the quad-layout HALS solve with the oracle on small shapes and print where they differ."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
os.environ["NNF_HALS_FORCE"] = "quad"
for (r, n, sweeps) in [(10, 100, 1), (10, 100, 3), (3, 1, 2), (16, 64, 1), (20, 64, 1), (50, 129, 1)]:
    rng = np.random.RandomState(r * 1000 + n)
    A = rng.rand(4 * r, r)
    UtU, UtM, V0 = A.T @ A, A.T @ rng.rand(4 * r, n), rng.rand(r, n)
    Vo, *_ = orc.hals_nnls_acc(UtM, UtU, V0, maxiter=sweeps, alpha=math.inf, delta=0.0)
    Vd = torch.tensor(V0, dtype=torch.float32, device="cuda")
    eng.hals_sweeps(torch.tensor(UtM, dtype=torch.float32, device="cuda"), torch.tensor(UtU, dtype=torch.float32, device="cuda"), Vd, sweeps)
    got = Vd.cpu().numpy()
    err = np.abs(got - Vo)
    rows = np.where(err.max(axis=1) > 1e-4)[0]
    print(f"r={r} n={n} sweeps={sweeps}: max err {err.max():.3e}; bad rows {rows.tolist()}; bad cols (first 10) {np.where(err.max(axis=0) > 1e-4)[0][:10].tolist()}")
    if len(rows):
        k = rows[0]
        print("   row", k, "got", got[k, :4], "want", Vo[k, :4], "in", V0[k, :4])
