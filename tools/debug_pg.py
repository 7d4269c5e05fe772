"""Dev tool: projected-gradient core kernel vs a NumPy loop, printing the status block."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda").contiguous()
for dims, sparse in [((9, 9, 3), 0.0), ((4, 3, 2), 0.05), ((16, 12, 20), 0.0), ((1, 5, 1), 0.0)]:
    rng = np.random.RandomState(sum(dims))
    shape = tuple(5 * d + 3 for d in dims)
    F = [rng.rand(shape[i], dims[i]) for i in range(3)]
    T = orc.multi_mode_dot(rng.rand(*dims), F) + 0.01 * rng.rand(*shape)
    MtX = orc.multi_mode_dot(T, F, transpose=True).astype(np.float32).astype(np.float64)
    M = [(f.T @ f).astype(np.float32).astype(np.float64) for f in F]
    core0 = rng.rand(*dims).astype(np.float32).astype(np.float64)
    sig = [np.linalg.svd(m_, compute_uv=False)[0] for m_ in M]
    step = round(float(np.prod([1 / s for s in sig])), 6)
    core, cnt, upd0, upd, hist = core0.copy(), 1, 0, 1, []
    while cnt <= 300 and upd >= 0.01 * upd0:
        grad = -MtX + orc.multi_mode_dot(core, M) + sparse
        dc = np.minimum(step * grad, core); core = core - dc; upd = np.sqrt(np.sum(dc ** 2)); hist.append(upd)
        if cnt == 1: upd0 = upd
        cnt += 1
    nrm2 = float(np.sum(T ** 2))
    st = eng.ntd_core_pg(dev(core0), dev(MtX), [dev(m_) for m_ in M], sparse, 0.01, 300, nrm2).cpu().numpy()
    print(dims, "numpy: iters", cnt - 1, "step", step, "raw", float(np.prod([1 / s for s in sig])), "upd0", upd0, "upd", upd)
    print("      kernel:", st)
