"""Randomised shapes above rank 128: products, costs and solves against float64 / the oracle (odd sizes, ragged chunks, ranks that
leave 1 ... 127 rows in the last chunk).   python tools/probes/bigrank_stress.py [cases=40] [seed=0]"""
import math
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nnfac_oracle as orc  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()  # noqa: E731
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))  # noqa: E731


def run(cases=40, seed=0, verbose=True):
    """The cases that need a look (empty: all inside 3e-5 for products / costs, 3e-4 for the solve, equal sweep counts)."""
    rng = np.random.RandomState(seed)
    eng = get_engine("cuda:0")
    bad = []
    for c in range(cases):
        _case(eng, rng, c, bad, verbose)
    return bad


def _case(eng, rng, c, bad, verbose):
    r = int(rng.choice([129, 130, 131, 144, 160, 191, 192, 193, 200, 255, 256, 257, 300, 383, 385, 400]))
    m = int(rng.choice([r, r + 1, 257, 500, 777, 1000, 2049, 5000]))
    n = int(rng.choice([1, 3, 63, 64, 65, 130, 257, 500, 1001]))
    m = max(m, 1)
    X = rng.rand(m, n).astype(np.float32) + 0.05
    Ut = (rng.rand(r, m) / math.sqrt(r)).astype(np.float32)
    V = rng.rand(r, n).astype(np.float32)
    X64, U64, V64 = X.astype(np.float64), Ut.astype(np.float64), V.astype(np.float64)
    Xd, Utd, Vd = dev(X), dev(Ut), dev(V)
    P = U64.T @ V64
    errs = {"gramV": rel(eng.gram(Vd).cpu().numpy(), V64 @ V64.T), "gramU": rel(eng.gram(Utd).cpu().numpy(), U64 @ U64.T),
            "xty": rel(eng.xty(Xd, Utd).cpu().numpy(), U64 @ X64), "xht": rel(eng.xht(Xd, Vd).cpu().numpy(), V64 @ X64.T)}
    w = np.sum((X64 - P) ** 2)
    errs["frob"] = abs(float(eng.frob_resid(Xd, Utd, Vd)) - w) / w
    for beta in (1, 0.5):
        w = orc.beta_divergence(X64, P, beta)
        errs[f"beta{beta}"] = abs(float(eng.betadiv(Xd, Utd, Vd, beta)) - w) / abs(w)
    # a solve on the V side of this shape
    UtU, UtM = U64 @ U64.T + 1e-3 * np.eye(r), U64 @ X64
    opts = [{}, {"sparsity_coefficient": 0.01}, {"normalize": True}, {"nonzero": True}][c % 4]
    Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM, UtU, V64.copy(), maxiter=12, alpha=math.inf, delta=0.01, **opts)
    Vs = dev(V)
    st = eng.hals_solve(dev(UtM), dev(UtU), Vs, 12, delta=0.01, sparsity=opts.get("sparsity_coefficient"),
                        normalize=opts.get("normalize", False), nonzero=opts.get("nonzero", False)).cpu()
    errs["hals"] = rel(Vs.cpu().numpy(), Vo)
    cnt_ok = int(st[1]) == cnto and int(st[3]) == 0
    worst = max(errs.values())
    flag = "" if (worst < 3e-4 and max(v for k, v in errs.items() if k != "hals") < 3e-5 and cnt_ok) else "   <-- CHECK"
    if flag:
        bad.append((c, m, n, r, errs, int(st[1]), cnto))
    if verbose:
        print(f"case {c:3d}  {m:5d} x {n:4d} rank {r:3d} {str(opts):34s} worst {worst:.1e}  sweeps {int(st[1]) - 1}/{cnto - 1}{flag}", flush=True)


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("cases to check:", len(bad))
    for b in bad:
        print(b)
