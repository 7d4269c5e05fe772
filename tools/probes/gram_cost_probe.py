"""How accurate is the Frobenius cost of an NMF iterate through the Gram identity
        ||X - U V||^2 = ||X||^2 - 2 <V, U^T X> + <U^T U, V V^T>
when U^T X, U^T U, V V^T are the fp32 outputs the iteration has on hand anyway (nnf_xty_f32, nnf_gram_f32) and only the three
small inner products are taken in fp64 (nnf_dot_f32)?  Compared with the fp64 residual (truth) and with the streaming cost
kernel (nnf_frob_resid_f32) on the iterates of config B's own run; each term is also swapped for its fp64 value in turn to
see which one carries the error.   python tools/probes/gram_cost_probe.py [m n r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nn_fac_amd.engine import get_engine
from nn_fac_amd import nmf as nmf_mod
m, n, r = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (100000, 2000, 50)
rng = np.random.RandomState(0)
W, H = rng.rand(m, r), rng.rand(r, n)
Xh = (W @ H + 1e-2 * rng.rand(m, n)).astype(np.float32)
U0, V0 = rng.rand(m, r).astype(np.float32), rng.rand(r, n).astype(np.float32)
dev = torch.device("cuda:0")
eng = get_engine(dev)
X = torch.from_numpy(Xh).to(dev)
Ut, V = torch.from_numpy(U0).to(dev).t().contiguous(), torch.from_numpy(V0).to(dev)
ws = nmf_mod._StepBuffers(X, r)
X64 = X.double()
N64 = float((X64 * X64).sum())


def truth(Ut, V):
    tot = 0.0
    for lo in range(0, m, 20000):
        R = X64[lo:lo + 20000] - Ut[:, lo:lo + 20000].double().t() @ V.double()
        tot += float((R * R).sum())
    return tot


done = 0
for upto in (1, 3, 10, 23, 60):
    Ut, V = nmf_mod.run_steps(eng, ws, X, r, Ut, V, upto - done, "hals", 2, [None, None], [], [False, False], True,
                              lambda it, c, sw: False)
    done = upto
    t = truth(Ut, V)
    direct = float(eng.frob_resid(X, Ut, V))
    UtM, G2, VVt = eng.xty(X, Ut), eng.gram(Ut), eng.gram(V)
    A, B = float(eng.dot(V, UtM)), float(eng.dot(G2, VVt))
    UtM64, G264, VVt64 = Ut.double() @ X64, Ut.double() @ Ut.double().t(), V.double() @ V.double().t()
    A64, B64 = float((V.double() * UtM64).sum()), float((G264 * VVt64).sum())
    Bg = float((G264 * VVt.double()).sum())     # only V V^T from the fp32 kernel
    Bv = float((G2.double() * VVt64).sum())     # only U^T U from the fp32 kernel
    ident = N64 - 2 * A + B

    def rel(x):
        return (x - t) / t
    print(f"after {upto:3d} iterations: cost {t:.6e}  cost/||X||^2 {t / N64:.2e}   direct kernel {rel(direct):+.2e}   identity {rel(ident):+.2e}"
          f"   [fp64 terms {rel(N64 - 2 * A64 + B64):+.1e}; only <V,UtX> fp32 {rel(N64 - 2 * A + B64):+.1e}; only UtU fp32 "
          f"{rel(N64 - 2 * A64 + Bv):+.1e}; only VVt fp32 {rel(N64 - 2 * A64 + Bg):+.1e}]", flush=True)
    e = (UtM.double() - UtM64) / UtM64
    g = (G2.double() - G264) / G264
    print(f"      entry errors: UtX mean {float(e.mean()):+.2e} rms {float(e.pow(2).mean().sqrt()):.2e};  UtU mean {float(g.mean()):+.2e} rms "
          f"{float(g.pow(2).mean().sqrt()):.2e}", flush=True)
