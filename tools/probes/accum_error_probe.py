"""Rounding of the fp32 cross product U^T X and Gram U^T U against fp64, per entry (relative rms), as a function of the number of
rows summed -- the constants the Gram-identity cost's error estimate (nnf_gram_cost_kernel) is built on.
    python tools/probes/accum_error_probe.py [MxNxR ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(100000, 2000, 50), (1000000, 2000, 50), (125000, 4000, 100), (1000000, 4000, 100)]
for m, n, r in cases:
    g = torch.Generator(device="cuda").manual_seed(3)
    W = torch.rand(m, r, device="cuda", generator=g)
    H = torch.rand(r, n, device="cuda", generator=g)
    X = W @ H
    X += 1e-2 * torch.rand(m, n, device="cuda", generator=g)
    Ut = torch.rand(r, m, device="cuda", generator=g)
    got = eng.xty(X, Ut).double()
    gg = eng.gram(Ut).double()
    ref = torch.zeros(r, n, dtype=torch.float64, device="cuda")
    gref = torch.zeros(r, r, dtype=torch.float64, device="cuda")
    step = 1 << 15
    for lo in range(0, m, step):
        u = Ut[:, lo:lo + step].double()
        ref += u @ X[lo:lo + step].double()
        gref += u @ u.t()
    ea = ((got - ref) / ref)
    eb = ((gg - gref) / gref)
    print(f"m={m:8d} n={n:5d} r={r:3d}: U^T X rel err rms {float(ea.pow(2).mean().sqrt()):.2e} max {float(ea.abs().max()):.2e} mean {float(ea.mean()):+.1e} | "
          f"U^T U rel err rms {float(eb.pow(2).mean().sqrt()):.2e} max {float(eb.abs().max()):.2e} mean {float(eb.mean()):+.1e}", flush=True)
    del X, W, Ut
    torch.cuda.empty_cache()
