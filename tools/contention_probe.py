"""What slows the V-side sweep kernel down when another kernel shares the chip?  The solve (50 x 2000, 100 sweeps) is timed
alone and next to a loop of each candidate on a second stream + context."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine, get_side_engine
eng = get_engine()
seng, sstream = get_side_engine("cuda:0", "cost")
g = torch.Generator(device="cuda").manual_seed(0)
m, n, r = 100000, 2000, 50
X = torch.rand(m, n, device="cuda", generator=g); Ut = torch.rand(r, m, device="cuda", generator=g); V = torch.rand(r, n, device="cuda", generator=g)
W = torch.rand(400, r, device="cuda", generator=g); G = (W.t() @ W).contiguous(); M = torch.rand(r, n, device="cuda", generator=g) * 100
out = torch.empty(1, dtype=torch.float64, device="cuda")
Y = torch.empty_like(X)
cands = {"alone": None, "cost (frob)": lambda: seng.frob_resid(X, Ut, V, out=out), "xht": lambda: seng.xht(X, V),
         "xty": lambda: seng.xty(X, Ut), "copy 800 MB": lambda: Y.copy_(X), "gram U": lambda: seng.gram(Ut)}
main = torch.cuda.current_stream()
for name, fn in cands.items():
    ts = []
    for rep in range(4):
        Vc = V.clone()
        torch.cuda.synchronize()
        if fn is not None:
            with torch.cuda.stream(sstream):
                for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        eng.hals_solve(M, G, Vc, 100, delta=0.0)
        b.record(main)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"{name:14s}: V-side solve of 100 sweeps {min(ts[1:]):7.1f} us ({min(ts[1:])/100:.2f} us/sweep)", flush=True)
