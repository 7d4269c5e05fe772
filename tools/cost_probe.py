"""Times nnf_frob_resid_f32 / nnf_xht_f32 / nnf_xty_f32 at config B (run once per NNF_COST_CSPLIT setting)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine
m, n, r = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100000, 2000, 50)
eng = get_engine()
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand(m, n, device="cuda", generator=g)
Ut = torch.rand(r, m, device="cuda", generator=g)
V = torch.rand(r, n, device="cuda", generator=g)
out = torch.empty(1, dtype=torch.float64, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("csplit", os.environ.get("NNF_COST_CSPLIT", "default"),
      "frob %.1f us" % t(lambda: eng.frob_resid(X, Ut, V, out=out)),
      "kl %.1f us" % t(lambda: eng.betadiv(X, Ut, V, 1.0, out=out)),
      "xht %.1f us" % t(lambda: eng.xht(X, V)), "xty %.1f us" % t(lambda: eng.xty(X, Ut)), flush=True)
