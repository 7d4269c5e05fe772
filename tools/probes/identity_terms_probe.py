"""Which term of the Gram-identity cost's error estimate decides at config E (10^6 x 4000 rank 100), on a late-run iterate:
runs N iterations of the loop, then evaluates the identity on the final operands with each rounding figure alone, with the
fp32 and the fp64 Gram, next to the streaming kernel's cost.   python tools/probes/identity_terms_probe.py [iterations=20]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402
from nn_fac_amd import nmf as nmf_mod  # noqa: E402

its = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
eng = get_engine(dev)
m, n, r = 1000000, 4000, 100
parts = [bench.synth_nmf_block_device(m // 8, n, r, b, 977, dev, torch) for b in range(8)]
X = torch.cat([p[0] for p in parts])
Ut = torch.cat([p[1] for p in parts]).t().contiguous()
del parts
V = torch.rand(r, n, device=dev, generator=torch.Generator(device=dev).manual_seed(4242))
ws = nmf_mod._StepBuffers(X, r)
log = []


def retired(it, cost, sw):
    log.append((it, float(cost), ws.direct_cost))
    return False


retired.revise_last = lambda c: None
Ut, V = nmf_mod.run_steps(eng, ws, X, r, Ut, V, its, "hals", 2, [None, None], [], [False, False], True, retired)
print("calibration (sigma_a, bias_a, sigma_g):", ws.cross_rounding, "| left the identity:", ws.direct_cost,
      "at iteration", next((i for i, _, d in log if d), None))
UtM = eng.xty(X, Ut)
G64 = torch.empty((r, r), dtype=torch.float64, device=dev)
UtU = eng.gram(Ut, out64=G64)
nx2 = eng.dot(X, X)
direct = float(eng.frob_resid(X, Ut, V))
print(f"direct cost {direct:.6e}   ||X||^2 {float(nx2):.6e}   bound 5e-4 * cost = {5e-4 * direct:.4e}")
o = torch.zeros(3, dtype=torch.float64, device=dev)
sa, ba, sg = ws.cross_rounding
for name, rd, g in (("fp32 Gram, all terms", (sa, ba), None), ("fp64 Gram, all terms", (sa, ba, sg), G64),
                    ("fp64 Gram, sigma_a only", (sa, 0.0, 0.0), G64), ("fp64 Gram, bias_a only", (0.0, ba, 0.0), G64),
                    ("fp64 Gram, sigma_g only", (0.0, 0.0, sg), G64), ("fp32 Gram, sigma_B only", (0.0, 0.0), None)):
    eng.gram_cost(V, UtM, UtU, nx2, o, rounding=rd, UtU64=g)
    c, f, e = o.cpu().tolist()
    print(f"{name:28s} cost {c:.6e}  (identity - direct)/direct = {(c - direct) / direct:+.3e}   estimate {e:.4e}   flag {int(f)}")
