"""Randomised check of the stopping decision (nmf.py:320, ntf.py:337) under the Gram-identity cost: for random NMF / NTF problems
and a `tol` placed between two consecutive cost differences of a pilot run, the HALS loop must stop at the same iteration with
the identity cost (default: identity until two costs come within `tol` +- their error estimates, then the pass over the data)
as with NNF_COST=direct (every cost by the pass over the data), with bitwise equal factors, and -- reported, not required: the
fp32 direct cost has its own rounding -- at the iteration the fp64 oracle stops at.  Test infrastructure: imports oracle/.
    python tools/stress_stop.py [seed] [cases]"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.nmf import compute_nmf
from nn_fac_amd.ntf import compute_ntf
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.RandomState(seed)
bad = notes = 0
for c in range(cases):
    kind = "nmf" if rng.rand() < 0.5 else "ntf"
    iters = 14
    k = int(rng.randint(3, 10))
    os.environ.pop("NNF_COST", None)
    if kind == "nmf":
        m, n, r = int(rng.choice([60, 300, 2000, 9000])), int(rng.choice([20, 50, 257])), int(rng.choice([2, 5, 9, 17]))
        X, U0, V0 = orc.synth_nmf(m, n, r, seed=int(rng.randint(1 << 30)), dtype=np.float32)
        kw = dict(update_rule="hals", return_costs=True, deterministic=True)
        run = lambda tol: compute_nmf(X, r, U0, V0, n_iter_max=iters, tol=tol, **kw)
        ora = lambda tol: orc.compute_nmf(X.astype(np.float64), r, U0.astype(np.float64), V0.astype(np.float64), n_iter_max=iters, tol=tol, **kw)
        costs_of = lambda out: out[2]
        facs_of = lambda out: [out[0], out[1]]
        desc = (kind, m, n, r)
    else:
        shape = tuple(int(x) for x in rng.choice([8, 15, 30, 44], size=3))
        R = int(rng.choice([2, 4, 7]))
        T, F0 = orc.synth_ntf(shape, R, seed=int(rng.randint(1 << 30)), dtype=np.float32)
        kw = dict(update_rule="hals", alpha=math.inf, return_costs=True)
        run = lambda tol: compute_ntf(T, R, F0, n_iter_max=iters, tol=tol, sparsity_coefficients=[None] * 3, normalize=[False] * 3, **kw)
        ora = lambda tol: orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], n_iter_max=iters, tol=tol, **kw)
        costs_of = lambda out: out[1]
        facs_of = lambda out: list(out[0])
        desc = (kind, shape, R)
    pilot = costs_of(run(0))
    d = [abs(pilot[i - 1] - pilot[i]) for i in range(1, len(pilot))]
    if not all(np.isfinite(d)) or d[k - 1] == d[k]:
        continue
    w = float(rng.choice([0.5, 0.1, 0.9, 0.999]))          # where between the two differences the threshold sits
    tol = w * d[k - 1] + (1 - w) * d[k]
    a = run(tol)
    os.environ["NNF_COST"] = "direct"
    b = run(tol)
    os.environ.pop("NNF_COST", None)
    o = ora(tol)
    ca, cb, co = costs_of(a), costs_of(b), costs_of(o)
    same = len(ca) == len(cb) and all(np.array_equal(x, y) for x, y in zip(facs_of(a), facs_of(b)))
    tail = len(ca) < iters and ca[-2:] == cb[-2:] if len(ca) >= 2 else True
    if not same or (len(ca) < iters and not tail):
        bad += 1
        print("CASE", c, desc, f"tol {tol:.3e}: identity run {len(ca)} iterations, direct run {len(cb)}, oracle {len(co)}; last costs", ca[-2:], cb[-2:], flush=True)
    elif len(ca) != len(co):
        notes += 1
        i = min(len(ca), len(co)) - 1
        print("NOTE", c, desc, f"tol {tol:.3e}: device (both cost forms) stops after {len(ca)} iterations, fp64 oracle after {len(co)}; "
              f"|dcost| there: device {abs(cb[i - 1] - cb[i]):.6e}, oracle {abs(co[i - 1] - co[i]):.6e}", flush=True)
print(f"stress_stop seed {seed}: {cases} cases, {bad} flagged, {notes} where the fp32 run and the fp64 oracle stop apart")
