"""Why does stress_parity.py seed 81, case 43 (128 x 130, rank 128, HALS with sparsity) stop its 7th inner solve after 39
sweeps on the device and after 38 in the fp64 oracle?  Prints eps / eps0 (the quantity nnls.py:156 compares with delta = 0.01)
sweep by sweep around the stopping sweep for
  (a) the device kernel (fp32) on the device's own operands of that solve,
  (b) the fp64 oracle on THE SAME operands (the device's fp32 cross product, Gram and start factor, cast to fp64),
  (c) the fp64 oracle on its own trajectory (what the parity test compares with).
(a) vs (b) isolates the sweep arithmetic, (b) vs (c) the three outer iterations of fp32 history in the operands.
Test infrastructure: imports oracle/.   python tools/probes/seed81_probe.py [seed] [case] [solve index]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.nmf import compute_nmf
from nn_fac_amd import engine as _engine

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 81
want = int(sys.argv[2]) if len(sys.argv) > 2 else 43
solve = int(sys.argv[3]) if len(sys.argv) > 3 else 6          # 0-based index in the sweep log (U, V, U, V, ...)
rng = np.random.RandomState(seed)
for c in range(want + 1):                                      # the generator of tools/stress_parity.py, draw for draw
    r = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 32, 33, 48, 50, 63, 64, 65, 100, 127, 128]))
    m = int(rng.choice([r, r + 1, 64, 97, 255, 256, 257, 700, 1500]))
    n = int(rng.choice([r, r + 3, 16, 61, 64, 130, 257, 600]))
    m, n = max(m, r), max(n, r)
    rule, beta = [("hals", 2), ("mu", 1), ("mu", 2), ("mu", 0.5), ("mu", 0), ("mu", 3)][rng.randint(6)]
    if rule == "mu" and r > 64 and beta != 2 and m * n > 300000:
        continue
    X = (rng.rand(m, r) @ rng.rand(r, n) + 1e-2 * rng.rand(m, n)).astype(np.float32)
    if rng.rand() < 0.3 and rule == "hals":
        X[rng.rand(m, n) < 0.3] = 0.0
    U0, V0 = rng.rand(m, r).astype(np.float32) + 0.01, rng.rand(r, n).astype(np.float32) + 0.01
    sp = [None, None] if rule == "mu" or rng.rand() < 0.6 else [float(rng.rand() * 0.1), float(rng.rand() * 0.1)]
    nz = [False, bool(rng.rand() < 0.3)] if rule == "hals" else [False, False]
print("case", want, (m, n, r, rule, beta, sp, nz))
assert rule == "hals" and not nz[1]
outer, side = divmod(solve, 2)
kw = dict(tol=0, update_rule=rule, beta=beta, sparsity_coefficients=sp, normalize=nz, return_costs=True, deterministic=True)
dev = torch.device("cuda:0")
eng = _engine.get_engine(dev)
Xd = torch.from_numpy(X).to(dev)
X64 = X.astype(np.float64)
U, V = U0, V0
Uo, Vo = U0.astype(np.float64), V0.astype(np.float64)
if outer:
    U, V, _, _ = compute_nmf(X, r, U0, V0, n_iter_max=outer, **kw)
    Uo, Vo, _, _ = orc.compute_nmf(X64, r, Uo, Vo, n_iter_max=outer, **kw)
print(f"state after {outer} outer iterations: relU {np.linalg.norm(U - Uo) / np.linalg.norm(Uo):.2e} "
      f"relV {np.linalg.norm(V - Vo) / np.linalg.norm(Vo):.2e}")
Ut, Vd = torch.from_numpy(np.ascontiguousarray(U.T)).to(dev), torch.from_numpy(np.ascontiguousarray(V)).to(dev)
if side == 1:      # the V-side solve of that outer iteration: first its U-side update on both trajectories
    raise SystemExit("V-side solves: not needed for this case")
G, Cx = eng.gram(Vd), eng.xht(Xd, Vd)                         # VVt, VMt on the device (fp32)
NS = 48
nd32 = eng.hals_sweeps(Cx, G, Ut.clone(), NS, sparsity=sp[0]).cpu().numpy()
log_b, log_c = [], []
orc.hals_nnls_acc(Cx.cpu().numpy().astype(np.float64), G.cpu().numpy().astype(np.float64), U.T.astype(np.float64), maxiter=NS,
                  alpha=np.inf, delta=0.0, sparsity_coefficient=sp[0], sweep_log=log_b)
orc.hals_nnls_acc(Vo @ X64.T, Vo @ Vo.T, Uo.T.copy(), maxiter=NS, alpha=np.inf, delta=0.0, sparsity_coefficient=sp[0],
                  sweep_log=log_c)
print("sweep   (a) device fp32     (b) fp64, same operands   (c) fp64 oracle trajectory      [eps / eps0; stop when < 0.01]")
for s in range(NS):
    a, b, c_ = nd32[s] / nd32[0], log_b[s] / log_b[0], log_c[s] / log_c[0]
    mark = "".join(ch if v < 0.01 else "." for ch, v in zip("abc", (a, b, c_)))
    if s < 3 or min(a, b, c_) < 0.02:
        print(f"{s + 1:4d}   {a:.9f}         {b:.9f}               {c_:.9f}      {mark}")
first = [next((s + 1 for s, v in enumerate(x) if v / x[0] < 0.01), None) for x in (nd32, log_b, log_c)]
print("first sweep below delta:", dict(zip("abc", first)), " eps0:", float(nd32[0]), log_b[0], log_c[0])
