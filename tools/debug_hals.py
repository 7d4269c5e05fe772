import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd.update_rules.nnls import hals_nnls_acc
from nn_fac_amd.engine import get_engine
g = np.load(os.path.join(ROOT, "tests/golden/g1_hals.npz"))
np.set_printoptions(precision=6, linewidth=200)
def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)
bad = 0
for c in range(int(g["ncases"])):
    s = int(g[f"c{c}_shape"]); kwv = g[f"c{c}_kw"]
    kw = dict(maxiter=int(kwv[0]), delta=float(kwv[1]), alpha=math.inf, normalize=bool(kwv[3]), nonzero=bool(kwv[4]))
    if kwv[2] >= 0: kw["sparsity_coefficient"] = float(kwv[2])
    V, eps, cnt, rho = hals_nnls_acc(g[f"s{s}_UtM"], g[f"s{s}_UtU"], g[f"s{s}_Vin"], **kw)
    e = rel(V, g[f"c{c}_V"])
    flag = "BAD" if (e > 2e-4 or cnt != int(g[f"c{c}_cnt"])) else "ok "
    bad += flag == "BAD"
    print(flag, c, g[f"s{s}_Vin"].shape, kw, "rel=%.2e cnt=%d/%d eps=%.4e/%.4e" % (e, cnt, int(g[f"c{c}_cnt"]), eps, float(g[f"c{c}_eps"])))
    if flag == "BAD" and V.size <= 16:
        print("   got ", V.ravel()); print("   want", g[f"c{c}_V"].ravel()); print("   Vin ", g[f"s{s}_Vin"].ravel())
        print("   UtM ", g[f"s{s}_UtM"].ravel()); print("   UtU ", g[f"s{s}_UtU"].ravel())
print("bad cases:", bad)
