"""Generate tests/golden/g8_hals_coupling.npz by running the REAL reference's hals_coupling_nnls_acc
(ax-le/nn-fac @ /root/reference, nn_fac/update_rules/nnls.py:204-352; the module imports with NumPy alone).

TEST INFRASTRUCTURE ONLY; build container only (the reference never travels, the .npz does).
The reference has no test of its own for this function (its header flags the block as sandbox code), so the fixture is
"outputs of the reference itself run here"; the oracle restatement is asserted equal at generation time.

Usage:  python oracle/gen_golden_g8.py
"""
import math
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden_g8.py needs /root/reference (build container only)")
    sys.path.insert(0, HERE)
    sys.path.insert(0, REF)
    import nnfac_oracle as orc
    import nn_fac.update_rules.nnls as ref_nnls
    rng = np.random.RandomState(808)
    g = {}
    case = 0
    shapes = [(3, 1, 40), (10, 100, 300), (50, 300, 900), (64, 97, 400), (100, 70, 300)]   # (r, n, m behind the Gram)
    for si, (r, n, m) in enumerate(shapes):
        U = rng.rand(m, r)
        M = U @ rng.rand(r, n) + 1e-2 * rng.rand(m, n)
        UtU, UtM = U.T @ U, U.T @ M
        Vin, Vt = rng.rand(r, n), rng.rand(r, n)
        g[f"s{si}_UtM"], g[f"s{si}_UtU"], g[f"s{si}_Vin"], g[f"s{si}_Vt"] = UtM, UtU, Vin, Vt
        variants = [dict(mu=0.5), dict(mu=25.0, maxiter=100), dict(mu=0.0, delta=0.0, maxiter=7)]
        if si <= 2:
            variants += [dict(mu=3.0, normalize=True, maxiter=20), dict(mu=1e3, maxiter=2), dict(mu=0.7, maxiter=1)]
        if si == 1:
            variants += [dict(mu=2.0, nonzero=True, maxiter=30), dict(mu=2.0, zero_diag=4, maxiter=30)]
        for kw in variants:
            kw = dict(kw)
            G = UtU
            zd = kw.pop("zero_diag", None)
            if zd is not None:                       # a zero Gram diagonal: the row is skipped even though mu > 0
                G = UtU.copy()
                G[zd, zd] = 0.0
            mu = kw.pop("mu")
            Vr, eps, cnt, _ = ref_nnls.hals_coupling_nnls_acc(UtM, G, Vin, Vt, mu, alpha=math.inf, **kw)
            log = []
            Vo, eps_o, cnt_o, _ = orc.hals_coupling_nnls_acc(UtM, G, Vin, Vt, mu, alpha=math.inf, sweep_log=log, **kw)
            assert cnt == cnt_o and np.allclose(Vr, Vo, rtol=1e-12, atol=1e-15) and np.isclose(eps, eps_o, rtol=1e-12)
            p = f"c{case}_"
            g[p + "shape"], g[p + "V"], g[p + "eps"], g[p + "cnt"] = np.int64(si), Vr, np.float64(eps), np.int64(cnt)
            g[p + "nodelta"] = np.array(log)
            g[p + "kw"] = np.array([mu, kw.get("maxiter", 500), kw.get("delta", 0.01), float(kw.get("normalize", False)),
                                    float(kw.get("nonzero", False)), -1.0 if zd is None else float(zd)])
            case += 1
    g["ncases"] = np.int64(case)
    path = os.path.join(OUT, "g8_hals_coupling.npz")
    np.savez_compressed(path, **g)
    print(f"G8 ok: {case} cases, {os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    main()
