"""Generate tests/golden/g10_deep_nmf.npz by running the REAL reference's deep_KL_NMF, one_step_deep_KL_nmf and deep_KL_mu
(ax-le/nn-fac @ /root/reference: nn_fac/deep_nmf.py:13-113, nn_fac/update_rules/deep_mu.py:8-14; scipy.special.lambertw).

TEST INFRASTRUCTURE ONLY; build container only.  Uses gen_golden.py's in-memory tensorly stand-in (this path never calls a
tensorly function).  The reference has no test of its own for this driver: the fixture is "outputs of the reference itself
run here"; the oracle restatement is asserted equal at generation time.

Usage:  python oracle/gen_golden_g10.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def main():
    if not os.path.isdir(gg.REF):
        raise SystemExit("gen_golden_g10.py needs /root/reference (build container only)")
    orc = gg._install_tensorly_standin()
    sys.path.insert(0, gg.REF)
    import nn_fac.deep_nmf as ref_deep
    import nn_fac.update_rules.deep_mu as ref_dmu
    rng = np.random.RandomState(7)
    g = {}
    # ---- deep_KL_mu alone: strictly positive operands, lambda over three decades (small lambda = large exp(a/lambda))
    m, n, r = 40, 30, 6
    W_Lm1 = rng.rand(m, n) + 0.05
    W_L, H_L = rng.rand(m, r) + 0.05, rng.rand(r, n) + 0.05
    WHn = rng.rand(m, r) + 0.05
    g["mu_W_Lm1"], g["mu_W_L"], g["mu_H_L"], g["mu_WHn"] = W_Lm1, W_L, H_L, WHn
    # lambda = 0.02: a / lambda ~ 750 > log(DBL_MAX), np.exp overflows to inf in the reference, lambertw(inf) = inf and the
    # update collapses to its 1e-12 floor -- part of the reference's observable behaviour, kept in the fixture
    lams = np.array([0.02, 0.3, 1.0, 7.5, 60.0])
    g["mu_lambdas"] = lams
    for i, lam in enumerate(lams):
        out = ref_dmu.deep_KL_mu(W_Lm1, W_L.copy(), H_L, WHn, lam)
        assert np.allclose(out, orc.deep_KL_mu(W_Lm1, W_L.copy(), H_L, WHn, lam), rtol=1e-12, atol=0)
        g[f"mu_out{i}"] = out
    # ---- one step from a custom start (three layers)
    data = rng.rand(50, 70) @ np.diag(rng.rand(70) + 0.2) + 0.05
    ranks = [10, 6, 3]
    W0 = [rng.rand(50, 10) + 0.1, rng.rand(50, 6) + 0.1, rng.rand(50, 3) + 0.1]
    H0 = [rng.rand(10, 70) + 0.1, rng.rand(6, 10) + 0.1, rng.rand(3, 6) + 0.1]
    lam = np.array([1.0, 0.6, 2.5])
    Wr, Hr, er = ref_deep.one_step_deep_KL_nmf(data, [w.copy() for w in W0], [h.copy() for h in H0], list(ranks), lam, 1e-6)
    Wo, Ho, eo = orc.one_step_deep_KL_nmf(data, [w.copy() for w in W0], [h.copy() for h in H0], list(ranks), lam, 1e-6)
    for a, b in zip(Wr + Hr, Wo + Ho):
        assert np.allclose(a, b, rtol=1e-11, atol=1e-14)
    assert np.allclose(er, eo, rtol=1e-11)
    g["step_data"], g["step_ranks"], g["step_lambda"] = data, np.array(ranks), lam
    for i in range(3):
        g[f"step_W0_{i}"], g[f"step_H0_{i}"], g[f"step_W_{i}"], g[f"step_H_{i}"] = W0[i], H0[i], Wr[i], Hr[i]
    g["step_errors"] = np.array(er)
    # ---- the whole driver from the multilayer (NNDSVD) start.  (init="custom" cannot be run: deep_nmf.py:46 assigns a
    # one-element LIST to an array element, which NumPy >= 1.24 refuses; the oracle / product accept the scalar.)
    Wr, Hr, rec, _ = ref_deep.deep_KL_NMF(data, list(ranks), n_iter_max_each_nmf=6, n_iter_max_deep_loop=6, tol=0,
                                          return_errors=True, deterministic=True, seed=3)
    Wo, Ho, reco = orc.deep_KL_NMF(data, list(ranks), n_iter_max_each_nmf=6, n_iter_max_deep_loop=6, tol=0,
                                   deterministic=True, seed=3)
    for a, b in zip(Wr + Hr, Wo + Ho):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12)
    assert np.allclose(rec, reco, rtol=1e-9, equal_nan=True)
    for i in range(3):
        g[f"ml_W_{i}"], g[f"ml_H_{i}"] = Wr[i], Hr[i]
    g["ml_errors"] = rec
    path = os.path.join(gg.OUT, "g10_deep_nmf.npz")
    np.savez_compressed(path, **g)
    print(f"G10 ok: {os.path.getsize(path)/1e3:.0f} kB")


if __name__ == "__main__":
    main()
