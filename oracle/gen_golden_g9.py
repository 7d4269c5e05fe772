"""Generate tests/golden/g9_multilayer.npz by running the REAL reference's multilayer_beta_NMF
(ax-le/nn-fac @ /root/reference, nn_fac/multilayer_nmf.py:7-51; NNDSVD start values, MU layers, normalize_WH).

TEST INFRASTRUCTURE ONLY; build container only.  Uses gen_golden.py's in-memory tensorly stand-in (the NMF path never calls
a tensorly function).  The reference has no test of its own for this driver: the fixture is "outputs of the reference
itself run here"; the oracle restatement is asserted equal at generation time.

Usage:  python oracle/gen_golden_g9.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def main():
    if not os.path.isdir(gg.REF):
        raise SystemExit("gen_golden_g9.py needs /root/reference (build container only)")
    orc = gg._install_tensorly_standin()
    sys.path.insert(0, gg.REF)
    import nn_fac.multilayer_nmf as ref_ml
    rng = np.random.RandomState(42)
    data = rng.rand(60, 80) @ np.diag(rng.rand(80)) + 0.05
    ranks = [12, 8, 4]
    g = {"data": data, "ranks": np.array(ranks), "n_iter": np.int64(6), "seed": np.int64(3)}
    for beta in (1, 2, 0.5):
        Wr, Hr, er, _ = ref_ml.multilayer_beta_NMF(data.copy(), list(ranks), beta=beta, n_iter_max_each_nmf=6,
                                                   return_errors=True, deterministic=True, seed=3)
        Wo, Ho, eo = orc.multilayer_beta_NMF(data.copy(), list(ranks), beta=beta, n_iter_max_each_nmf=6, deterministic=True,
                                             seed=3)
        for a, b in zip(Wr + Hr, Wo + Ho):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), beta
        assert np.allclose(er, eo, rtol=1e-10)
        for i in range(len(ranks)):
            g[f"b{beta}_W{i}"], g[f"b{beta}_H{i}"] = Wr[i], Hr[i]
        g[f"b{beta}_errors"] = er
    path = os.path.join(gg.OUT, "g9_multilayer.npz")
    np.savez_compressed(path, **g)
    print(f"G9 ok: {os.path.getsize(path)/1e3:.0f} kB")


if __name__ == "__main__":
    main()
