"""The CPU oracle against the fixtures produced by the real reference (oracle/gen_golden.py).  No GPU needed."""
import math

import numpy as np
import pytest

import nnfac_oracle as orc


def _kw(vec):
    kw = dict(maxiter=int(vec[0]), delta=float(vec[1]), alpha=math.inf)
    if vec[2] >= 0:
        kw["sparsity_coefficient"] = float(vec[2])
    kw["normalize"], kw["nonzero"] = bool(vec[3]), bool(vec[4])
    return kw


def test_g0_reference_known_answers(golden):
    g = golden("g0_known_answers.npz")
    data, rank = g["data"], int(g["rank"])
    assert abs(data[0][0] - 2.143518599859098) < 1e-7            # reference tests/NMF_tests.py:68
    for rule, beta, seed in (("hals", 2, 0), ("mu", 2, 82), ("mu", 1, 82), ("mu", 0, 82)):
        tag = f"{rule}_b{beta}_s{seed}"
        U, V, costs, _ = orc.nmf(data, rank, init="random", n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                 return_costs=True, deterministic=True, seed=seed)
        u00, v00, c0, c1 = g[f"known_{tag}"]                    # literals of tests/NMF_tests.py:76-81,...,130-135
        assert abs(U[0][0] - u00) < 1e-7 and abs(V[0][0] - v00) < 1e-7
        assert abs(costs[0] - c0) < 1e-7 and abs(costs[-1] - c1) < 1e-7
        np.testing.assert_allclose(U, g[f"U_{tag}"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(V, g[f"V_{tag}"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(costs, g[f"costs_{tag}"], rtol=1e-12)


def test_g0_random_init_stream():
    U, V = orc.nmf_random_init((73, 25), 9, 0)
    assert abs(U[0][0] - 0.5488135) < 1e-7 and abs(V[0][0] - 1.15834001e-01) < 1e-7   # NMF_tests.py:40-41


def test_g0_nndsvd_known_answer(golden):
    """tests/NMF_tests.py:33-36 of the reference: NNDSVD start values of the 73 x 25 rank-9 problem."""
    g = golden("g0_known_answers.npz")
    U, V = orc.nndsvd(g["data"], int(g["rank"]))
    assert abs(U[0][0] - 1.4604530858567824) < 1e-7 and abs(V[0][0] - 1.3118383377996725) < 1e-7
    assert U.shape == (73, 9) and V.shape == (9, 25) and U.min() >= 1e-12 and V.min() >= 1e-12
    # sign convention of the SVD does not matter
    Uf, Vf = orc.nndsvd(g["data"][::-1, ::-1].copy(), int(g["rank"]))
    np.testing.assert_allclose(Uf[::-1], U, rtol=1e-9, atol=1e-12)


def test_g1_hals(golden):
    g = golden("g1_hals.npz")
    for c in range(int(g["ncases"])):
        s = int(g[f"c{c}_shape"])
        log = []
        V, eps, cnt, _ = orc.hals_nnls_acc(g[f"s{s}_UtM"], g[f"s{s}_UtU"], g[f"s{s}_Vin"], sweep_log=log,
                                           **_kw(g[f"c{c}_kw"]))
        assert cnt == int(g[f"c{c}_cnt"]), c
        np.testing.assert_allclose(V, g[f"c{c}_V"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(eps, float(g[f"c{c}_eps"]), rtol=1e-12)
        np.testing.assert_allclose(log, g[f"c{c}_nodelta"], rtol=1e-12)


def _kw8(vec):
    kw = dict(maxiter=int(vec[1]), delta=float(vec[2]), normalize=bool(vec[3]), nonzero=bool(vec[4]), alpha=math.inf)
    return float(vec[0]), kw, (None if vec[5] < 0 else int(vec[5]))


def test_g8_hals_coupling(golden):
    """nnls.py:204-352 (coupled sweep used by PARAFAC2): outputs of the real reference, oracle/gen_golden_g8.py."""
    g = golden("g8_hals_coupling.npz")
    for c in range(int(g["ncases"])):
        s = int(g[f"c{c}_shape"])
        mu, kw, zd = _kw8(g[f"c{c}_kw"])
        G = g[f"s{s}_UtU"].copy()
        if zd is not None:
            G[zd, zd] = 0.0
        log = []
        V, eps, cnt, _ = orc.hals_coupling_nnls_acc(g[f"s{s}_UtM"], G, g[f"s{s}_Vin"], g[f"s{s}_Vt"], mu,
                                                    sweep_log=log, **kw)
        assert cnt == int(g[f"c{c}_cnt"]), c
        np.testing.assert_allclose(V, g[f"c{c}_V"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(eps, float(g[f"c{c}_eps"]), rtol=1e-12)
        np.testing.assert_allclose(log, g[f"c{c}_nodelta"], rtol=1e-12)
    r = np.random.RandomState(0)
    G = r.rand(8, 8)
    G[2, 2] = 0
    with pytest.raises(ValueError):       # nnls.py:331-332: a plain ValueError, not ZeroColumnWhenUnautorized
        orc.hals_coupling_nnls_acc(r.rand(8, 8), G, r.rand(8, 8), r.rand(8, 8), 1.0, nonzero=True)


def test_g9_multilayer(golden):
    """multilayer_nmf.py:7-51 (NNDSVD start, MU layers, normalize_WH): outputs of the real reference, oracle/gen_golden_g9.py."""
    g = golden("g9_multilayer.npz")
    ranks = [int(x) for x in g["ranks"]]
    for beta in (1, 2, 0.5):
        W, H, errors = orc.multilayer_beta_NMF(g["data"].copy(), list(ranks), beta=beta, n_iter_max_each_nmf=int(g["n_iter"]),
                                               deterministic=True, seed=int(g["seed"]))
        for i in range(len(ranks)):
            np.testing.assert_allclose(W[i], g[f"b{beta}_W{i}"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(H[i], g[f"b{beta}_H{i}"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(H[i].sum(axis=1), 1.0, rtol=1e-12)       # normalize_WH(., "H")
        np.testing.assert_allclose(errors, g[f"b{beta}_errors"], rtol=1e-10)
    with pytest.raises(ValueError):
        orc.multilayer_beta_NMF(g["data"], [4, 8], n_iter_max_each_nmf=2)


def test_hals_argument_errors():
    r = np.random.RandomState(0)
    with pytest.raises(orc.ArgumentException):
        orc.hals_nnls_acc(r.rand(8, 8), r.rand(8, 8), np.array([]))
    with pytest.raises(orc.ArgumentException):
        orc.hals_nnls_acc(r.rand(8), r.rand(8, 8), r.rand(8, 8))
    G = r.rand(8, 8)
    G[2, 2] = 0
    orc.hals_nnls_acc(r.rand(8, 8), G, r.rand(8, 8))
    with pytest.raises(orc.ZeroColumnWhenUnautorized):
        orc.hals_nnls_acc(r.rand(8, 8), G, r.rand(8, 8), nonzero=True)


def test_g2_mu_and_divergence(golden):
    g = golden("g2_mu.npz")
    U, V, M = g["U"], g["V"], g["M"]
    for b in (0, 0.5, 1, 1.5, 2, 3, 4):
        b = int(b) if float(b).is_integer() else b
        np.testing.assert_allclose(orc.switch_alternate_mu(M, U, V, b, "U"), g[f"muU_b{b}"], rtol=1e-13)
        np.testing.assert_allclose(orc.switch_alternate_mu(M, U, V, b, "V"), g[f"muV_b{b}"], rtol=1e-13)
        np.testing.assert_allclose(orc.beta_divergence(M, U @ V, b), float(g[f"div_b{b}"]), rtol=1e-13)
        assert orc.gamma_beta(b) == float(g[f"gamma_b{b}"])
    with pytest.raises(orc.InvalidArgumentValue):
        orc.mu_betadivmin(U, V, M, -1)
    with pytest.raises(orc.InvalidArgumentValue):
        orc.switch_alternate_mu(M, U, V, 1, "X")


def test_g4_config_a(golden):
    g = golden("g4_nmf_configA.npz")
    for dt, tag, tol in ((np.float64, "f64", 1e-12), (np.float32, "f32", 2e-5)):
        X, U0, V0 = g["X"].astype(dt), g["U0"].astype(dt), g["V0"].astype(dt)
        for rule, beta in (("hals", 2), ("mu", 2), ("mu", 1), ("mu", 0), ("mu", 1.5), ("mu", 3)):
            sw = []
            U, V, costs, _ = orc.nmf(X, 10, init="custom", U_0=U0, V_0=V0, n_iter_max=10, tol=0, update_rule=rule,
                                     beta=beta, return_costs=True, deterministic=True, sweeps=sw)
            k = f"{rule}_b{beta}_{tag}"
            np.testing.assert_allclose(U, g[f"U_{k}"], rtol=tol, atol=tol)
            np.testing.assert_allclose(V, g[f"V_{k}"], rtol=tol, atol=tol)
            np.testing.assert_allclose(costs, g[f"costs_{k}"], rtol=max(tol, 1e-12))
            assert list(sw) == list(g[f"sweeps_{k}"])
    X, U0, V0 = g["X"], g["U0"], g["V0"]
    U, V, costs, _ = orc.nmf(X, 10, init="custom", U_0=U0, V_0=V0, n_iter_max=6, tol=0, update_rule="hals",
                             sparsity_coefficients=[0.05, 0.1], normalize=[False, True], return_costs=True,
                             deterministic=True)
    np.testing.assert_allclose(U, g["U_hals_sparse_norm"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(costs, g["costs_hals_sparse_norm"], rtol=1e-11)
    U, V, costs, _ = orc.nmf(X, 10, init="custom", U_0=U0, V_0=V0, n_iter_max=4, tol=0, update_rule="hals",
                             fixed_modes=[0], return_costs=True, deterministic=True)
    np.testing.assert_allclose(U, U0)            # mode 0 fixed: U untouched
    np.testing.assert_allclose(V, g["V_hals_fixed0"], rtol=1e-11, atol=1e-13)


def test_g6_ntf(golden):
    g = golden("g6_ntf.npz")
    for name in ("small", "cube", "ragged"):
        T = g[f"{name}_T"]
        F0 = [g[f"{name}_F0_{i}"] for i in range(3)]
        R = F0[0].shape[1]
        unf = [orc.unfold(T, m) for m in range(3)]
        nrm = np.sqrt(np.sum(T ** 2))
        # unfold/khatri_rao convention: unfold(cp(F), mode) == F[mode] @ khatri_rao(F, skip=mode).T
        cp = np.einsum('ir,jr,kr->ijk', *F0)
        for mode in range(3):
            np.testing.assert_allclose(orc.unfold(cp, mode), F0[mode] @ orc.khatri_rao(F0, skip_matrix=mode).T,
                                       rtol=1e-12)
        for rule, beta in (("hals", 2), ("mu", 2), ("mu", 1)):
            f = [x.copy() for x in F0]
            costs = []
            for _ in range(5):
                f, c = orc.one_ntf_step(unf, R, f, nrm, rule, beta, [None] * 3, [], [False] * 3, alpha=math.inf,
                                        delta=0.01)
                costs.append(c)
            for i in range(3):
                np.testing.assert_allclose(f[i], g[f"{name}_{rule}_b{beta}_F{i}"], rtol=1e-10, atol=1e-13)
            np.testing.assert_allclose(costs, g[f"{name}_{rule}_b{beta}_costs"], rtol=1e-10)


def ntd_reference_tensor(shape, ranks):
    """The tensor of tests/NTD_tests.py:18-27 (setUp), rebuilt from the seeds (nothing of it is stored)."""
    import random
    np.random.seed(0)
    random.seed(0)
    assert (random.randint(3, 10), random.randint(3, 10), random.randint(3, 10)) == tuple(ranks)
    assert (random.randint(20, 100), random.randint(20, 100), random.randint(20, 100)) == tuple(shape)
    for mo in range(3):
        np.random.rand(shape[mo], ranks[mo])
    np.random.rand(*ranks)
    return np.abs(orc.random_tucker_full(shape, ranks, 0)) + 1e-2 * np.random.rand(*shape)


def test_g7_ntd_reference_known_answers(golden):
    """The reference's own NTD known-answer tests (NTD_tests.py:141-255) through the restatement."""
    g = golden("g7_ntd.npz")
    shape, ranks = tuple(int(x) for x in g["ref_shape"]), tuple(int(x) for x in g["ref_ranks"])
    T = ntd_reference_tensor(shape, ranks)
    assert abs(T[0][0][0] - 21.974433828159626) < 1e-7
    for rule, beta in (("hals", 2), ("mu", 2), ("mu", 1), ("mu", 0)):
        c0, f0 = orc.ntd_random_init(shape, list(ranks), 0)
        sw, pg = [], []
        core, facs, costs, _ = orc.compute_ntd(T, list(ranks), c0, f0, n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                               sparsity_coefficients=[None] * 4, fixed_modes=[], normalize=[False] * 4,
                                               return_costs=True, deterministic=True, sweeps=sw, pg_iters=pg)
        tag = f"ref_{rule}_b{beta}"
        np.testing.assert_allclose(core, g[f"{tag}_core"], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(costs, g[f"{tag}_costs"], rtol=1e-6, atol=1e-12)
        for i in range(3):
            np.testing.assert_allclose(facs[i], g[f"{tag}_F{i}"], rtol=1e-7, atol=1e-10)
        assert list(sw) == list(g[f"{tag}_sweeps"]) and list(pg) == list(g[f"{tag}_pg"])


def test_g7_ntd_small_steps(golden):
    g = golden("g7_ntd.npz")
    T, c0 = g["small_T"], g["small_core0"]
    f0 = [g[f"small_F0_{i}"] for i in range(3)]
    rk = list(c0.shape)
    nrm = np.sqrt(np.sum(T ** 2))
    cases = {"plain": ([None] * 4, [], [False] * 4, None), "sparse": ([0.01, None, 0.02, 0.05], [], [False] * 4, None),
             "norm": ([None] * 4, [], [True, False, True, True], 1), "fixed1": ([None] * 4, [1], [False] * 4, None)}
    for name, (sp, fixed, norm, mcn) in cases.items():
        core, f, costs = c0.copy(), [x.copy() for x in f0], []
        for _ in range(4):
            core, f, c = orc.one_ntd_step(T, rk, core, f, nrm, list(sp), list(fixed), list(norm), mcn, alpha=math.inf,
                                          delta=0.01)
            costs.append(c)
        np.testing.assert_allclose(core, g[f"small_hals_{name}_core"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(costs, g[f"small_hals_{name}_costs"], rtol=1e-8, atol=1e-13)
        for i in range(3):
            np.testing.assert_allclose(f[i], g[f"small_hals_{name}_F{i}"], rtol=1e-8, atol=1e-12)
    for beta in (0, 0.5, 1, 2, 3):
        core, f, costs = c0.copy(), [x.copy() for x in f0], []
        for _ in range(4):
            core, f, c = orc.one_ntd_step_mu(T, rk, core, f, beta, nrm, [], [False] * 4, None)
            costs.append(c)
        np.testing.assert_allclose(core, g[f"small_mu_b{beta}_core"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(costs, g[f"small_mu_b{beta}_costs"], rtol=1e-9)
        np.testing.assert_allclose(orc.mu_tensorial(c0, f0, T, beta), g[f"small_mut_b{beta}"], rtol=1e-12)


# ---- g10: deep KL-NMF (deep_nmf.py:13-113, deep_mu.py:8-14), outputs of the real reference ----
def test_oracle_deep_kl_mu_matches_reference(golden):
    import warnings
    g = golden("g10_deep_nmf.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # lambda = 0.02 overflows np.exp in the reference too (kept behaviour)
        for i, lam in enumerate(g["mu_lambdas"]):
            out = orc.deep_KL_mu(g["mu_W_Lm1"], g["mu_W_L"].copy(), g["mu_H_L"], g["mu_WHn"], lam)
            np.testing.assert_allclose(out, g[f"mu_out{i}"], rtol=1e-12, atol=0)


def test_oracle_deep_nmf_matches_reference(golden):
    g = golden("g10_deep_nmf.npz")
    ranks = [int(x) for x in g["step_ranks"]]
    W0 = [g[f"step_W0_{i}"].copy() for i in range(3)]
    H0 = [g[f"step_H0_{i}"].copy() for i in range(3)]
    W, H, e = orc.one_step_deep_KL_nmf(g["step_data"], W0, H0, ranks, g["step_lambda"], 1e-6)
    for i in range(3):
        np.testing.assert_allclose(W[i], g[f"step_W_{i}"], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(H[i], g[f"step_H_{i}"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(e, g["step_errors"], rtol=1e-11)
    W, H, rec = orc.deep_KL_NMF(g["step_data"], list(ranks), n_iter_max_each_nmf=6, n_iter_max_deep_loop=6, tol=0,
                                deterministic=True, seed=3)
    for i in range(3):
        np.testing.assert_allclose(W[i], g[f"ml_W_{i}"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(H[i], g[f"ml_H_{i}"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rec, g["ml_errors"], rtol=1e-9, equal_nan=True)


# ---- Tucker (HOSVD + HOOI) initialiser of NTD: pinned by the reference's own known answers for init="tucker" ----
NTD_TESTS_SHAPE, NTD_TESTS_RANKS = (53, 85, 82), (9, 9, 3)      # what NTD_tests.py:18-21 draws with random.seed(0)


TUCKER_INIT_KNOWN = {   # (factors[0][0][0], factors[1][0][0], factors[2][0][0], core[0,0,0], cost[0], cost[-1])
    ("hals", 2): (0.16504481330298995, 0.09847086272185894, 0.11680262111792158, 11039.862648258559,
                  0.00027083233922590056, 0.00010638116104305596),          # NTD_tests.py:168-175
    ("mu", 2): (0.1633567459395657, 0.09484478066313659, 0.1174295516693132, 11046.430317228587,
                22653.665491321422, 21679.048477120345),                    # NTD_tests.py:208-215
}


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 2)])
def test_tucker_init_known_answers(rule, beta):
    """ntd(init="tucker", n_iter_max=10, tol=1e-8, deterministic=True, seed=0) of the reference's tests (NTD_tests.py:157-175,
    197-215), through the restated tensorly tucker (oracle tucker_hooi) + the oracle's NTD loop: assertAlmostEqual precision."""
    T, ranks = ntd_reference_tensor(NTD_TESTS_SHAPE, NTD_TESTS_RANKS), list(NTD_TESTS_RANKS)
    c0, f0 = orc.ntd_tucker_init(T, ranks)
    core, F, costs, _ = orc.compute_ntd(T, ranks, c0, f0, n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                        sparsity_coefficients=[None] * 4, normalize=[False] * 4, return_costs=True,
                                        deterministic=True)
    got = (F[0][0][0], F[1][0][0], F[2][0][0], core[0, 0, 0], costs[0], costs[-1])
    for a, b in zip(got, TUCKER_INIT_KNOWN[(rule, beta)]):
        assert abs(a - b) <= 5e-8 * max(1.0, abs(b)), (got, TUCKER_INIT_KNOWN[(rule, beta)])
