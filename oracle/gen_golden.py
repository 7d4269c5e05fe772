"""Generate tests/golden/*.npz by running the REAL reference (ax-le/nn-fac @ /root/reference).

TEST INFRASTRUCTURE ONLY.  Runs in the build container only: refuses to run when
/root/reference is absent (it never travels to the GPU box; the .npz fixtures do).

tensorly 0.6.0 (reference setup.py:30) is not installable offline.  The NMF path never
calls a tensorly function, so an in-memory module object is enough for it; for the NTF
path the handful of tensorly calls (unfold, khatri_rao, norm, dot, ...) are given their
published 0.6.0 semantics in NumPy.  Before any fixture is written
the reference's own known answers (tests/NMF_tests.py) are re-asserted through this set-up,
so a broken stand-in cannot poison the fixtures.

Usage:  python oracle/gen_golden.py
"""
import math
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_tensorly_standin():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import nnfac_oracle as orc  # only for the tensor helper semantics (unfold/khatri_rao/...)
    tl = types.ModuleType("tensorly")
    tl.tensor = lambda x, **kw: np.array(x)
    tl.ones, tl.abs, tl.dot, tl.transpose, tl.ndim, tl.conj = np.ones, np.abs, np.dot, np.transpose, np.ndim, np.conj
    tl.unfold = orc.unfold
    tl.fold = orc.fold
    tl.tensor_to_vec = lambda t: t.reshape(-1)

    def norm(t, order=2):
        return np.sqrt(np.sum(np.abs(t) ** 2)) if order == 2 else np.sum(np.abs(t))
    tl.norm = norm
    base = types.ModuleType("tensorly.base")
    base.unfold, base.fold = orc.unfold, orc.fold
    tenalg = types.ModuleType("tensorly.tenalg")
    tenalg.khatri_rao = orc.khatri_rao
    tenalg.mode_dot = orc.mode_dot
    tenalg.multi_mode_dot = orc.multi_mode_dot
    tenalg.inner = lambda a, b: np.sum(a * b)
    tenalg.contract = lambda a, ma, b, mb: np.tensordot(a, b, axes=(ma, mb))
    dec = types.ModuleType("tensorly.decomposition")

    def _no_tucker(*a, **k):
        raise NotImplementedError("tensorly.decomposition.tucker is not restated (initialiser, out of scope)")
    dec.tucker = _no_tucker
    tl.base, tl.tenalg, tl.decomposition = base, tenalg, dec
    for name, mod in (("tensorly", tl), ("tensorly.base", base), ("tensorly.tenalg", tenalg),
                      ("tensorly.decomposition", dec)):
        sys.modules[name] = mod
    return orc


def main():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden.py needs /root/reference (build container only)")
    orc = _install_tensorly_standin()
    sys.path.insert(0, REF)
    from nn_fac.nmf import nmf as ref_nmf
    import nn_fac.nmf as ref_nmf_mod
    import nn_fac.ntf as ref_ntf_mod
    import nn_fac.update_rules.nnls as ref_nnls
    import nn_fac.update_rules.mu as ref_mu
    import nn_fac.utils.beta_divergence as ref_bd
    import nn_fac.utils.initialize_factors as ref_init
    os.makedirs(OUT, exist_ok=True)

    # ---------------- G0: the reference's own known answers (tests/NMF_tests.py) ----------------
    np.random.seed(0)
    random.seed(0)
    rank = random.randint(3, 10)
    shape = (random.randint(20, 100), random.randint(20, 100))
    U0 = np.random.rand(shape[0], rank)
    V0 = np.random.rand(rank, shape[1])
    data = U0 @ V0 + 1e-2 * np.random.rand(*shape)
    assert abs(data[0][0] - 2.143518599859098) < 1e-7          # NMF_tests.py:68
    Ui, Vi = ref_init.nmf_initialization(data, rank, init_type="random", deterministic=True, seed=0)
    assert abs(Ui[0][0] - 0.5488135) < 1e-7 and abs(Vi[0][0] - 1.15834001e-01) < 1e-7   # :40-41
    known = {  # rule, beta, seed -> (U00, V00, cost0, cost_last)   NMF_tests.py:76-81,94-99,112-117,130-135
        ("hals", 2, 0): (0.55430769, 0.11523809, 0.009438764349822035, 0.008805158842036184),
        ("mu", 2, 82): (0.35280947364767296, 0.44719984549809116, 111.43110252634743, 68.8373870926001),
        ("mu", 1, 82): (0.3718053134990678, 0.4367362187193684, 51.47596084683006, 32.742423893466851),
        ("mu", 0, 82): (0.32746152037135323, 0.4098870587115991, 71.40741383137126, 20.041539547898314),
    }
    g0 = {"data": data, "rank": rank}
    for (rule, beta, seed), (u00, v00, c0, c1) in known.items():
        U, V, costs, _ = ref_nmf(data, rank, init="random", n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                 sparsity_coefficients=[None, None], fixed_modes=[], normalize=[False, False],
                                 return_costs=True, deterministic=True, seed=seed)
        for got, want in ((U[0][0], u00), (V[0][0], v00), (costs[0], c0), (costs[-1], c1)):
            assert abs(got - want) < 1e-7, (rule, beta, got, want)
        tag = f"{rule}_b{beta}_s{seed}"
        g0[f"U_{tag}"], g0[f"V_{tag}"], g0[f"costs_{tag}"] = U, V, np.array(costs)
        g0[f"known_{tag}"] = np.array([u00, v00, c0, c1])
        # the oracle must reproduce the reference on the same call
        Uo, Vo, co, _ = orc.nmf(data, rank, init="random", n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                return_costs=True, deterministic=True, seed=seed)
        assert np.allclose(Uo, U, rtol=1e-12, atol=1e-14) and np.allclose(co, costs, rtol=1e-12)
    np.savez_compressed(os.path.join(OUT, "g0_known_answers.npz"), **g0)
    print("G0 ok: reference known answers re-asserted; oracle == reference")

    # ---------------- G1: hals_nnls_acc single calls ----------------
    # inputs stored once per shape (s{i}_*), outputs per case (c{j}_*; c{j}_shape -> input set)
    g1 = {}
    case = 0
    rng = np.random.RandomState(1234)

    def put(case, sidx, kw, V, eps, cnt, log):
        p = f"c{case}_"
        g1[p + "shape"], g1[p + "V"] = np.int64(sidx), V
        g1[p + "eps"], g1[p + "cnt"], g1[p + "nodelta"] = np.float64(eps), np.int64(cnt), np.array(log)
        g1[p + "kw"] = np.array([kw.get("maxiter", 500), kw.get("delta", 0.01),
                                 -1.0 if kw.get("sparsity_coefficient") is None else kw["sparsity_coefficient"],
                                 float(kw.get("normalize", False)), float(kw.get("nonzero", False))])

    sidx = 0
    for r, n in ((3, 1), (10, 100), (50, 129), (16, 64), (33, 200), (100, 70)):
        A = rng.rand(3 * r + 5, r)
        M = A @ rng.rand(r, n) + 0.05 * rng.rand(3 * r + 5, n)
        UtU, UtM, Vin = A.T @ A, A.T @ M, rng.rand(r, n)
        g1[f"s{sidx}_UtM"], g1[f"s{sidx}_UtU"], g1[f"s{sidx}_Vin"] = UtM, UtU, Vin
        variants = [dict(), dict(sparsity_coefficient=0.1), dict(normalize=True), dict(nonzero=True),
                    dict(maxiter=1), dict(maxiter=2), dict(maxiter=100, delta=0.0), dict(maxiter=7, delta=1e-6),
                    dict(sparsity_coefficient=0.3, normalize=True, maxiter=20)]
        for kw in variants:
            kw = dict(alpha=math.inf, **kw)
            V, eps, cnt, rho = ref_nnls.hals_nnls_acc(UtM, UtU, Vin, **kw)
            log = []
            Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM, UtU, Vin, sweep_log=log, **kw)
            assert cnt == cnto and np.allclose(V, Vo, rtol=1e-12, atol=1e-15) and np.isclose(eps, epso, rtol=1e-12)
            put(case, sidx, kw, V, eps, cnt, log)
            case += 1
        sidx += 1
    # zero Gram diagonal (tests/nnls_tests.py:35-38): row silently skipped
    UtU = rng.rand(8, 8); UtU = UtU @ UtU.T; UtU[2, 2] = 0
    UtM, Vin = rng.rand(8, 8), rng.rand(8, 8)
    g1[f"s{sidx}_UtM"], g1[f"s{sidx}_UtU"], g1[f"s{sidx}_Vin"] = UtM, UtU, Vin
    log = []
    V, eps, cnt, _ = ref_nnls.hals_nnls_acc(UtM, UtU, Vin, alpha=math.inf)
    orc.hals_nnls_acc(UtM, UtU, Vin, alpha=math.inf, sweep_log=log)
    put(case, sidx, {}, V, eps, cnt, log); case += 1; sidx += 1
    # all-zero row forced by the projection + nonzero guard
    UtU = np.eye(4) + 0.1; UtM = rng.rand(4, 6); UtM[1, :] = -5.0; Vin = rng.rand(4, 6)
    g1[f"s{sidx}_UtM"], g1[f"s{sidx}_UtU"], g1[f"s{sidx}_Vin"] = UtM, UtU, Vin
    kw = dict(nonzero=True, maxiter=5)
    log = []
    V, eps, cnt, _ = ref_nnls.hals_nnls_acc(UtM, UtU, Vin, alpha=math.inf, **kw)
    Vo = orc.hals_nnls_acc(UtM, UtU, Vin, alpha=math.inf, sweep_log=log, **kw)[0]
    assert np.allclose(V, Vo, rtol=1e-12, atol=1e-30)
    put(case, sidx, kw, V, eps, cnt, log); case += 1; sidx += 1
    # rectangular quirk (tests/nnls_tests.py:44-45): r taken from UtM (8), UtU 15x15, V 15x1
    UtU = rng.rand(15, 15); UtM = rng.rand(8, 1); Vin = rng.rand(15, 1)
    g1[f"s{sidx}_UtM"], g1[f"s{sidx}_UtU"], g1[f"s{sidx}_Vin"] = UtM, UtU, Vin
    log = []
    V, eps, cnt, _ = ref_nnls.hals_nnls_acc(UtM, UtU, Vin, alpha=math.inf)
    Vo = orc.hals_nnls_acc(UtM, UtU, Vin, alpha=math.inf, sweep_log=log)[0]
    assert np.allclose(V, Vo, rtol=1e-12)
    put(case, sidx, {}, V, eps, cnt, log); case += 1; sidx += 1
    g1["ncases"] = np.int64(case)
    np.savez_compressed(os.path.join(OUT, "g1_hals.npz"), **g1)
    print(f"G1 ok: {case} hals_nnls_acc cases")

    # ---------------- G2/G3: mu_betadivmin, switch_alternate_mu, beta_divergence, gamma_beta ----------------
    g2 = {}
    rng = np.random.RandomState(77)
    m, n, r = 37, 29, 6
    U, V = rng.rand(m, r) + 0.05, rng.rand(r, n) + 0.05
    M = (rng.rand(m, r) @ rng.rand(r, n)) + 0.05
    g2["U"], g2["V"], g2["M"] = U, V, M
    betas = [0, 0.5, 1, 1.5, 2, 3, 4]
    g2["betas"] = np.array(betas, dtype=np.float64)
    for b in betas:
        g2[f"muU_b{b}"] = ref_mu.switch_alternate_mu(M, U, V, b, "U")
        g2[f"muV_b{b}"] = ref_mu.switch_alternate_mu(M, U, V, b, "V")
        g2[f"div_b{b}"] = np.float64(ref_bd.beta_divergence(M, U @ V, b))
        g2[f"gamma_b{b}"] = np.float64(ref_bd.gamma_beta(b))
        assert np.allclose(orc.switch_alternate_mu(M, U, V, b, "U"), g2[f"muU_b{b}"], rtol=1e-13)
        assert np.allclose(orc.switch_alternate_mu(M, U, V, b, "V"), g2[f"muV_b{b}"], rtol=1e-13)
        assert np.isclose(orc.beta_divergence(M, U @ V, b), g2[f"div_b{b}"], rtol=1e-13)
    np.savez_compressed(os.path.join(OUT, "g2_mu.npz"), **g2)
    print("G2/G3 ok")

    # ---------------- G4: config A end-to-end (200x100 rank 10), fp64 and fp32 ----------------
    g4 = {}
    X64, U064, V064 = orc.synth_nmf(200, 100, 10, seed=0, dtype=np.float64)
    g4["X"], g4["U0"], g4["V0"] = X64, U064, V064
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        X, U0, V0 = X64.astype(dt), U064.astype(dt), V064.astype(dt)
        for rule, beta in (("hals", 2), ("mu", 2), ("mu", 1), ("mu", 0), ("mu", 1.5), ("mu", 3)):
            U, V, costs, _ = ref_nmf(X, 10, init="custom", U_0=U0, V_0=V0, n_iter_max=10, tol=0, update_rule=rule,
                                     beta=beta, return_costs=True, deterministic=True)
            sw = []
            Uo, Vo, co, _ = orc.nmf(X, 10, init="custom", U_0=U0, V_0=V0, n_iter_max=10, tol=0, update_rule=rule,
                                    beta=beta, return_costs=True, deterministic=True, sweeps=sw)
            tol = 1e-12 if dt == np.float64 else 2e-5
            assert np.allclose(Uo, U, rtol=tol, atol=tol) and np.allclose(co, costs, rtol=tol), (rule, beta, tag)
            k = f"{rule}_b{beta}_{tag}"
            g4[f"U_{k}"], g4[f"V_{k}"], g4[f"costs_{k}"], g4[f"sweeps_{k}"] = U, V, np.array(costs), np.array(sw)
    # sparsity + normalize + fixed modes variants (fp64)
    U, V, costs, _ = ref_nmf(X64, 10, init="custom", U_0=U064, V_0=V064, n_iter_max=6, tol=0, update_rule="hals",
                             sparsity_coefficients=[0.05, 0.1], normalize=[False, True], return_costs=True,
                             deterministic=True)
    g4["U_hals_sparse_norm"], g4["V_hals_sparse_norm"], g4["costs_hals_sparse_norm"] = U, V, np.array(costs)
    U, V, costs, _ = ref_nmf(X64, 10, init="custom", U_0=U064, V_0=V064, n_iter_max=4, tol=0, update_rule="hals",
                             fixed_modes=[0], return_costs=True, deterministic=True)
    g4["U_hals_fixed0"], g4["V_hals_fixed0"], g4["costs_hals_fixed0"] = U, V, np.array(costs)
    np.savez_compressed(os.path.join(OUT, "g4_nmf_configA.npz"), **g4)
    print("G4 ok")

    # ---------------- G5: mid-size NMF 2000x500 r50 fp32: cost trajectory + sweeps + strided samples ----------------
    X, U0, V0 = orc.synth_nmf(2000, 500, 50, seed=3, dtype=np.float32)
    g5 = {}
    for rule, beta, nit in (("hals", 2, 5), ("mu", 1, 5)):
        U, V, costs, _ = ref_nmf(X, 50, init="custom", U_0=U0, V_0=V0, n_iter_max=nit, tol=0, update_rule=rule,
                                 beta=beta, return_costs=True, deterministic=True)
        sw = []
        Uo, Vo, co, _ = orc.nmf(X, 50, init="custom", U_0=U0, V_0=V0, n_iter_max=nit, tol=0, update_rule=rule,
                                beta=beta, return_costs=True, deterministic=True, sweeps=sw)
        assert np.allclose(co, costs, rtol=1e-4)
        k = f"{rule}_b{beta}"
        g5[f"U_{k}"], g5[f"V_{k}"], g5[f"costs_{k}"], g5[f"sweeps_{k}"] = U[::16].copy(), V[:, ::4].copy(), \
            np.array(costs, dtype=np.float64), np.array(sw)
    g5["seed"], g5["shape"] = np.int64(3), np.array([2000, 500, 50])
    np.savez_compressed(os.path.join(OUT, "g5_nmf_mid.npz"), **g5)
    print("G5 ok")

    # ---------------- G6: NTF via one_ntf_step(alpha=inf) ----------------
    g6 = {}
    for name, shape, R in (("small", (12, 10, 8), 4), ("cube", (40, 40, 40), 6), ("ragged", (33, 17, 21), 5)):
        T, F0 = orc.synth_ntf(shape, R, seed=11, dtype=np.float64)
        g6[f"{name}_T"] = T
        for i, f in enumerate(F0):
            g6[f"{name}_F0_{i}"] = f
        unf = [orc.unfold(T, mo) for mo in range(3)]
        nrm = np.sqrt(np.sum(T ** 2))
        for rule, beta in (("hals", 2), ("mu", 2), ("mu", 1)):
            fr, fo = [f.copy() for f in F0], [f.copy() for f in F0]
            costs = []
            for it in range(5):
                fr, c = ref_ntf_mod.one_ntf_step(unf, R, fr, nrm, rule, beta, [None] * 3, [], [False] * 3,
                                                 alpha=math.inf, delta=0.01)
                fo, c2 = orc.one_ntf_step(unf, R, fo, nrm, rule, beta, [None] * 3, [], [False] * 3,
                                          alpha=math.inf, delta=0.01)
                assert np.isclose(c, c2, rtol=1e-10), (name, rule, beta, c, c2)
                costs.append(c)
            for i in range(3):
                assert np.allclose(fr[i], fo[i], rtol=1e-10, atol=1e-13)
                g6[f"{name}_{rule}_b{beta}_F{i}"] = fr[i]
            g6[f"{name}_{rule}_b{beta}_costs"] = np.array(costs)
    np.savez_compressed(os.path.join(OUT, "g6_ntf.npz"), **g6)
    print("G6 ok")

    # ---------------- G7: NTD (ntd.py, mu.py:99-159) ----------------
    # the reference's own known answers first (tests/NTD_tests.py:18-27 setUp, :138-155, :177-195, :217-255)
    rnd = types.ModuleType("tensorly.random")
    rnd.random_tucker = lambda shape, rank, full=True, random_state=None: orc.random_tucker_full(shape, rank, random_state)
    sys.modules["tensorly.random"] = rnd
    sys.modules["tensorly"].random = rnd
    import random as pyrandom
    import nn_fac.ntd as ref_ntd
    np.random.seed(0)
    pyrandom.seed(0)
    ranks = (pyrandom.randint(3, 10), pyrandom.randint(3, 10), pyrandom.randint(3, 10))
    shape = (pyrandom.randint(20, 100), pyrandom.randint(20, 100), pyrandom.randint(20, 100))
    for mo in range(3):
        np.random.rand(shape[mo], ranks[mo])        # setUp draws factors_0..2 and core before the tensor
    np.random.rand(*ranks)
    Tref = np.abs(rnd.random_tucker(shape, ranks, full=True, random_state=0)) + 1e-2 * np.random.rand(*shape)
    assert abs(Tref[0][0][0] - 21.974433828159626) < 1e-7                       # NTD_tests.py:141
    known = {("hals", 2): (0.5501411956914489, 0.9680069293664532, 0.965086018254149, 0.3744157888431357,
                           2.6164388105612055e-08, 2.603936417799217e-08),     # :148-155
             ("mu", 2): (0.5489250094099122, 0.9679994929177957, 0.9650887516147171, 0.3744138868288453,
                         1.5935015225944391, 1.5931775725367523),             # :188-195
             ("mu", 1): (0.5489424379755086, 0.9679939115774175, 0.9650587287572271, 0.3744133064030978,
                         0.12936809612191502, 0.1293171172587153),            # :228-235
             ("mu", 0): (0.5488704375518113, 0.9680879599528461, 0.9650465314632987, 0.3744250029550508,
                         0.01749656252808407, 0.014723505531139436)}          # :248-255
    g7 = {"ref_shape": np.array(shape), "ref_ranks": np.array(ranks)}
    for (rule, beta), want in known.items():
        core, facs, costs, toc = ref_ntd.ntd(Tref, list(ranks), init="random", n_iter_max=10, tol=1e-8, update_rule=rule,
                                             beta=beta, sparsity_coefficients=[None] * 4, fixed_modes=[],
                                             normalize=[False] * 4, verbose=False, return_costs=True, deterministic=True,
                                             seed=0)
        got = (facs[0][0][0], facs[1][0][0], facs[2][0][0], core[0, 0, 0], costs[0], costs[-1])
        assert all(abs(a - b) < 5e-8 for a, b in zip(got, want)), (rule, beta, got, want)   # assertAlmostEqual: 7 places
        c0, f0 = orc.ntd_random_init(shape, list(ranks), 0)
        sw, pg = [], []
        co, fo, cso, _ = orc.compute_ntd(Tref, list(ranks), c0, f0, n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                         sparsity_coefficients=[None] * 4, fixed_modes=[], normalize=[False] * 4,
                                         return_costs=True, deterministic=True, sweeps=sw, pg_iters=pg)
        assert np.allclose(co, core, rtol=1e-7, atol=1e-10) and np.allclose(cso, costs, rtol=1e-6, atol=1e-12)
        tag = f"ref_{rule}_b{beta}"
        g7[f"{tag}_core"], g7[f"{tag}_costs"] = core, np.array(costs)
        for i in range(3):
            assert np.allclose(fo[i], facs[i], rtol=1e-7, atol=1e-10)
            g7[f"{tag}_F{i}"] = facs[i]
        g7[f"{tag}_sweeps"], g7[f"{tag}_pg"] = np.array(sw), np.array(pg)
    # small seeded problem: step functions called directly, options exercised
    rs = np.random.RandomState(7)
    shp, rk = (10, 9, 8), [4, 3, 2]
    Ts = orc.multi_mode_dot(rs.rand(*rk), [rs.rand(shp[i], rk[i]) for i in range(3)]) + 0.05 * rs.rand(*shp)
    c0 = rs.rand(*rk)
    f0 = [rs.rand(shp[i], rk[i]) for i in range(3)]
    g7["small_T"], g7["small_core0"] = Ts, c0
    for i in range(3):
        g7[f"small_F0_{i}"] = f0[i]
    nrm = np.sqrt(np.sum(Ts ** 2))
    cases = {"plain": dict(sp=[None] * 4, fixed=[], norm=[False] * 4, mcn=None),
             "sparse": dict(sp=[0.01, None, 0.02, 0.05], fixed=[], norm=[False] * 4, mcn=None),
             "norm": dict(sp=[None] * 4, fixed=[], norm=[True, False, True, True], mcn=1),
             "fixed1": dict(sp=[None] * 4, fixed=[1], norm=[False] * 4, mcn=None)}
    for name, cfg in cases.items():
        cr, fr = c0.copy(), [f.copy() for f in f0]
        co, fo = c0.copy(), [f.copy() for f in f0]
        costs = []
        for it in range(4):
            cr, fr, c = ref_ntd.one_ntd_step(Ts, rk, cr, fr, nrm, list(cfg["sp"]), list(cfg["fixed"]), list(cfg["norm"]),
                                             cfg["mcn"], alpha=math.inf, delta=0.01)
            co, fo, c2 = orc.one_ntd_step(Ts, rk, co, fo, nrm, list(cfg["sp"]), list(cfg["fixed"]), list(cfg["norm"]),
                                          cfg["mcn"], alpha=math.inf, delta=0.01)
            assert np.isclose(c, c2, rtol=1e-8, atol=1e-13), (name, it, c, c2)
            costs.append(c)
        assert np.allclose(cr, co, rtol=1e-8, atol=1e-12)
        g7[f"small_hals_{name}_core"], g7[f"small_hals_{name}_costs"] = cr, np.array(costs)
        for i in range(3):
            assert np.allclose(fr[i], fo[i], rtol=1e-8, atol=1e-12)
            g7[f"small_hals_{name}_F{i}"] = fr[i]
    for beta in (0, 0.5, 1, 2, 3):
        cr, fr = c0.copy(), [f.copy() for f in f0]
        co, fo = c0.copy(), [f.copy() for f in f0]
        costs = []
        for it in range(4):
            cr, fr, c = ref_ntd.one_ntd_step_mu(Ts, rk, cr, fr, beta, nrm, [], [False] * 4, None)
            co, fo, c2 = orc.one_ntd_step_mu(Ts, rk, co, fo, beta, nrm, [], [False] * 4, None)
            assert np.isclose(c, c2, rtol=1e-9), (beta, it, c, c2)
            costs.append(c)
        assert np.allclose(cr, co, rtol=1e-9, atol=1e-13)
        g7[f"small_mu_b{beta}_core"], g7[f"small_mu_b{beta}_costs"] = cr, np.array(costs)
        for i in range(3):
            assert np.allclose(fr[i], fo[i], rtol=1e-9, atol=1e-13)
            g7[f"small_mu_b{beta}_F{i}"] = fr[i]
        g7[f"small_mut_b{beta}"] = ref_mu.mu_tensorial(c0, f0, Ts, beta)
        assert np.allclose(g7[f"small_mut_b{beta}"], orc.mu_tensorial(c0, f0, Ts, beta), rtol=1e-12)
    np.savez_compressed(os.path.join(OUT, "g7_ntd.npz"), **g7)
    print("G7 ok")
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"fixtures written to {os.path.normpath(OUT)}: {sz/1e6:.2f} MB")


if __name__ == "__main__":
    main()
