"""Ranks above 128 on the matrix path (the reference takes any rank <= min(shape): nn_fac/nmf.py:175-178, nnls.py:158): the
contractions and cost passes walk the rank in chunks of 128, the sweeps run in the generic kernel on columns in global memory.
Kernels against fp64 NumPy / the oracle, then nmf() end to end.  Needs a MI355X."""
import math

import numpy as np
import pytest
import torch

import nnfac_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(built_lib):
    from nn_fac_amd.engine import get_engine
    assert torch.cuda.is_available()
    return get_engine("cuda:0")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# one chunk + a ragged one, two full chunks + one row, three chunks, ragged everything, X rows not 16-byte aligned
SHAPES = [(2000, 500, 200), (777, 333, 257), (1000, 260, 129), (513, 131, 300), (4100, 1024, 256), (300, 310, 290)]


@pytest.mark.parametrize("m,n,r", SHAPES)
def test_gram_xty_xht_costs_above_rank_128(eng, m, n, r):
    rng = np.random.RandomState(m * 7 + n * 3 + r)
    X = rng.rand(m, n).astype(np.float32) + 0.05
    Ut = (rng.rand(r, m) / math.sqrt(r)).astype(np.float32)
    V = rng.rand(r, n).astype(np.float32)
    Xd, Utd, Vd = dev(X), dev(Ut), dev(V)
    X64, U64, V64 = X.astype(np.float64), Ut.astype(np.float64), V.astype(np.float64)
    G = eng.gram(Vd).cpu().numpy()
    assert rel(G, V64 @ V64.T) < 1e-5 and np.array_equal(G, G.T)          # symmetric bit for bit, as below 128
    assert rel(eng.gram(Utd).cpu().numpy(), U64 @ U64.T) < 1e-5
    assert rel(eng.xty(Xd, Utd).cpu().numpy(), U64 @ X64) < 1e-5
    assert rel(eng.xht(Xd, Vd).cpu().numpy(), V64 @ X64.T) < 1e-5
    P = U64.T @ V64
    want = np.sum((X64 - P) ** 2)
    assert abs(float(eng.frob_resid(Xd, Utd, Vd)) - want) <= 1e-5 * want
    for beta in (2, 1, 0, 1.5):
        want = orc.beta_divergence(X64, P, beta)
        got = float(eng.betadiv(Xd, Utd, Vd, beta))
        assert abs(got - want) <= 2e-5 * abs(want) + 1e-9, (beta, got, want)
    # calls below 128 after calls above it (the scratch stays registered)
    assert abs(float(eng.frob_resid(Xd, Utd[:100], Vd[:100])) - np.sum((X64 - U64[:100].T @ V64[:100]) ** 2)) <= 1e-5 * want + 1e-3


def test_cost_above_rank_128_from_the_workspace_tail_and_without_room(built_lib):
    """Without caller scratch the m x n model lives in the tail of the context workspace; a workspace too small for it is a
    clean NNF_ERR_WORKSPACE, not a wrong number."""
    from nn_fac_amd.engine import Engine, EngineError, _ptr, _ld
    from nn_fac_amd import _lib
    rng = np.random.RandomState(5)
    m, n, r = 3000, 700, 190
    X, Ut, V = rng.rand(m, n), rng.rand(r, m) / 14, rng.rand(r, n)
    want = np.sum((X.astype(np.float32).astype(np.float64) - Ut.astype(np.float32).astype(np.float64).T @ V.astype(np.float32).astype(np.float64)) ** 2)
    Xd, Utd, Vd = dev(X), dev(Ut), dev(V)
    o = torch.empty(1, dtype=torch.float64, device="cuda")
    e = Engine(torch.device("cuda:0"), workspace_bytes=64 << 20)
    _lib.check(e.lib.nnf_frob_resid_f32(e.ctx, _ptr(Xd), m, n, _ld(Xd), _ptr(Utd), _ld(Utd), _ptr(Vd), _ld(Vd), r, _ptr(o), None),
               "nnf_frob_resid_f32")
    assert abs(float(o) - want) <= 1e-5 * want
    small = Engine(torch.device("cuda:0"), workspace_bytes=12 << 20)
    with pytest.raises(EngineError):
        _lib.check(small.lib.nnf_frob_resid_f32(small.ctx, _ptr(Xd), m, n, _ld(Xd), _ptr(Utd), _ld(Utd), _ptr(Vd), _ld(Vd), r, _ptr(o),
                                                None), "nnf_frob_resid_f32")


@pytest.mark.parametrize("r,ncols", [(200, 500), (129, 3000), (257, 700), (300, 64), (160, 20000)])
@pytest.mark.parametrize("opts", [{}, {"sparsity_coefficient": 0.05}, {"normalize": True}, {"nonzero": True}])
def test_hals_above_rank_128_vs_oracle(eng, r, ncols, opts):
    if ncols > 4000 and opts:
        pytest.skip("one large case is enough")
    rng = np.random.RandomState(r + ncols)
    A = rng.rand(3 * r, r)
    UtU = A.T @ A
    cols = min(ncols, 1000)                      # the large case: identical column tiles (same decisions, oracle on one tile)
    reps = ncols // cols if ncols > 4000 else 1
    cols = ncols if reps == 1 else cols
    UtM_s = A.T @ (A @ rng.rand(r, cols) + 0.05 * rng.rand(3 * r, cols))
    V0_s = rng.rand(r, cols)
    Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM_s, UtU, V0_s.copy(), maxiter=40, alpha=math.inf, delta=0.01, **opts)
    Vd = dev(np.tile(V0_s, (1, reps)))
    st = eng.hals_solve(dev(np.tile(UtM_s, (1, reps))), dev(UtU), Vd, 40, delta=0.01, sparsity=opts.get("sparsity_coefficient"),
                        normalize=opts.get("normalize", False), nonzero=opts.get("nonzero", False)).cpu()
    got = Vd.cpu().numpy()
    assert int(st[3]) == 0 and int(st[1]) == cnto
    assert rel(got[:, :cols], Vo) < 2e-4
    if reps > 1:
        assert np.array_equal(got[:, :cols], got[:, -cols:])


def test_hals_blind_sweeps_and_snapshots_above_rank_128(eng):
    rng = np.random.RandomState(9)
    r, n = 150, 900
    A = rng.rand(400, r)
    UtU, UtM, V0 = A.T @ A, A.T @ rng.rand(400, n), rng.rand(r, n)
    log = []
    cur, want = V0.copy(), []
    for s in range(6):
        cur, *_ = orc.hals_nnls_acc(UtM, UtU, cur, maxiter=1, alpha=math.inf, delta=0.0, sweep_log=log)
        want.append(cur.copy())
    Vd = dev(V0)
    snaps = torch.zeros((4, r, n), dtype=torch.float32, device="cuda")
    nd = eng.hals_sweeps(dev(UtM), dev(UtU), Vd, 6, snapshots=snaps, snap_first=2).cpu().numpy()
    assert rel(Vd.cpu().numpy(), want[-1]) < 1e-4
    for j in range(4):
        assert rel(snaps[j].cpu().numpy(), want[2 + j]) < 1e-4
    assert np.array_equal(snaps[3].cpu().numpy(), Vd.cpu().numpy())
    np.testing.assert_allclose(nd, log, rtol=5e-3)


def test_hals_nnls_acc_drop_in_above_rank_128(built_lib):
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    rng = np.random.RandomState(2)
    r, n = 180, 300
    A = rng.rand(500, r)
    UtU, UtM, V0 = A.T @ A, A.T @ rng.rand(500, n), rng.rand(r, n)
    Vo, epso, cnto, rho = orc.hals_nnls_acc(UtM, UtU, V0.copy(), maxiter=30, alpha=math.inf, delta=0.01)
    V, eps, cnt, _ = hals_nnls_acc(UtM, UtU, V0, maxiter=30, alpha=math.inf, delta=0.01)
    assert cnt == cnto and rel(V, Vo) < 2e-4 and abs(eps - epso) <= 5e-3 * epso


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1), ("mu", 2), ("mu", 0.5)])
def test_nmf_at_rank_200_vs_oracle(built_lib, rule, beta):
    """nmf(X, 200) HALS and MU against the oracle at 5e-4 on a 2000 x 500 problem."""
    from nn_fac_amd.nmf import compute_nmf
    X, U0, V0 = orc.synth_nmf(2000, 500, 200, seed=4, dtype=np.float32)
    sw = []
    kw = dict(n_iter_max=4, tol=0, update_rule=rule, beta=beta, return_costs=True, deterministic=True)
    U, V, costs, _ = compute_nmf(X, 200, U0, V0, sweep_log=sw, **kw)
    so = []
    Uo, Vo, co, _ = orc.compute_nmf(X.astype(np.float64), 200, U0.astype(np.float64), V0.astype(np.float64), sweeps=so, **kw)
    assert rel(U, Uo) < 5e-4 and rel(V, Vo) < 5e-4
    np.testing.assert_allclose(costs, co, rtol=1e-3)
    if rule == "hals":
        assert sw == so, (sw, so)


def test_nmf_driver_at_rank_130_random_init(built_lib):
    from nn_fac_amd.nmf import nmf
    rng = np.random.RandomState(1)
    X = (rng.rand(400, 130) @ rng.rand(130, 600)).astype(np.float32)
    U, V, costs, _ = nmf(X, 130, init="random", n_iter_max=6, tol=0, update_rule="hals", return_costs=True, deterministic=True)
    assert U.shape == (400, 130) and V.shape == (130, 600) and np.all(U >= 0) and np.all(V >= 0)
    assert all(b <= a * (1 + 1e-4) for a, b in zip(costs, costs[1:]))
    want = np.linalg.norm(X.astype(np.float64) - U.astype(np.float64) @ V.astype(np.float64)) ** 2      # nmf.py:452, no sparsity
    assert abs(costs[-1] - want) <= 1e-3 * want


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_mttkrp_and_cp_cost_above_rank_128(eng, mode):
    rng = np.random.RandomState(11 + mode)
    I, J, K, R = 60, 70, 90, 150
    T = rng.rand(I, J, K).astype(np.float32) + 0.05
    F = [(rng.rand(d, R) / np.sqrt(R)).astype(np.float32) for d in (I, J, K)]
    Ft = [dev(f.T) for f in F]
    Td = torch.from_numpy(T).cuda()
    F64 = [f.astype(np.float64) for f in F]
    want = orc.unfold(T.astype(np.float64), mode) @ orc.khatri_rao(F64, skip_matrix=mode)
    assert rel(eng.mttkrp3(Td, Ft, mode).cpu().numpy().T, want) < 1e-5
    if mode == 0:
        model = (F64[0] @ orc.khatri_rao(F64, skip_matrix=0).T).reshape(T.shape)
        for beta in (2, 1, 0.5):
            w = orc.beta_divergence(T.astype(np.float64), model, beta)
            g = float(eng.cp3_betadiv(Td, Ft, beta))
            assert abs(g - w) <= 2e-5 * abs(w), (beta, g, w)
        Y = eng.ttm3(Td, Ft[2], 2)                      # R x I x J
        assert rel(Y.cpu().numpy(), np.einsum("ijk,kr->rij", T.astype(np.float64), F64[2])) < 1e-5
        for axis, other in ((2, 1), (1, 0)):
            got = eng.mttkrp3_from_partial(Y, Ft[other], axis).cpu().numpy()
            y = Y.cpu().numpy().astype(np.float64)
            ref = np.einsum("rab,rb->ra", y, F64[1].T) if axis == 2 else np.einsum("rab,ra->rb", y, F64[0].T)
            assert rel(got, ref) < 1e-5


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1), ("mu", 2)])
def test_ntf_at_rank_140_vs_oracle(built_lib, rule, beta):
    """NTF above rank 128 (the reference puts no limit on the CP rank): MTTKRPs and cost passes in rank chunks, sweeps in the
    generic kernel; against the fp64 oracle on a 150 x 141 x 160 tensor."""
    from nn_fac_amd.ntf import compute_ntf
    shape, R = (150, 141, 160), 140
    T, F0 = orc.synth_ntf(shape, R, seed=5, dtype=np.float32)
    kw = dict(n_iter_max=3, tol=0, update_rule=rule, beta=beta, alpha=math.inf, sparsity_coefficients=[None] * 3,
              normalize=[False] * 3, return_costs=True)
    F, costs, _ = compute_ntf(T, R, [f.copy() for f in F0], **kw)
    Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], **kw)
    for a, b in zip(F, Fo):
        assert rel(a, b) < 1e-3
    np.testing.assert_allclose(costs, co, rtol=2e-3)


def test_randomised_shapes_above_rank_128(built_lib):
    """tools/probes/bigrank_stress.py: odd sizes, ragged last chunks (ranks 129 ... 400), every flag of the solve -- products and
    costs within 3e-5 of float64, solves within 3e-4 of the oracle with equal sweep counts."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bigrank_stress", os.path.join(root, "tools", "probes", "bigrank_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.run(cases=24, seed=3, verbose=False)
    assert not bad, bad[:3]
