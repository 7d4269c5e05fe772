"""MTTKRP kernels and the NTF driver against the oracle and the reference fixtures (g6).  Needs a MI355X."""
import math

import numpy as np
import pytest
import torch

import nnfac_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.fixture(scope="module")
def eng(built_lib):
    from nn_fac_amd.engine import get_engine
    return get_engine("cuda:0")


@pytest.mark.parametrize("shape,R", [((12, 10, 8), 4), ((40, 40, 40), 6), ((33, 17, 21), 5), ((300, 64, 128), 30),
                                     ((64, 300, 70), 17), ((5, 7, 260), 3), ((128, 128, 128), 64), ((70, 50, 90), 100)])
def test_mttkrp_all_modes(eng, shape, R):
    rng = np.random.RandomState(sum(shape) + R)
    T = rng.rand(*shape).astype(np.float32)
    F = [rng.rand(s, R).astype(np.float32) for s in shape]
    Td, Ft = dev(T), [dev(f.T) for f in F]
    T64, F64 = T.astype(np.float64), [f.astype(np.float64) for f in F]
    for mode in range(3):
        want = orc.unfold(T64, mode) @ orc.khatri_rao(F64, skip_matrix=mode)       # ntf.py:448-449
        got = eng.mttkrp3(Td, Ft, mode).cpu().numpy().T
        assert rel(got, want) < 1e-5, mode
    model = np.einsum('ir,jr,kr->ijk', *F64)
    Tp = (T64 + 0.1).astype(np.float32).astype(np.float64)
    for beta in (2, 1, 0, 1.5):
        want = orc.beta_divergence(Tp, model, beta)
        got = float(eng.cp3_betadiv(dev(Tp), Ft, beta))
        assert abs(got - want) <= 3e-5 * abs(want), (beta, got, want)


@pytest.mark.parametrize("name", ["small", "cube", "ragged"])
@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 2), ("mu", 1)])
def test_ntf_against_reference_fixtures(golden, built_lib, name, rule, beta):
    from nn_fac_amd.ntf import compute_ntf
    g = golden("g6_ntf.npz")
    T = g[f"{name}_T"]
    F0 = [g[f"{name}_F0_{i}"] for i in range(3)]
    R = F0[0].shape[1]
    F, costs, toc = compute_ntf(T, R, F0, n_iter_max=5, tol=0, update_rule=rule, beta=beta,
                                sparsity_coefficients=[None] * 3, fixed_modes=[], normalize=[False] * 3,
                                return_costs=True, alpha=math.inf, delta=0.01)
    tolF, tolC = (2e-3, 2e-3) if rule == "hals" else (5e-5, 5e-5)
    for i in range(3):
        assert isinstance(F[i], np.ndarray) and F[i].shape == F0[i].shape
        assert rel(F[i], g[f"{name}_{rule}_b{beta}_F{i}"]) < tolF, i
    np.testing.assert_allclose(costs, g[f"{name}_{rule}_b{beta}_costs"], rtol=tolC)


def test_one_ntf_step_signature(golden, built_lib):
    from nn_fac_amd.ntf import one_ntf_step, ntf
    from nn_fac_amd.utils import errors as err
    g = golden("g6_ntf.npz")
    T = g["small_T"]
    F0 = [g[f"small_F0_{i}"] for i in range(3)]
    unf = [orc.unfold(T, m) for m in range(3)]
    nrm = np.sqrt(np.sum(T ** 2))
    F, c = one_ntf_step(unf, 4, F0, nrm, "hals", 2, [None] * 3, [], [False] * 3, alpha=math.inf)
    Fo, co = orc.one_ntf_step(unf, 4, [f.copy() for f in F0], nrm, "hals", 2, [None] * 3, [], [False] * 3,
                              alpha=math.inf)
    assert abs(c - co) <= 2e-3 * co
    for i in range(3):
        assert rel(F[i], Fo[i]) < 2e-3
    with pytest.raises(err.CustomNotEngouhFactors):
        ntf(T, 4, init="custom", factors_0=F0[:2])
    with pytest.raises(err.CustomNotValidFactors):
        ntf(T, 4, init="custom", factors_0=[F0[0], None, F0[2]])
    with pytest.raises(err.InvalidArgumentValue):
        one_ntf_step(unf, 4, F0, nrm, "hals", 1, [None] * 3, [], [False] * 3)
    # default alpha = 0.5: wall-clock dependent like the reference; must run and return finite costs
    out = ntf(T, 4, init="custom", factors_0=F0, n_iter_max=3, return_costs=True)
    assert np.isfinite(out[1]).all()


def test_ntf_mid_size_vs_oracle(built_lib):
    """120 x 100 x 80 rank 12 against the fp64 oracle (deterministic HALS)."""
    from nn_fac_amd.ntf import compute_ntf
    T, F0 = orc.synth_ntf((120, 100, 80), 12, seed=2, dtype=np.float32)
    sw, swo = [], []
    F, costs, _ = compute_ntf(T, 12, F0, n_iter_max=4, tol=0, update_rule="hals", return_costs=True, alpha=math.inf,
                              sparsity_coefficients=[None] * 3, normalize=[False] * 3, sweep_log=sw)
    Fo, co, _ = orc.compute_ntf(T.astype(np.float64), 12, [f.astype(np.float64) for f in F0], n_iter_max=4, tol=0,
                                update_rule="hals", return_costs=True, alpha=math.inf, sweeps=swo)
    for i in range(3):
        assert rel(F[i], Fo[i]) < 2e-3
    np.testing.assert_allclose(costs, co, rtol=2e-3)
    assert sw == swo


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1)])
def test_ntf_early_stop_drops_the_speculative_iteration(built_lib, rule, beta):
    """compute_ntf keeps one iteration in flight ahead of the stopping test (run_ntf_steps): a stop at iteration k returns
    the factors of iteration k, bit for bit, and its costs -- bit for bit too where one kernel evaluates every cost (MU); a
    HALS run evaluates the two costs the stopping test fires on by the pass over T instead of the Gram identity
    (test_ntf_hals_cost_near_the_stopping_threshold), within 1e-4 of each other."""
    from nn_fac_amd.ntf import compute_ntf
    T, F0 = orc.synth_ntf((30, 25, 20), 4, seed=5, dtype=np.float32)
    kw = dict(update_rule=rule, beta=beta, alpha=math.inf, sparsity_coefficients=[None] * 3, normalize=[False] * 3,
              return_costs=True)
    _, costs, _ = compute_ntf(T, 4, F0, n_iter_max=12, tol=0, **kw)
    k = 5
    tol = 0.5 * (abs(costs[k - 1] - costs[k]) + abs(costs[k] - costs[k + 1]))
    first = next(i for i in range(1, len(costs)) if abs(costs[i - 1] - costs[i]) < tol)
    Fs, cs, toc = compute_ntf(T, 4, F0, n_iter_max=12, tol=tol, **kw)
    assert len(cs) == first + 1 == len(toc) and first < 11
    Fk, ck, _ = compute_ntf(T, 4, F0, n_iter_max=first + 1, tol=0, **kw)
    assert ck == costs[:first + 1]
    if rule == "mu":
        assert cs == ck
    else:
        assert cs[:first - 1] == ck[:first - 1]
        np.testing.assert_allclose(cs, ck, rtol=5e-4)
    for a, b in zip(Fs, Fk):
        assert np.array_equal(a, b)


def _spy_states(monkeypatch):
    from nn_fac_amd import ntf as ntf_mod
    made = []

    class Spy(ntf_mod._NtfState):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)
    monkeypatch.setattr(ntf_mod, "_NtfState", Spy)
    return made


@pytest.mark.parametrize("noise,expect_direct", [(0.3, False), (0.0, True)])
@pytest.mark.parametrize("shape,R", [((60, 50, 40), 6), ((9, 8, 7, 6), 3)])
def test_ntf_hals_cost_through_the_gram_identity_and_its_guard(built_lib, shape, R, noise, expect_direct, monkeypatch):
    """HALS loops take the cost from the reference's own expression (ntf.py:462-470) on the last mode's operands, inner
    products in fp64 (nnf_nmf_gram_cost_f32 with the two Grams of `cross`): no pass over T.  With a real residual that is what
    runs and agrees with the fp64 oracle and with the pass over T (NNF_COST=direct); on an almost exact fit the kernel's
    error estimate flags the iterate and the run goes on with the pass over T -- same factors either way."""
    from nn_fac_amd.ntf import compute_ntf
    rng = np.random.RandomState(11)
    true = [rng.rand(s, R) for s in shape]
    T = true[0]
    for f in true[1:]:                                   # the CP model, one mode at a time: (..., R) x (dim, R) -> (..., dim, R)
        T = T[..., None, :] * f
    T = T.sum(-1)
    T = (T + noise * rng.rand(*shape)).astype(np.float32)
    if noise:
        F0 = [rng.rand(s, R).astype(np.float32) for s in shape]
    else:
        F0 = [(f * (1 + 1e-3 * rng.rand(*f.shape))).astype(np.float32) for f in true]
    iters = 6
    kw = dict(n_iter_max=iters, tol=0, update_rule="hals", return_costs=True, alpha=math.inf)
    Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], **kw)
    made = _spy_states(monkeypatch)
    nm = len(shape)
    F, costs, _ = compute_ntf(T, R, F0, sparsity_coefficients=[None] * nm, normalize=[False] * nm, **kw)
    assert made[-1].direct_cost == expect_direct
    assert len(costs) == iters
    np.testing.assert_allclose(costs, co, rtol=2e-3, atol=2e-9)
    for a, b in zip(F, Fo):
        assert rel(a, b) < 2e-3
    monkeypatch.setenv("NNF_COST", "direct")
    F2, costs2, _ = compute_ntf(T, R, F0, sparsity_coefficients=[None] * nm, normalize=[False] * nm, **kw)
    for a, b in zip(F, F2):
        assert np.array_equal(a, b)                       # the cost evaluation never touches the factors
    np.testing.assert_allclose(costs, costs2, rtol=5e-4, atol=1e-10)


def test_ntf_hals_cost_near_the_stopping_threshold(built_lib, monkeypatch):
    """The identity cost carries an absolute error of ~1e-9 ||T||^2; the stopping test (ntf.py:337) compares a cost DIFFERENCE
    with `tol`.  Once two consecutive costs differ by `tol` give or take their error estimates, both are evaluated again by the
    pass over T (and every later one): the run stops at the iteration a run with NNF_COST=direct stops at, and the two costs
    the test fired on are bitwise those of that run."""
    from nn_fac_amd.ntf import compute_ntf
    T, F0 = orc.synth_ntf((40, 30, 20), 4, seed=3, dtype=np.float32)
    kw = dict(update_rule="hals", alpha=math.inf, sparsity_coefficients=[None] * 3, normalize=[False] * 3, return_costs=True)
    _, costs, _ = compute_ntf(T, 4, F0, n_iter_max=14, tol=0, **kw)
    k = 7
    tol = 0.5 * (abs(costs[k - 1] - costs[k]) + abs(costs[k] - costs[k + 1]))
    made = _spy_states(monkeypatch)
    Fs, cs, _ = compute_ntf(T, 4, F0, n_iter_max=14, tol=tol, **kw)
    assert made[-1].direct_cost
    monkeypatch.setenv("NNF_COST", "direct")
    Fd, cd, _ = compute_ntf(T, 4, F0, n_iter_max=14, tol=tol, **kw)
    assert len(cs) == len(cd) < 14
    assert cs[-2:] == cd[-2:]
    np.testing.assert_allclose(cs, cd, rtol=5e-4)          # (the guard's own bound: few terms, little averaging at this size)
    for a, b in zip(Fs, Fd):
        assert np.array_equal(a, b)
    _, co, _ = orc.compute_ntf(T.astype(np.float64), 4, [f.astype(np.float64) for f in F0], n_iter_max=14, tol=tol,
                               update_rule="hals", return_costs=True, alpha=math.inf)
    assert len(co) == len(cs)


@pytest.mark.parametrize("shape,R", [((7, 1, 5), 3), ((17, 5, 1), 1), ((6, 5, 4), 1), ((1, 33, 2), 4)])
@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1), ("mu", 2)])
def test_ntf_degenerate_dimensions(built_lib, shape, R, rule, beta):
    """Modes of length 1 and rank 1: k x 1 and 1 x k operands, whose size-1 stride PyTorch leaves arbitrary (found by
    tools/stress_tensor.py: the boundary refused them)."""
    from nn_fac_amd.ntf import compute_ntf
    T, F0 = orc.synth_ntf(shape, R, seed=sum(shape) + R, dtype=np.float32)
    kw = dict(n_iter_max=3, tol=0, update_rule=rule, beta=beta, alpha=math.inf, sparsity_coefficients=[None] * 3,
              normalize=[False] * 3, return_costs=True)
    F, costs, _ = compute_ntf(T, R, F0, **kw)
    Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], **kw)
    # a rank above a mode's length makes that mode's Hadamard Gram singular ((1, 33, 2) at rank 4: condition number 3e21) and
    # the factor is then only determined along the path the arithmetic takes: the bound for such a factor is what the
    # REFERENCE's own algorithm moves by when it runs in fp32 (the oracle on float32 inputs), times 3 -- 6e-3 there -- and
    # the usual 5e-4 wherever that is smaller
    F32, _, _ = orc.compute_ntf(T, R, [f.copy() for f in F0], **kw)
    for a, b, c in zip(F, Fo, F32):
        assert a.shape == b.shape and rel(a, b) < max(5e-4, 3 * rel(c.astype(np.float64), b))
    assert np.all(np.isfinite(costs))
    np.testing.assert_allclose(costs, co, rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("shape,R", [((12, 10, 8), 4), ((33, 65, 17), 7), ((64, 128, 300), 30), ((5, 700, 3), 2),
                                     ((301, 3, 129), 50)])
def test_mttkrp_from_shared_partial(built_lib, shape, R):
    """Dimension tree (nnf_ttm3_f32 mode 2 + nnf_mttkrp3_from_partial_f32): the mode-0 and mode-1 right-hand sides from ONE
    pass over T must equal unfolded[mode] @ khatri_rao (ntf.py:448-449) within the single-kernel tolerance."""
    from nn_fac_amd.engine import get_engine
    eng = get_engine("cuda:0")
    rng = np.random.RandomState(sum(shape) + R)
    T = rng.rand(*shape).astype(np.float32)
    F = [rng.rand(s, R).astype(np.float32) for s in shape]
    Td = torch.from_numpy(T).cuda()
    Ft = [torch.from_numpy(f.T.copy()).cuda() for f in F]
    Y = eng.ttm3(Td, Ft[2], 2)
    T64, F64 = T.astype(np.float64), [f.astype(np.float64) for f in F]
    np.testing.assert_allclose(Y.cpu().numpy(), np.einsum('ijk,kr->rij', T64, F64[2]), rtol=2e-5)
    got0 = eng.mttkrp3_from_partial(Y, Ft[1], 2).cpu().numpy().T
    got1 = eng.mttkrp3_from_partial(Y, Ft[0], 1).cpu().numpy().T
    for mode, got in ((0, got0), (1, got1)):
        want = orc.unfold(T64, mode) @ orc.khatri_rao(F64, skip_matrix=mode)
        assert np.linalg.norm(got - want) <= 1e-5 * np.linalg.norm(want), (mode, shape)


@pytest.mark.parametrize("shape,R", [((12, 10, 8), 4), ((33, 65, 17), 7), ((64, 128, 300), 30), ((5, 700, 3), 2),
                                     ((301, 3, 129), 50), ((40, 41, 42), 64), ((37, 29, 70), 18), ((90, 11, 65), 33),
                                     ((700, 130, 70), 17), ((41, 33, 129), 20), ((60, 7, 64), 19)])     # 16q+1..2 ranks: leftover ranks on the VALU pipe; 3/2-tile rounds
def test_fused_cost_and_partial(built_lib, shape, R):
    """nnf_cp3_partial_cost_f32: ONE pass over T gives ||T - [[F0,F1,F2]]||^2 (ntf.py:470) and the partial product
    Y = T x_2 F2^T of the next iteration -- against fp64 NumPy and against the two separate entry points."""
    from nn_fac_amd.engine import get_engine
    eng = get_engine("cuda:0")
    rng = np.random.RandomState(sum(shape) * R)
    F = [rng.rand(s, R).astype(np.float32) for s in shape]
    T = (np.einsum('ir,jr,kr->ijk', *F) * (1 + 0.05 * rng.randn(*shape))).astype(np.float32)   # near fit: small residual
    Td = torch.from_numpy(T).cuda()
    Ft = [torch.from_numpy(f.T.copy()).cuda() for f in F]
    Y = torch.empty((R, shape[0], shape[1]), dtype=torch.float32, device="cuda")
    cost = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.cp3_partial_cost(Td, Ft, Y, cost)
    T64, F64 = T.astype(np.float64), [f.astype(np.float64) for f in F]
    want = np.sum((T64 - np.einsum('ir,jr,kr->ijk', *F64)) ** 2)
    assert abs(float(cost) - want) <= 2e-5 * want, (float(cost), want)
    np.testing.assert_allclose(Y.cpu().numpy(), np.einsum('ijk,kr->rij', T64, F64[2]), rtol=2e-5)
    assert abs(float(cost) - 2 * float(eng.cp3_betadiv(Td, Ft, 2))) <= 1e-5 * want


@pytest.mark.parametrize("shape,R", [((9, 8, 7, 6), 3), ((5, 6, 4, 3, 7), 2), ((20, 3, 17, 11), 5)])
@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1), ("mu", 2)])
def test_ntf_order_n(built_lib, shape, R, rule, beta):
    """Tensors of order 4 and 5 (the reference loops over arbitrary modes, ntf.py:309-311,437-456): every mode's MTTKRP runs
    through the 3-way kernels on a grouped view of T (_NtfState.view3) -- factors, costs and sweep counts vs the oracle."""
    from nn_fac_amd.ntf import compute_ntf
    rng = np.random.RandomState(len(shape) * 100 + R)
    gen = [rng.rand(s, R) for s in shape]
    T = gen[0]
    for g in gen[1:]:
        T = T[..., None, :] * g
    T = (T.sum(axis=-1) + 1e-2 * rng.rand(*shape)).astype(np.float32)
    F0 = [rng.rand(s, R).astype(np.float32) + 0.01 for s in shape]
    N = len(shape)
    kw = dict(n_iter_max=4, tol=0, update_rule=rule, beta=beta, return_costs=True, alpha=math.inf,
              sparsity_coefficients=[None] * N, normalize=[False] * N)
    sw, swo = [], []
    F, costs, _ = compute_ntf(T, R, F0, sweep_log=sw, **kw)
    Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], sweeps=swo, **kw)
    tol = 2e-3 if rule == "hals" else 5e-5
    for i in range(N):
        assert F[i].shape == Fo[i].shape and rel(F[i], Fo[i]) < tol, (i, rel(F[i], Fo[i]))
    np.testing.assert_allclose(costs, co, rtol=2e-3 if rule == "hals" else 1e-4)
    assert sw == swo


@pytest.mark.parametrize("shape,R,fixed,sparsity,normalize", [
    ((30, 25, 20), 4, [2], [0.05, None, None], [False, False, False]),
    ((30, 25, 20), 4, [0], [None, None, None], [False, True, False]),
    ((30, 25, 20), 4, [], [None, 0.1, 0.02], [True, False, False]),
    ((12, 10, 9, 8), 3, [3], [None, 0.05, None, None], [False, False, True, False]),
    ((12, 10, 9, 8), 3, [0, 2], [None] * 4, [False] * 4),
])
def test_ntf_hals_options_against_oracle(built_lib, shape, R, fixed, sparsity, normalize, monkeypatch):
    """Fixed modes (the LAST UPDATED mode, whose operands carry the identity cost, is then not the last mode), sparsity terms
    (added to the identity cost afterwards: matrix 1-norms, ntf.py:466-470) and normalised factors, order 3 and 4: factors, costs
    and inner sweep counts against the fp64 oracle; the factors do not depend on how the cost is evaluated."""
    from nn_fac_amd.ntf import compute_ntf
    rng = np.random.RandomState(sum(shape) + R + len(fixed))
    true = [rng.rand(s, R) for s in shape]
    T = true[0]
    for f in true[1:]:
        T = T[..., None, :] * f
    T = (T.sum(-1) + 0.05 * rng.rand(*shape)).astype(np.float32)
    F0 = [rng.rand(s, R).astype(np.float32) + 0.1 for s in shape]
    kw = dict(n_iter_max=5, tol=0, update_rule="hals", return_costs=True, alpha=math.inf, fixed_modes=list(fixed),
              sparsity_coefficients=list(sparsity), normalize=list(normalize))
    sw, swo = [], []
    F, costs, _ = compute_ntf(T, R, F0, sweep_log=sw, **kw)
    kwo = dict(kw, sparsity_coefficients=list(sparsity), fixed_modes=list(fixed), normalize=list(normalize))
    Fo, co, _ = orc.compute_ntf(T.astype(np.float64), R, [f.astype(np.float64) for f in F0], sweeps=swo, **kwo)
    assert sw == swo
    np.testing.assert_allclose(costs, co, rtol=2e-3)
    for i, (a, b) in enumerate(zip(F, Fo)):
        assert rel(a, b) < 2e-3, i
        if i in fixed:
            assert np.array_equal(np.asarray(a), F0[i])
    monkeypatch.setenv("NNF_COST", "direct")
    F2, costs2, _ = compute_ntf(T, R, F0, **dict(kw, sparsity_coefficients=list(sparsity), fixed_modes=list(fixed),
                                                 normalize=list(normalize)))
    for a, b in zip(F, F2):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(costs, costs2, rtol=5e-4)
