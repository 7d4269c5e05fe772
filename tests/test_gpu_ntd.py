"""GPU parity: NTD pieces and drivers vs the oracle and the G7 fixtures (reference outputs).

Stated fp32 tolerances: mode products / single kernels rel <= 1e-5; projected-gradient core update (fp64 inside the
kernel, fp32 inputs) rel <= 1e-5 with the iteration count equal; NTD-HALS factors and core rel_fro <= 1e-3 after 4-10 outer
iterations, inner sweep counts equal; NTD-MU rel_fro <= 1e-4, cost rel <= 1e-4.  The NTD-HALS cost is the reference's
Gram-form expression (ntd.py:639), ~1e-8 of its terms on near-exact data: it is compared with an ABSOLUTE tolerance of
2e-6 on the normalised value (fp32 inputs cannot resolve more).
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import nnfac_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(built_lib):
    from nn_fac_amd.engine import get_engine
    return get_engine("cuda:0")


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda").contiguous()


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("shape,r", [((53, 85, 82), 9), ((7, 5, 3), 2), ((130, 33, 70), 17), ((64, 64, 64), 64), ((9, 300, 11), 5)])
def test_ttm3_all_modes(eng, shape, r):
    rng = np.random.RandomState(sum(shape) + r)
    T = rng.rand(*shape)
    for mode in range(3):
        F = rng.rand(shape[mode], r)
        want = orc.mode_dot(T, F.T, mode)                       # axis `mode` replaced by r, in place
        got = eng.ttm3(dev(T), dev(F.T.copy()), mode).cpu().numpy()
        if mode == 0:
            assert got.shape == (r, shape[1], shape[2])
        elif mode == 1:
            assert got.shape == (shape[0], r, shape[2])
        else:
            assert got.shape == (r, shape[0], shape[1])
            got = np.moveaxis(got, 0, 2)
        assert rel(got, want) < 1e-5


@pytest.mark.parametrize("dims,sparse,scale", [((9, 9, 3), 0.0, 1.0), ((4, 3, 2), 0.05, 1.0), ((16, 12, 20), 0.0, 0.1),
                                               ((1, 5, 1), 0.0, 1.0),
                                               ((20, 20, 20), 0.0, 0.1),    # fp32 storage in LDS, fp64 accumulation
                                               ((24, 24, 24), 0.0, 0.1)])   # does not fit in LDS: workspace path
def test_core_projected_gradient(eng, dims, sparse, scale):
    rng = np.random.RandomState(sum(dims))
    shape = tuple(5 * d + 3 for d in dims)
    F = [scale * rng.rand(shape[i], dims[i]) for i in range(3)]   # scale keeps the 6-decimal step away from 0
    core_true = rng.rand(*dims)
    T = orc.multi_mode_dot(core_true, F) + 0.01 * rng.rand(*shape)
    # fp32-rounded inputs on both sides
    MtX = orc.multi_mode_dot(T, F, transpose=True).astype(np.float32).astype(np.float64)
    M = [(f.T @ f).astype(np.float32).astype(np.float64) for f in F]
    core0 = rng.rand(*dims).astype(np.float32).astype(np.float64)
    step = 1.0
    for m_ in M:
        step *= 1 / np.linalg.svd(m_, compute_uv=False)[0]
    step = round(step, 6)
    core, cnt, upd0, upd = core0.copy(), 1, 0, 1
    while cnt <= 300 and upd >= 0.01 * upd0:
        grad = -MtX + orc.multi_mode_dot(core, M) + sparse * np.ones(core.shape)
        dc = np.minimum(step * grad, core)
        core = core - dc
        upd = np.sqrt(np.sum(dc ** 2))
        if cnt == 1:
            upd0 = upd
        cnt += 1
    nrm2 = float(np.sum(T ** 2))
    want_err = nrm2 - 2 * np.sum(MtX * core) + np.sum(orc.multi_mode_dot(core, M) * core)
    cd = dev(core0)
    st = eng.ntd_core_pg(cd, dev(MtX), [dev(m_) for m_ in M], sparse, 0.01, 300, nrm2).cpu().numpy()
    fp32_store = 4500 < core.size < 9500
    assert abs(int(st[0]) - (cnt - 1)) <= (3 if fp32_store else 0)      # the stop test compares fp32-rounded updates there
    assert abs(st[3] - step) <= 1e-12
    assert rel(cd.cpu().numpy(), core) < (1e-4 if fp32_store else 1e-5)
    assert abs(st[4] - want_err) <= 1e-6 * nrm2


def _small(golden):
    g = golden("g7_ntd.npz")
    return g, g["small_T"], g["small_core0"], [g[f"small_F0_{i}"] for i in range(3)]


@pytest.mark.parametrize("name,sp,fixed,norm,mcn", [("plain", [None] * 4, [], [False] * 4, None),
                                                    ("sparse", [0.01, None, 0.02, 0.05], [], [False] * 4, None),
                                                    ("norm", [None] * 4, [], [True, False, True, True], 1),
                                                    ("fixed1", [None] * 4, [1], [False] * 4, None)])
def test_one_ntd_step_against_reference_fixture(golden, name, sp, fixed, norm, mcn):
    from nn_fac_amd.ntd import one_ntd_step
    g, T, core, f = _small(golden)
    rk = list(core.shape)
    nrm = np.sqrt(np.sum(T ** 2))
    costs = []
    for _ in range(4):
        core, f, c = one_ntd_step(T, rk, core, f, nrm, list(sp), list(fixed), list(norm), mcn, alpha=math.inf, delta=0.01)
        costs.append(c)
    assert rel(core, g[f"small_hals_{name}_core"]) < 1e-3
    for i in range(3):
        assert rel(f[i], g[f"small_hals_{name}_F{i}"]) < 1e-3
    np.testing.assert_allclose(costs, g[f"small_hals_{name}_costs"], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("beta", [0, 0.5, 1, 2, 3])
def test_one_ntd_step_mu_and_mu_tensorial_against_reference_fixture(golden, beta):
    from nn_fac_amd.ntd import one_ntd_step_mu
    from nn_fac_amd.update_rules.mu import mu_tensorial
    g, T, core, f = _small(golden)
    rk = list(core.shape)
    assert rel(mu_tensorial(core, f, T, beta), g[f"small_mut_b{beta}"]) < 2e-5
    costs = []
    for _ in range(4):
        core, f, c = one_ntd_step_mu(T, rk, core, f, beta, None, [], [False] * 4, None)
        costs.append(c)
    assert rel(core, g[f"small_mu_b{beta}_core"]) < 1e-4
    for i in range(3):
        assert rel(f[i], g[f"small_mu_b{beta}_F{i}"]) < 1e-4
    np.testing.assert_allclose(costs, g[f"small_mu_b{beta}_costs"], rtol=1e-4)


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 2), ("mu", 1), ("mu", 0)])
def test_ntd_on_the_reference_test_problem(golden, rule, beta):
    """The configuration of the reference's known-answer tests (NTD_tests.py:138-255): 53x85x82, ranks (9,9,3), random
    init with seed 0, 10 iterations -- against the reference's full outputs stored in G7."""
    from nn_fac_amd.ntd import ntd
    from test_oracle_golden import ntd_reference_tensor
    g = golden("g7_ntd.npz")
    shape, ranks = tuple(int(x) for x in g["ref_shape"]), [int(x) for x in g["ref_ranks"]]
    T = ntd_reference_tensor(shape, ranks)
    core, facs, costs, toc = ntd(T, list(ranks), init="random", n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                 sparsity_coefficients=[None] * 4, fixed_modes=[], normalize=[False] * 4, verbose=False,
                                 return_costs=True, deterministic=True, seed=0)
    tag = f"ref_{rule}_b{beta}"
    tol = 1e-3 if rule == "hals" else 1e-4
    assert rel(core, g[f"{tag}_core"]) < tol
    for i in range(3):
        assert rel(facs[i], g[f"{tag}_F{i}"]) < tol
    want = g[f"{tag}_costs"]
    if rule == "hals":
        assert len(costs) == len(want)
        np.testing.assert_allclose(costs, want, rtol=0, atol=2e-6)
    else:
        np.testing.assert_allclose(costs, want, rtol=1e-4)


def test_tucker_hooi_against_the_oracle():
    """The device HOSVD + HOOI (utils/initialize_factors.tucker_hooi, fp64) against the oracle's restatement of tensorly's
    tucker, compared after the absolute values the only caller takes (signs of singular vectors are arbitrary)."""
    from nn_fac_amd.utils.initialize_factors import tucker_hooi
    from test_oracle_golden import ntd_reference_tensor, NTD_TESTS_SHAPE, NTD_TESTS_RANKS
    cases = [(ntd_reference_tensor(NTD_TESTS_SHAPE, NTD_TESTS_RANKS), list(NTD_TESTS_RANKS)),
             (np.random.RandomState(3).rand(12, 9, 8, 7), [3, 4, 2, 3]),
             (np.random.RandomState(4).rand(400, 6, 5), [4, 3, 2])]          # Gram route of the thin SVD (400 >> 30)
    for T, ranks in cases:
        want_c, want_f = orc.tucker_hooi(T, ranks)
        got_c, got_f = tucker_hooi(T, ranks)
        assert rel(np.abs(got_c), np.abs(want_c)) < 1e-8
        for a, b in zip(got_f, want_f):
            assert rel(np.abs(a), np.abs(b)) < 1e-8
        tc, tf = tucker_hooi(torch.tensor(T, dtype=torch.float32, device="cuda"), ranks)      # device in -> device out
        assert tc.is_cuda and tc.dtype == torch.float32 and rel(tc.abs().cpu().numpy(), np.abs(want_c)) < 1e-4


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 2)])
def test_ntd_tucker_init_known_answers(rule, beta):
    """ntd(init="tucker") on the reference's test problem against the reference's literal known answers
    (NTD_tests.py:157-175 HALS, :197-215 MU beta=2) at the file's fp32 tolerances, and against the oracle's full outputs."""
    from nn_fac_amd.ntd import ntd
    from test_oracle_golden import ntd_reference_tensor, NTD_TESTS_SHAPE, NTD_TESTS_RANKS, TUCKER_INIT_KNOWN
    T, ranks = ntd_reference_tensor(NTD_TESTS_SHAPE, NTD_TESTS_RANKS), list(NTD_TESTS_RANKS)
    core, facs, costs, toc = ntd(T, list(ranks), init="tucker", n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                 sparsity_coefficients=[None] * 4, fixed_modes=[], normalize=[False] * 4, verbose=False,
                                 return_costs=True, deterministic=True, seed=0)
    c0, f0 = orc.ntd_tucker_init(T, ranks)
    wc, wf, wcosts, _ = orc.compute_ntd(T, ranks, c0, f0, n_iter_max=10, tol=1e-8, update_rule=rule, beta=beta,
                                        sparsity_coefficients=[None] * 4, normalize=[False] * 4, return_costs=True,
                                        deterministic=True)
    tol = 1e-3 if rule == "hals" else 1e-4
    assert rel(core, wc) < tol
    for i in range(3):
        assert rel(facs[i], wf[i]) < tol
    k = TUCKER_INIT_KNOWN[(rule, beta)]
    got = (facs[0][0][0], facs[1][0][0], facs[2][0][0], core[0, 0, 0])
    for a, b in zip(got, k[:4]):
        assert abs(a - b) <= 10 * tol * abs(b)
    if rule == "hals":
        assert len(costs) == len(wcosts)
        np.testing.assert_allclose(costs, wcosts, rtol=0, atol=2e-6)
        assert abs(costs[0] - k[4]) < 2e-6 and abs(costs[-1] - k[5]) < 2e-6
    else:
        np.testing.assert_allclose(costs, wcosts, rtol=1e-4)
        assert abs(costs[0] - k[4]) < 1e-4 * k[4] and abs(costs[-1] - k[5]) < 1e-4 * k[5]


def test_ntd_chromas_init_fixes_identity_first_factor():
    """initialize_factors.py:77-80 + ntd.py:240-241: Tucker start, W = I12 and mode 0 fixed."""
    from nn_fac_amd.ntd import ntd
    T = np.random.RandomState(5).rand(12, 10, 9)
    core, facs = ntd(T, [12, 4, 3], init="chromas", n_iter_max=3, tol=1e-8, sparsity_coefficients=[None] * 4,
                     fixed_modes=[], normalize=[False] * 4, deterministic=True, seed=0)
    np.testing.assert_array_equal(np.asarray(facs[0]), np.identity(12))
    c0, f0 = orc.ntd_tucker_init(T, [12, 4, 3])
    f0[0] = np.identity(12)
    wc, wf = orc.compute_ntd(T, [12, 4, 3], c0, f0, n_iter_max=3, tol=1e-8, sparsity_coefficients=[None] * 4,
                             fixed_modes=[0], normalize=[False] * 4, deterministic=True)
    assert rel(core, wc) < 1e-3 and rel(facs[1], wf[1]) < 1e-3 and rel(facs[2], wf[2]) < 1e-3


def test_ntd_argument_errors():
    """Raise sites of ntd.py:213,228,232,234 and initialize_factors.py:83."""
    from nn_fac_amd.ntd import ntd
    from nn_fac_amd.utils import errors as err
    T = np.random.RandomState(0).rand(6, 5, 4)
    kw = dict(sparsity_coefficients=[None] * 4, normalize=[False] * 4)
    with pytest.raises(err.InvalidRanksException):
        ntd(T, [3, 4], init="random", **kw)
    with pytest.raises(err.InvalidInitializationType):
        ntd(T, [2, 4, 3], init="string", **kw)
    f = [np.ones((6, 2)), np.ones((5, 2)), np.ones((4, 2))]
    with pytest.raises(err.CustomNotEngouhFactors):
        ntd(T, [2, 2, 2], init="custom", factors_0=f[:2], **kw)
    with pytest.raises(err.CustomNotValidFactors):
        ntd(T, [2, 2, 2], init="custom", factors_0=[f[0], f[1], None], **kw)
    with pytest.raises(err.CustomNotValidCore):
        ntd(T, [2, 2, 2], init="custom", factors_0=f, core_0=None, **kw)


def test_ntd_wall_clock_rule_runs_and_decreases():
    """deterministic=False keeps the reference's time-dependent sweep budget (alpha = 0.5, ntd.py:399-400): results vary
    from run to run by design; the cost must still go down and the outputs keep shape and sign."""
    from nn_fac_amd.ntd import ntd
    rng = np.random.RandomState(3)
    shape, ranks = (40, 30, 20), [5, 4, 3]
    F = [rng.rand(shape[i], ranks[i]) for i in range(3)]
    T = orc.multi_mode_dot(rng.rand(*ranks), F) + 0.01 * rng.rand(*shape)
    core, facs, costs, toc = ntd(T, list(ranks), init="random", n_iter_max=6, tol=0, sparsity_coefficients=[None] * 4,
                                 normalize=[False] * 4, return_costs=True, deterministic=False)
    assert core.shape == tuple(ranks) and [f.shape for f in facs] == [(shape[i], ranks[i]) for i in range(3)]
    assert all(np.all(f >= 0) for f in facs) and np.all(core >= 0)
    assert costs[-1] < costs[0] and len(toc) == len(costs) == 6


@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1)])
def test_ntd_early_stop_drops_the_speculative_iteration(built_lib, rule, beta):
    """compute_ntd keeps one iteration in flight ahead of the stopping test: a stop at iteration k returns the core,
    factors and costs of iteration k, bit for bit."""
    from nn_fac_amd.ntd import compute_ntd
    rng = np.random.RandomState(3)
    shape, ranks = (20, 18, 16), (4, 3, 2)
    F = [rng.rand(s, q) for s, q in zip(shape, ranks)]
    T = (np.einsum('abc,ia,jb,kc->ijk', rng.rand(*ranks), *F) + 1e-2 * rng.rand(*shape)).astype(np.float32)
    C0 = rng.rand(*ranks).astype(np.float32)
    F0 = [rng.rand(s, q).astype(np.float32) for s, q in zip(shape, ranks)]
    kw = dict(update_rule=rule, beta=beta, sparsity_coefficients=[None] * 4, normalize=[False] * 4, return_costs=True,
              deterministic=True)
    _, _, costs, _ = compute_ntd(T, ranks, C0, F0, n_iter_max=12, tol=0, **kw)
    k = 5
    tol = 0.5 * (abs(costs[k - 1] - costs[k]) + abs(costs[k] - costs[k + 1]))
    first = next(i for i in range(1, len(costs)) if abs(costs[i - 1] - costs[i]) < tol)
    Cs, Fs, cs, toc = compute_ntd(T, ranks, C0, F0, n_iter_max=12, tol=tol, **kw)
    assert len(cs) == first + 1 == len(toc) and first < 11
    Ck, Fk, ck, _ = compute_ntd(T, ranks, C0, F0, n_iter_max=first + 1, tol=0, **kw)
    assert cs == ck == costs[:first + 1]
    assert np.array_equal(Cs, Ck)
    for a, b in zip(Fs, Fk):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shape,ranks", [((9, 8, 7, 6), (3, 2, 3, 2)), ((6, 5, 4, 5, 4), (2, 2, 2, 3, 2)), ((20, 6, 11, 9), (4, 3, 5, 2))])
@pytest.mark.parametrize("rule,beta", [("hals", 2), ("mu", 1), ("mu", 2)])
def test_ntd_order_n(built_lib, shape, ranks, rule, beta):
    """Tensors of order 4 and 5 (the reference loops over arbitrary modes, ntd.py:534-557): every mode product is the 3-way
    kernel on a view (left, I_n, right), the core update merges the trailing core modes (Kronecker Gram) -- core, factors,
    costs, sweep and projected-gradient counts vs the oracle."""
    from nn_fac_amd.ntd import compute_ntd
    rng = np.random.RandomState(sum(shape) + sum(ranks))
    N = len(shape)
    G = rng.rand(*ranks)
    Fs = [rng.rand(s, q) for s, q in zip(shape, ranks)]
    T = orc.multi_mode_dot(G, Fs) + 1e-2 * rng.rand(*shape)
    T = T.astype(np.float32)
    C0 = (rng.rand(*ranks) + 0.05).astype(np.float32)
    F0 = [(rng.rand(s, q) + 0.05).astype(np.float32) for s, q in zip(shape, ranks)]
    kw = dict(n_iter_max=3, tol=0, update_rule=rule, beta=beta, return_costs=True, deterministic=True,
              sparsity_coefficients=[None] * (N + 1), normalize=[False] * (N + 1))
    sw, pg, swo, pgo = [], [], [], []
    core, F, costs, _ = compute_ntd(T, list(ranks), C0, F0, sweep_log=sw, pg_log=pg, **kw)
    co, Fo, cso, _ = orc.compute_ntd(T.astype(np.float64), list(ranks), C0.astype(np.float64), [f.astype(np.float64) for f in F0],
                                     sweeps=swo, pg_iters=pgo, **kw)
    tol = 5e-3 if rule == "hals" else 1e-4
    assert core.shape == co.shape and rel(core, co) < tol, rel(core, co)
    for i in range(N):
        assert rel(F[i], Fo[i]) < tol, (i, rel(F[i], Fo[i]))
    np.testing.assert_allclose(costs, cso, rtol=5e-3 if rule == "hals" else 2e-4)
    if rule == "hals":
        assert sw == swo and pg == pgo, (sw, swo, pg, pgo)


def test_core_update_on_one_workgroup_per_slab(built_lib, monkeypatch):
    """nnf_ntd_core_pg_f32 for a core of a few thousand entries (16 x 12 x 12 = 2304): the projected-gradient loop of
    ntd.py:588-619 runs on one workgroup per mode-0 slab with one grid barrier per step -- same iteration count, step, core
    and error as the one-workgroup form (NNF_NTD_PG_MULTI=0; both fp64, the mode products in a different order) and as the
    oracle's loop."""
    import subprocess
    import json
    code = r"""
import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
rng = np.random.RandomState(3)
d = (16, 12, 12)
F = [rng.rand(40 + 5 * i, d[i]) for i in range(3)]
M = [f.T @ f for f in F]
core = rng.rand(*d) + 0.05
MtX = np.einsum('abc,ia,jb,kc->ijk', rng.rand(*d), *M)[:d[0], :d[1], :d[2]] if False else None
G = rng.rand(*d)
MtX = np.einsum('abc,xa,yb,zc->xyz', G, *M)          # MtX of an exactly representable tensor: the loop converges towards G
c = torch.from_numpy(core.astype(np.float32)).cuda().contiguous()
st = eng.ntd_core_pg(c, torch.from_numpy(MtX.astype(np.float32)).cuda(), [torch.from_numpy(m.astype(np.float32)).cuda() for m in M],
                     0.01, 0.01, 300, 12345.0).cpu().tolist()
print(json.dumps({"status": st, "core": c.cpu().double().numpy().ravel().tolist()}))
"""
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, NNF_NTD_PG_MULTI=flag)
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    multi, single = outs
    assert multi["status"][5] == 0 and int(multi["status"][0]) == int(single["status"][0]) and int(multi["status"][0]) >= 3
    assert multi["status"][3] == single["status"][3]                     # the rounded step
    np.testing.assert_allclose(multi["status"][1:3], single["status"][1:3], rtol=1e-9)
    np.testing.assert_allclose(multi["status"][4], single["status"][4], rtol=1e-9)
    np.testing.assert_allclose(multi["core"], single["core"], rtol=1e-6, atol=1e-9)
    # and the oracle's loop on the same (fp32-rounded) operands
    rng = np.random.RandomState(3)
    d = (16, 12, 12)
    F = [rng.rand(40 + 5 * i, d[i]) for i in range(3)]
    M = [(f.T @ f).astype(np.float32).astype(np.float64) for f in F]
    core = (rng.rand(*d) + 0.05).astype(np.float32).astype(np.float64)
    G = rng.rand(*d)
    MtX = np.einsum('abc,xa,yb,zc->xyz', G, *[f.T @ f for f in F]).astype(np.float32).astype(np.float64)
    step = 1.0
    for m in M:
        step *= 1 / np.linalg.svd(m, compute_uv=False)[0]
    step = round(step, 6)
    cnt, upd0, upd = 1, 0, 1
    while cnt <= 300 and upd >= 0.01 * upd0:
        grad = -MtX + orc.multi_mode_dot(core, M) + 0.01
        dc = np.minimum(step * grad, core)
        core = core - dc
        upd = np.sqrt(np.sum(dc ** 2))
        if cnt == 1:
            upd0 = upd
        cnt += 1
    assert int(multi["status"][0]) == cnt - 1 and abs(multi["status"][3] - step) < 1e-12
    assert rel(np.array(multi["core"]).reshape(d), core) < 1e-5
