"""Parity of the HIP kernels (through the C ABI) against the CPU oracle / fp64 NumPy.  Needs a MI355X."""
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


# shapes: (m, n, r) -- aligned, ragged, tiny, rank not a multiple of 16/4, one column / one row
SHAPES = [(200, 100, 10), (73, 25, 9), (1000, 260, 50), (513, 130, 33), (64, 64, 16), (300, 7, 3), (5, 300, 2),
          (257, 1, 1), (1, 257, 1), (2000, 500, 50), (777, 333, 100), (130, 70, 128), (4096, 1024, 64),
          (1000, 260, 52), (513, 132, 51), (900, 128, 36), (600, 64, 20), (700, 96, 68)]   # leftover-rank forms 16q + 3..4


@pytest.mark.parametrize("m,n,r", SHAPES)
def test_gram_xty_xht_frob(eng, m, n, r):
    rng = np.random.RandomState(m * 7 + n * 3 + r)
    X = rng.rand(m, n).astype(np.float32)
    Ut = rng.rand(r, m).astype(np.float32)
    V = rng.rand(r, n).astype(np.float32)
    Xd, Utd, Vd = dev(X), dev(Ut), dev(V)
    X64, U64, V64 = X.astype(np.float64), Ut.astype(np.float64), V.astype(np.float64)
    # single-call tolerance 1e-5 relative; fp32 MFMA chains with fp64 slab reduction do much better
    assert rel(eng.gram(Vd).cpu().numpy(), V64 @ V64.T) < 1e-5
    assert rel(eng.gram(Utd).cpu().numpy(), U64 @ U64.T) < 1e-5
    assert rel(eng.xty(Xd, Utd).cpu().numpy(), U64 @ X64) < 1e-5
    assert rel(eng.xht(Xd, Vd).cpu().numpy(), V64 @ X64.T) < 1e-5
    want = np.sum((X64 - U64.T @ V64) ** 2)
    got = float(eng.frob_resid(Xd, Utd, Vd))
    assert abs(got - want) <= 1e-5 * want


@pytest.mark.parametrize("r,K", [(50, 100000), (100, 40000), (100, 300000), (12, 700), (64, 1024), (33, 513), (200, 30000), (1, 5)])
def test_gram_with_its_fp64_sums(eng, r, K):
    """nnf_gram_f64_f32: the fp32 Gram is unchanged (bitwise the plain entry's) and the fp64 copy is the same sums before their
    rounding -- closer to the exact Gram than fp32 storage allows (3.4e-8 relative rms) wherever split-K slabs exist."""
    rng = np.random.RandomState(r * 7 + K)
    A = rng.rand(r, K).astype(np.float32)
    Ad = dev(A)
    G64 = torch.empty((r, r), dtype=torch.float64, device="cuda")
    G = eng.gram(Ad, out64=G64).cpu().numpy()
    assert np.array_equal(G, eng.gram(Ad).cpu().numpy())
    g64 = G64.cpu().numpy()
    assert np.array_equal(g64.astype(np.float32), G)
    exact = A.astype(np.float64) @ A.astype(np.float64).T
    e64 = np.sqrt(np.mean(((g64 - exact) / exact) ** 2))
    assert e64 < 6e-7                     # (K <= 1024: one workgroup, one fp32 chain per entry, no slabs)
    print(r, K, "relative rms error of the fp64 copy:", e64)
    if K > 1024:
        assert e64 < (2e-8 if r <= 128 else 6e-8), e64            # (accumulation inside a split of at most 512 columns only)


def test_gram_identity_cost_on_the_fp64_gram(eng):
    """nnf_nmf_gram_cost_g64_f32 against the fp64 residual on a late-run-like iterate (cost / ||X||^2 ~ 1e-3 at 20000 x 600 rank
    40): same cost as the fp32-Gram form within both estimates, a smaller estimate, and an error inside its own estimate."""
    rng = np.random.RandomState(3)
    m, n, r = 20000, 600, 40
    U, V = rng.rand(m, r), rng.rand(r, n)
    X = (U @ V * (1 + 0.03 * rng.randn(m, n))).astype(np.float32)
    Xd, Utd, Vd = dev(X), dev(U.T), dev(V)
    UtM = eng.xty(Xd, Utd)
    G64 = torch.empty((r, r), dtype=torch.float64, device="cuda")
    UtU = eng.gram(Utd, out64=G64)
    nx2 = eng.dot(Xd, Xd)
    o32 = torch.zeros(3, dtype=torch.float64, device="cuda")
    o64 = torch.zeros(3, dtype=torch.float64, device="cuda")
    eng.gram_cost(Vd, UtM, UtU, nx2, o32, rounding=(6e-8, 0.0))
    eng.gram_cost(Vd, UtM, UtU, nx2, o64, rounding=(6e-8, 0.0, 5e-9), UtU64=G64)
    want = np.sum((X.astype(np.float64) - Utd.cpu().numpy().astype(np.float64).T @ Vd.cpu().numpy().astype(np.float64)) ** 2)
    c32, f32_, e32 = o32.cpu().tolist()
    c64, f64_, e64 = o64.cpu().tolist()
    assert e64 < e32 and abs(c64 - want) <= e64 and abs(c32 - want) <= e32
    assert abs(c64 - want) <= 2e-5 * want and f64_ == 0.0


@pytest.mark.parametrize("K", [4, 60, 124, 128, 132, 500, 512, 772, 1000, 1024, 1028, 501])
@pytest.mark.parametrize("r", [1, 7, 16, 17, 30, 33, 48, 50, 64, 65])
def test_gram_of_short_factors(eng, K, r):
    """Factors of at most 1024 columns and rank <= 64 (NTF / NTD modes) take the one-workgroup Gram whose waves read their
    fragments straight from global memory (nnf_gram_small_kernel); K % 4 != 0, K > 1024 and r > 64 take the chunked
    one.  Every k-step count (1..8 per wave), ragged last step, rank tiles 1..4, a padded row stride, NaNs in the padding."""
    rng = np.random.RandomState(K * 131 + r)
    ld = K + 4 * (r % 3)
    buf = np.full((r + 1, ld), np.nan, dtype=np.float32)
    buf[:r, :K] = rng.rand(r, K).astype(np.float32) - 0.25
    Ad = torch.from_numpy(buf).cuda()[:r, :K]
    A64 = buf[:r, :K].astype(np.float64)
    got = eng.gram(Ad).cpu().numpy()
    want = A64 @ A64.T
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() + 1e-30
    assert np.array_equal(got, eng.gram(Ad).cpu().numpy())      # same bits on a second call


@pytest.mark.parametrize("m", [70001, 98304, 100000, 131072, 131075, 300007])
@pytest.mark.parametrize("n,r", [(70, 50), (129, 64), (70, 30), (129, 18), (64, 16), (500, 20)])
def test_xht_row_tilings(eng, m, n, r):
    """X H^T picks its rows-per-workgroup from m (one balanced round of (3,2)- or (4,3)-tile workgroups, or several
    rounds of 256-row workgroups): every branch, with ragged ends, against an fp64 product on the device -- in both kernel
    forms (ranks <= 32 stage X through LDS in 256-byte row pieces, k_xht_lds.hip; the others read fragments directly)."""
    g = torch.Generator(device="cuda").manual_seed(m + n)
    X = torch.rand(m, n, device="cuda", generator=g)
    V = torch.rand(r, n, device="cuda", generator=g)
    want = V.double() @ X.double().t()
    got = eng.xht(X, V).double()
    assert float((got - want).norm() / want.norm()) < 1e-5
    assert float((got - want).abs().max() / want.abs().max()) < 1e-5      # no row block missed or doubled


@pytest.mark.parametrize("m,n,r", [(1000, 260, 30), (513, 130, 17), (64, 64, 16), (300, 7, 3), (5, 300, 2), (257, 1, 1),
                                   (1, 257, 1), (250000, 500, 30), (40000, 2000, 32), (777, 333, 20), (131075, 70, 19), (70001, 500, 17), (300, 36, 32)])
def test_xht_lds_staged_equals_direct_fragments(eng, m, n, r, monkeypatch):
    """The LDS-staged X H^T (k_xht_lds.hip) changes how X reaches the MFMA operands, not the arithmetic: same k order per
    accumulator as nnf_xht_kernel (NNF_XHT=direct) -- the results are equal bit for bit; padded rows with NaN in the padding."""
    g = torch.Generator(device="cuda").manual_seed(m * 3 + n + r)
    buf = torch.full((m, n + 4 * (r % 3)), float("nan"), device="cuda")
    buf[:, :n] = torch.rand(m, n, device="cuda", generator=g) - 0.25
    X = buf[:, :n]
    V = torch.rand(r, n, device="cuda", generator=g)
    got = eng.xht(X, V).clone()
    monkeypatch.setenv("NNF_XHT", "direct")
    want = eng.xht(X, V)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)


def test_xht_lds_staged_random_shapes(eng, monkeypatch):
    """Sixty random (m, n, r <= 32, pitch) draws -- every row-tile split (32- / 48- / 64-row waves, one round or several), aligned
    and unaligned pitches, ragged column tails of every length class -- LDS-staged against fragment-load kernel, bit for bit."""
    rng = np.random.RandomState(20261005)
    for case in range(60):
        m = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 127, 129, 1000, 4097, 33000, 70001, 98305, 140000]))
        n = int(rng.choice([1, 3, 4, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 500, 513]))
        r = int(rng.choice([1, 2, 3, 15, 16, 17, 18, 19, 20, 29, 30, 31, 32]))
        pad = int(rng.choice([0, 0, 4, 12, 28, 100]))
        g = torch.Generator(device="cuda").manual_seed(case)
        buf = torch.full((m, n + pad), float("nan"), device="cuda")
        buf[:, :n] = torch.rand(m, n, device="cuda", generator=g) - 0.25
        X = buf[:, :n]
        V = torch.rand(r, n, device="cuda", generator=g)
        monkeypatch.delenv("NNF_XHT", raising=False)
        got = eng.xht(X, V).clone()
        monkeypatch.setenv("NNF_XHT", "direct")
        want = eng.xht(X, V)
        assert torch.isfinite(got).all(), (m, n, r, pad)
        assert torch.equal(got, want), (m, n, r, pad)
        if case % 10 == 0:
            ref = V.double() @ X.double().t()
            assert float((got.double() - ref).norm() / ref.norm()) < 1e-5


def test_views_with_leading_dimension(eng):
    """Row-sharded / padded storage: ld > cols, base pointer not 16-byte aligned."""
    rng = np.random.RandomState(5)
    m, n, r = 300, 96, 20
    big = rng.rand(m + 3, n + 5).astype(np.float32)
    Xd = dev(big)[2:2 + m, 1:1 + n]            # ldx = n+5, offset by one float: unaligned
    X64 = big[2:2 + m, 1:1 + n].astype(np.float64)
    Ut = rng.rand(r, m).astype(np.float32)
    V = rng.rand(r, n).astype(np.float32)
    assert Xd.stride(0) == n + 5
    assert rel(eng.xty(Xd, dev(Ut)).cpu().numpy(), Ut.astype(np.float64) @ X64) < 1e-5
    assert rel(eng.xht(Xd, dev(V)).cpu().numpy(), V.astype(np.float64) @ X64.T) < 1e-5
    want = np.sum((X64 - Ut.astype(np.float64).T @ V.astype(np.float64)) ** 2)
    assert abs(float(eng.frob_resid(Xd, dev(Ut), dev(V))) - want) <= 1e-5 * want
    # aligned but padded rows (ldx multiple of 4)
    big2 = rng.rand(m, n + 32).astype(np.float32)
    Xd2 = dev(big2)[:, :n]
    assert rel(eng.xty(Xd2, dev(Ut)).cpu().numpy(), Ut.astype(np.float64) @ big2[:, :n].astype(np.float64)) < 1e-5
    assert rel(eng.xht(Xd2, dev(V)).cpu().numpy(), V.astype(np.float64) @ big2[:, :n].astype(np.float64).T) < 1e-5


def test_xty_is_bitwise_reproducible(eng):
    rng = np.random.RandomState(9)
    X, Ut = dev(rng.rand(5000, 300)), dev(rng.rand(50, 5000))
    a = eng.xty(X, Ut).clone()
    for _ in range(3):
        assert torch.equal(a, eng.xty(X, Ut))


def test_nan_in_padding_does_not_leak(eng):
    """Column padding (ld > n) may hold anything, including NaN: results must not see it."""
    rng = np.random.RandomState(2)
    m, n, r = 150, 70, 12
    big = np.full((m, n + 10), np.nan, dtype=np.float32)
    big[:, :n] = rng.rand(m, n)
    Xd = dev(big)[:, :n]
    V = rng.rand(r, n).astype(np.float32)
    Ut = rng.rand(r, m).astype(np.float32)
    X64 = big[:, :n].astype(np.float64)
    assert rel(eng.xht(Xd, dev(V)).cpu().numpy(), V.astype(np.float64) @ X64.T) < 1e-5
    assert rel(eng.xty(Xd, dev(Ut)).cpu().numpy(), Ut.astype(np.float64) @ X64) < 1e-5
    want = np.sum((X64 - Ut.astype(np.float64).T @ V.astype(np.float64)) ** 2)
    assert abs(float(eng.frob_resid(Xd, dev(Ut), dev(V))) - want) <= 1e-5 * want


def _kw(vec):
    kw = dict(maxiter=int(vec[0]), delta=float(vec[1]), alpha=math.inf)
    if vec[2] >= 0:
        kw["sparsity_coefficient"] = float(vec[2])
    kw["normalize"], kw["nonzero"] = bool(vec[3]), bool(vec[4])
    return kw


LAYOUTS = ["lane", "quad"]      # one lane per column (k_hals_fast.hip) / four lanes per column (k_hals_quad.hip)
# persistent solves also have the one-wave-per-column push form (k_hals_wave.hip; the default up to 8192 columns -- "wave"
# forces nothing, it names the default; fixed-count / snapshot launches keep the two layouts above)
# "mfma": the push form on the matrix cores (k_hals_mfma.hip, ranks 48..100; the default for many columns at ranks >= 64) --
# forced for every rank it covers, the default layout for the others
LAYOUTS_SOLVE = LAYOUTS + ["wave", "mfma"]


@pytest.mark.parametrize("layout", LAYOUTS_SOLVE)
def test_hals_against_reference_fixtures(golden, layout, monkeypatch):
    """hals_nnls_acc through the drop-in signature vs the reference outputs stored in g1 (57 cases), both kernels."""
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    monkeypatch.setenv("NNF_HALS_FORCE", layout)
    g = golden("g1_hals.npz")
    bad = []
    for c in range(int(g["ncases"])):
        s = int(g[f"c{c}_shape"])
        kw = _kw(g[f"c{c}_kw"])
        V, eps, cnt, rho = hals_nnls_acc(g[f"s{s}_UtM"], g[f"s{s}_UtU"], g[f"s{s}_Vin"], **kw)
        want = g[f"c{c}_V"]
        assert V.dtype == want.dtype and V.shape == want.shape
        e = rel(V, want)
        # sweep counts must match the reference; factors within the single-call fp32 tolerance (1e-5,
        # relaxed to 2e-4 for the 100-sweep cases where fp32 rounding accumulates over sweeps)
        if cnt != int(g[f"c{c}_cnt"]) or e > 2e-4 or abs(eps - float(g[f"c{c}_eps"])) > 2e-3 * abs(float(g[f"c{c}_eps"])) + 1e-12:
            bad.append((c, cnt, int(g[f"c{c}_cnt"]), e, eps, float(g[f"c{c}_eps"])))
    assert not bad, bad


@pytest.mark.parametrize("layout", LAYOUTS_SOLVE)
def test_hals_coupling_against_reference_fixtures(golden, layout, monkeypatch):
    """hals_coupling_nnls_acc (nnls.py:204-352, PARAFAC2's caller of the sweep) vs the real reference's outputs (g8):
    sweep counts equal, factors within the fp32 single-call tolerance, both kernel layouts."""
    from nn_fac_amd.update_rules.nnls import hals_coupling_nnls_acc
    monkeypatch.setenv("NNF_HALS_FORCE", layout)
    g = golden("g8_hals_coupling.npz")
    bad = []
    for c in range(int(g["ncases"])):
        s = int(g[f"c{c}_shape"])
        vec = g[f"c{c}_kw"]
        mu, zd = float(vec[0]), (None if vec[5] < 0 else int(vec[5]))
        kw = dict(maxiter=int(vec[1]), delta=float(vec[2]), normalize=bool(vec[3]), nonzero=bool(vec[4]), alpha=math.inf)
        G = g[f"s{s}_UtU"].copy()
        if zd is not None:
            G[zd, zd] = 0.0
        Vin = g[f"s{s}_Vin"].copy()
        V, eps, cnt, rho = hals_coupling_nnls_acc(g[f"s{s}_UtM"], G, Vin, g[f"s{s}_Vt"], mu, **kw)
        want = g[f"c{c}_V"]
        assert V.dtype == want.dtype and V.shape == want.shape and np.array_equal(Vin, g[f"s{s}_Vin"])
        e = rel(V, want)
        if cnt != int(g[f"c{c}_cnt"]) or e > 2e-4 or abs(eps - float(g[f"c{c}_eps"])) > 2e-3 * abs(float(g[f"c{c}_eps"])) + 1e-12:
            bad.append((c, cnt, int(g[f"c{c}_cnt"]), e, eps, float(g[f"c{c}_eps"])))
        if zd is not None:      # the row with a zero Gram diagonal is not touched although mu > 0 (nnls.py:316)
            np.testing.assert_array_equal(V[zd], Vin[zd].astype(np.float32).astype(np.float64))
    assert not bad, bad
    r = np.random.RandomState(0)
    G = r.rand(8, 8)
    G[2, 2] = 0
    with pytest.raises(ValueError):       # nnls.py:331-332
        hals_coupling_nnls_acc(r.rand(8, 8), G, r.rand(8, 8), r.rand(8, 8), 1.0, nonzero=True)


def test_hals_zero_column_raises():
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    from nn_fac_amd.utils import errors as err
    r = np.random.RandomState(0)
    G = r.rand(8, 8)
    G = G @ G.T
    G[2, 2] = 0
    hals_nnls_acc(r.rand(8, 8), G, r.rand(8, 8))                       # silently skipped (nnls_tests.py:37)
    with pytest.raises(err.ZeroColumnWhenUnautorized):
        hals_nnls_acc(r.rand(8, 8), G, r.rand(8, 8), nonzero=True)     # nnls_tests.py:38
    # vector right-hand side with a larger Gram (nnls_tests.py:44-45)
    V, eps, cnt, rho = hals_nnls_acc(r.rand(8, 1), r.rand(15, 15), r.rand(15, 1))
    assert V.shape == (15, 1)


@pytest.mark.parametrize("r", [12, 50, 64, 100])
@pytest.mark.parametrize("layout", LAYOUTS_SOLVE)
def test_hals_zero_diagonal_row_is_left_alone(eng, layout, r, monkeypatch):
    """nnls.py:160: a row whose Gram diagonal is 0 is skipped whatever it holds (negative entries included), and the
    other rows keep seeing its values through the off-diagonal Gram entries."""
    monkeypatch.setenv("NNF_HALS_FORCE", layout)
    rng = np.random.RandomState(11)
    n = 700
    A = rng.rand(4 * r, r)
    UtU = A.T @ A
    UtU[5, 5] = 0.0
    UtM, V0 = A.T @ rng.rand(4 * r, n), rng.rand(r, n)
    V0[5] = -rng.rand(n)
    Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM, UtU, V0, maxiter=30, alpha=math.inf, delta=0.01)
    Vd = dev(V0)
    st = eng.hals_solve(dev(UtM), dev(UtU), Vd, 30, delta=0.01).cpu()
    got = Vd.cpu().numpy()
    assert np.array_equal(got[5], V0[5].astype(np.float32))
    assert int(st[1]) == cnto and rel(got, Vo) < 2e-4


def test_hals_forced_wave_layout_is_refused_where_it_does_not_fit(eng, monkeypatch):
    """NNF_HALS_FORCE=wave pins the wave-per-column kernel: a solve it cannot hold resident is refused, not moved to another
    layout (so the layout tests above are known to have run the kernel they name)."""
    from nn_fac_amd.engine import EngineError
    monkeypatch.setenv("NNF_HALS_FORCE", "wave")
    rng = np.random.RandomState(2)
    A = rng.rand(200, 50)
    UtU, UtM, V = dev(A.T @ A), dev(A.T @ rng.rand(200, 60000)), dev(rng.rand(50, 60000))
    with pytest.raises(EngineError):
        eng.hals_solve(UtM, UtU, V, 10, delta=0.01)
    monkeypatch.delenv("NNF_HALS_FORCE")
    assert int(eng.hals_solve(UtM, UtU, V, 10, delta=0.01).cpu()[3]) == 0


def test_hals_does_not_modify_inputs(eng):
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    r = np.random.RandomState(1)
    A = r.rand(40, 6)
    UtU, UtM, V0 = dev(A.T @ A), dev(A.T @ r.rand(40, 30)), dev(r.rand(6, 30))
    keep = V0.clone()
    V, eps, cnt, rho = hals_nnls_acc(UtM, UtU, V0, maxiter=10, alpha=math.inf)
    assert torch.equal(V0, keep) and isinstance(V, torch.Tensor) and V.is_cuda and not torch.equal(V, V0)


@pytest.mark.parametrize("r,ncols,layout", [(50, 100000, "lane"), (100, 20000, "lane"), (30, 500, "lane"), (50, 300000, "lane"),
                                            (30, 500, "quad"), (50, 2000, "quad"), (96, 8000, "quad"), (70, 16000, "quad"), (100, 4000, "quad"), (128, 3000, "quad"),
                                            (30, 500, "wave"), (50, 2000, "wave"), (100, 4000, "wave"), (128, 3000, "wave"), (64, 8000, "auto"), (64, 4000, "wave"),
                                            (65, 700, "wave"), (3, 50, "wave"), (1, 9, "wave"), (57, 8192, "auto"), (57, 4500, "wave"),
                                            (100, 20000, "auto"), (120, 9000, "lane"), (64, 70000, "lane"), (56, 40000, "lane"), (34, 40000, "lane"),
                                            (100, 125000, "auto"), (96, 50000, "mfma"), (80, 40000, "auto"), (64, 70000, "auto"), (52, 40000, "mfma"),
                                            (50, 100000, "mfma"), (48, 20000, "mfma"), (100, 3000, "mfma"), (77, 40000, "auto"), (93, 33000, "auto")])
def test_hals_large_vs_oracle(eng, r, ncols, layout, monkeypatch):
    """Resident and strided (ncols > resident threads) persistent solves vs the fp64 oracle; sweep counts equal."""
    if layout != "auto":
        monkeypatch.setenv("NNF_HALS_FORCE", layout)
    else:
        monkeypatch.delenv("NNF_HALS_FORCE", raising=False)
    rng = np.random.RandomState(r + ncols)
    A = rng.rand(4 * r, r)
    cols = min(ncols, 4000)                     # oracle on a slice is not possible (global stop rule) -> tile the problem
    UtU = A.T @ A
    UtM_small = A.T @ (A @ rng.rand(r, cols) + 0.05 * rng.rand(4 * r, cols))
    reps = ncols // cols
    UtM = np.tile(UtM_small, (1, reps))
    V0 = np.tile(rng.rand(r, cols), (1, reps))
    log = []
    Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM_small, UtU, V0[:, :cols], maxiter=100, alpha=math.inf, delta=0.01,
                                          sweep_log=log)
    Vd = dev(V0)
    st = eng.hals_solve(dev(UtM), dev(UtU), Vd, 100, delta=0.01).cpu()
    assert int(st[3]) == 0
    assert int(st[1]) == cnto                    # tiling scales eps and eps0 alike: same decisions
    got = Vd.cpu().numpy()
    assert rel(got[:, :cols], Vo) < 2e-4
    assert np.array_equal(got[:, :cols], got[:, -cols:])      # identical columns -> identical results
    assert abs(float(st[0]) - reps * epso) <= 5e-3 * reps * epso + 1e-10 * reps * log[0]   # (+ fp32 noise of an exact fit: rank 1)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_hals_fixed_sweeps_mode(eng, layout, monkeypatch):
    monkeypatch.setenv("NNF_HALS_FORCE", layout)
    rng = np.random.RandomState(3)
    r, n = 20, 3000
    A = rng.rand(80, r)
    UtU, UtM, V0 = A.T @ A, A.T @ (A @ rng.rand(r, n)), rng.rand(r, n)
    log = []
    Vo, *_ = orc.hals_nnls_acc(UtM, UtU, V0, maxiter=6, alpha=math.inf, delta=0.0, sweep_log=log)
    Vd = dev(V0)
    nd = eng.hals_sweeps(dev(UtM), dev(UtU), Vd, 6).cpu().numpy()
    assert rel(Vd.cpu().numpy(), Vo) < 1e-4
    np.testing.assert_allclose(nd, log, rtol=5e-3)


MU_SHAPES = [(200, 100, 10), (73, 25, 9), (1000, 260, 50), (513, 130, 33), (300, 7, 3), (5, 300, 2), (2000, 500, 64),
             (260, 150, 18), (640, 200, 49), (333, 77, 17), (900, 300, 52),   # 16q+1..2 ranks: leftover ranks on the VALU pipe
             (500, 260, 20), (410, 130, 19), (300, 90, 36),                   # 17..20: four leftover ranks; 36: padded
             (700, 333, 65), (1500, 400, 100), (300, 200, 128)]     # r > 64: ratio kernel + plain contractions


@pytest.mark.parametrize("m,n,r", MU_SHAPES)
@pytest.mark.parametrize("beta", [0, 0.5, 1, 1.5, 2, 3, 4])
def test_mu_and_betadiv_kernels(eng, m, n, r, beta):
    rng = np.random.RandomState(m + n + r)
    U = rng.rand(m, r) + 0.05
    V = rng.rand(r, n) + 0.05
    X = rng.rand(m, r) @ rng.rand(r, n) + 0.05
    Xd, Utd, Vd = dev(X), dev(U.T), dev(V)
    X32, U32, V32 = (a.astype(np.float32).astype(np.float64) for a in (X, U, V))
    wantU = orc.mu_betadivmin(U32, V32, X32, beta)
    wantV = orc.switch_alternate_mu(X32, U32, V32, beta, "V")
    assert rel(eng.mu_left(Xd, Utd, Vd, beta).cpu().numpy().T, wantU) < 2e-5
    assert rel(eng.mu_right(Xd, Utd, Vd, beta).cpu().numpy(), wantV) < 2e-5
    want = orc.beta_divergence(X32, U32 @ V32, beta)
    got = float(eng.betadiv(Xd, Utd, Vd, beta))
    assert abs(got - want) <= 2e-5 * abs(want)


@pytest.mark.parametrize("r", [50, 40, 18])
@pytest.mark.parametrize("beta", [1, 1.5])
def test_mu_kernels_never_use_the_padding_between_rows(eng, r, beta):
    """Factors and data handed over as row-strided views whose padding holds NaN (the chunk images of the fused kernels are
    staged through a buffer resource that spans whole rows: columns past the matrix must be masked, 0 * NaN otherwise)."""
    m, n = 333, 203
    rng = np.random.RandomState(r)

    def padded(a, pad):
        t = torch.full((a.shape[0], a.shape[1] + pad), float("nan"), dtype=torch.float32, device="cuda")
        t[:, :a.shape[1]] = torch.tensor(a, dtype=torch.float32)
        return t[:, :a.shape[1]]
    U, V = rng.rand(m, r) + 0.05, rng.rand(r, n) + 0.05
    X = rng.rand(m, r) @ rng.rand(r, n) + 0.05
    for pad in (4, 5):                  # 16-byte aligned rows and not
        Xp, Utp, Vp = padded(X, pad), padded(U.T.copy(), pad), padded(V, pad)
        gotU = eng.mu_left(Xp, Utp, Vp, beta).cpu().numpy()
        gotV = eng.mu_right(Xp, Utp, Vp, beta).cpu().numpy()
        wantU = eng.mu_left(dev(X), dev(U.T), dev(V), beta).cpu().numpy()
        wantV = eng.mu_right(dev(X), dev(U.T), dev(V), beta).cpu().numpy()
        assert np.isfinite(gotU).all() and np.isfinite(gotV).all()
        assert rel(gotU, wantU) < 1e-6 and rel(gotV, wantV) < 1e-6


@pytest.mark.parametrize("m", [98304, 100000, 131072, 131100])
@pytest.mark.parametrize("beta", [1, 0.5])
def test_mu_left_row_tilings(eng, m, beta):
    """The left MU update picks its rows-per-workgroup from m like X H^T (one balanced round of 256- and 192-row
    workgroups when that covers the matrix, beta = 1): every branch against an fp64 evaluation of mu.py:84-97 on the device."""
    n, r = 70, 50
    g = torch.Generator(device="cuda").manual_seed(m)
    Ut = torch.rand(r, m, device="cuda", generator=g) + 0.05
    V = torch.rand(r, n, device="cuda", generator=g) + 0.05
    X = (torch.rand(m, r, device="cuda", generator=g) @ torch.rand(r, n, device="cuda", generator=g)) + 0.05
    U64, V64, X64 = Ut.double().t(), V.double(), X.double()
    K = U64 @ V64
    if beta == 1:
        want = torch.clamp(U64 * ((X64 / K) @ V64.t() / V64.sum(dim=1)), min=1e-12)
    else:
        want = torch.clamp(U64 * ((K ** (beta - 2) * X64) @ V64.t() / (K ** (beta - 1) @ V64.t())) ** orc.gamma_beta(beta),
                           min=1e-12)
    got = eng.mu_left(X, Ut, V, beta).double().t()
    assert float((got - want).norm() / want.norm()) < 2e-5
    assert float(((got - want).abs() / want).max()) < 1e-3       # no row block missed or doubled


@pytest.mark.parametrize("beta", [0, 0.5, 1, 1.5, 3])
def test_betadiv_with_tiny_data_entries(eng, beta):
    """Data entries of 1e-12 next to a model of order 1 (the clamped zeros of an NNDSVD start, the inner layers of
    multilayer NMF): 1 + (x-p)/p rounds to a tiny negative number in fp32 and a logarithm of it is NaN -- the terms are
    formed from the ratio x/p instead."""
    rng = np.random.RandomState(7)
    m, n, r = 300, 130, 6
    U, V = rng.rand(m, r) + 0.1, rng.rand(r, n) + 0.1
    X = U @ V * (1 + 0.1 * rng.randn(m, n)).clip(0.5, 1.5)
    X[rng.rand(m, n) < 0.2] = 1e-12
    X32, U32, V32 = (a.astype(np.float32).astype(np.float64) for a in (X, U, V))
    want = orc.beta_divergence(X32, U32 @ V32, beta)
    got = float(eng.betadiv(dev(X32), dev(U32.T), dev(V32), beta))
    assert np.isfinite(got) and abs(got - want) <= 5e-5 * abs(want), (beta, got, want)


def test_betadiv_near_convergence_has_no_cancellation(eng):
    """K ~ X: the naive fp32 form of KL/IS loses everything; the h(t) = t - log1p(t) form does not."""
    rng = np.random.RandomState(4)
    m, n, r = 400, 300, 8
    U, V = rng.rand(m, r) + 0.1, rng.rand(r, n) + 0.1
    U32, V32 = U.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64)
    X = (U32 @ V32) * (1 + 1e-3 * rng.randn(m, n))
    X32 = X.astype(np.float32).astype(np.float64)
    for beta in (0, 1, 1.5, 3):
        want = orc.beta_divergence(X32, U32 @ V32, beta)
        got = float(eng.betadiv(dev(X32), dev(U32.T), dev(V32), beta))
        assert abs(got - want) <= 2e-3 * want, (beta, got, want)


def test_mu_dropin_signatures(golden):
    from nn_fac_amd.update_rules.mu import mu_betadivmin, switch_alternate_mu
    from nn_fac_amd.utils.beta_divergence import beta_divergence, gamma_beta
    g = golden("g2_mu.npz")
    U, V, M = g["U"], g["V"], g["M"]
    for b in (0, 0.5, 1, 1.5, 2, 3, 4):
        b = int(b) if float(b).is_integer() else b
        gotU = switch_alternate_mu(M, U, V, b, "U")
        gotV = switch_alternate_mu(M, U, V, b, "H")
        assert isinstance(gotU, np.ndarray) and gotU.shape == U.shape and gotV.shape == V.shape
        assert rel(gotU, g[f"muU_b{b}"]) < 2e-5 and rel(gotV, g[f"muV_b{b}"]) < 2e-5
        assert rel(mu_betadivmin(U, V, M, b), g[f"muU_b{b}"]) < 2e-5
        assert abs(beta_divergence(M, U @ V, b) - float(g[f"div_b{b}"])) <= 1e-5 * abs(float(g[f"div_b{b}"]))
        assert gamma_beta(b) == float(g[f"gamma_b{b}"])


def test_hals_sweep_snapshots_at_bench_size(eng):
    """The row-sharded step of bench.py asks for snapshots of a 50 x 100000 factor: 391 workgroups, two per CU."""
    rng = np.random.RandomState(9)
    r, n = 50, 100000
    A = rng.rand(200, r)
    UtU, UtM, V0 = dev(A.T @ A), dev(A.T @ (A @ rng.rand(r, 2000))).repeat(1, 50), dev(rng.rand(r, 2000)).repeat(1, 50)
    snaps = torch.empty((3, r, n), dtype=torch.float32, device="cuda")
    V3 = V0.clone()
    nd = eng.hals_sweeps(UtM, UtU, V3, 3, snapshots=snaps)
    assert torch.equal(snaps[2], V3)
    V2 = V0.clone()
    nd2 = eng.hals_sweeps(UtM, UtU, V2, 2)
    assert torch.equal(snaps[1], V2) and torch.equal(nd2, nd[:2])


@pytest.mark.parametrize("r,n,sp", [(64, 40000, None), (100, 2000, 0.05), (50, 3000, None), (80, 700, None)])
def test_hals_mfma_fixed_sweeps_snapshots_and_chunks(eng, r, n, sp, monkeypatch):
    """The matrix-core layout in fixed-count mode: against the fp64 oracle, snapshot s == a run of s sweeps (bitwise), and chunks
    that hand the residual state on (nnf_hals_sweeps_ex_f32) == one launch of all the sweeps, bit for bit -- also across a
    scheduled from-scratch residual (sweep 32) and with the snapshot window starting inside the launch."""
    monkeypatch.setenv("NNF_HALS_FORCE", "mfma")
    rng = np.random.RandomState(r + n)
    A = rng.rand(3 * r, r)
    UtU, UtM, V0 = dev(A.T @ A), dev(A.T @ (A @ rng.rand(r, n) + 0.1 * rng.rand(3 * r, n))), dev(rng.rand(r, n))
    assert eng.hals_resid_floats(r, n) > 0
    log = []
    Vo, *_ = orc.hals_nnls_acc(UtM.cpu().numpy().astype(np.float64), UtU.cpu().numpy().astype(np.float64), V0.cpu().numpy().astype(np.float64),
                               maxiter=6, alpha=math.inf, delta=0.0, sparsity_coefficient=sp, sweep_log=log)
    snaps = torch.empty((4, r, n), dtype=torch.float32, device="cuda")
    V6 = V0.clone()
    nd = eng.hals_sweeps(UtM, UtU, V6, 6, sparsity=sp, snapshots=snaps, snap_first=2)
    assert rel(V6.cpu().numpy(), Vo) < 1e-4
    np.testing.assert_allclose(nd.cpu().numpy(), log, rtol=5e-3)
    assert torch.equal(snaps[3], V6)
    for k in (3, 5):
        Vk = V0.clone()
        ndk = eng.hals_sweeps(UtM, UtU, Vk, k, sparsity=sp)
        assert torch.equal(snaps[k - 3], Vk) and torch.equal(ndk, nd[:k])
    # chunks 5 + 29 + 4 (the second one crosses the scheduled residual of sweep 32) against 38 sweeps in one launch
    want = V0.clone()
    ndw = eng.hals_sweeps(UtM, UtU, want, 38, sparsity=sp)
    nf = eng.hals_resid_floats(r, n)
    sa, sb = torch.empty(nf, device="cuda"), torch.empty(nf, device="cuda")
    got = V0.clone()
    n1 = eng.hals_sweeps(UtM, UtU, got, 5, sparsity=sp, resid_out=sa)
    n2 = eng.hals_sweeps(UtM, UtU, got, 29, sparsity=sp, sweeps_done=5, resid_in=sa, resid_out=sb)
    n3 = eng.hals_sweeps(UtM, UtU, got, 4, sparsity=sp, sweeps_done=34, resid_in=sb, resid_out=sa)
    assert torch.equal(got, want) and torch.equal(torch.cat([n1, n2, n3]), ndw)
    # without the state a chunked run agrees to rounding only
    loose = V0.clone()
    eng.hals_sweeps(UtM, UtU, loose, 5, sparsity=sp)
    eng.hals_sweeps(UtM, UtU, loose, 33, sparsity=sp)
    assert rel(loose.cpu().numpy(), want.cpu().numpy()) < 2e-4


@pytest.mark.parametrize("layout", LAYOUTS)
def test_hals_sweep_snapshots(eng, layout, monkeypatch):
    """nnf_hals_sweeps_f32 with snapshots: block s must hold V after sweep s+1 (what a shorter run would return)."""
    monkeypatch.setenv("NNF_HALS_FORCE", layout)
    rng = np.random.RandomState(8)
    r, n = 24, 5000
    A = rng.rand(96, r)
    UtU, UtM, V0 = dev(A.T @ A), dev(A.T @ (A @ rng.rand(r, n))), dev(rng.rand(r, n))
    snaps = torch.empty((5, r, n), dtype=torch.float32, device="cuda")
    Vfull = V0.clone()
    nd = eng.hals_sweeps(UtM, UtU, Vfull, 5, snapshots=snaps)
    assert torch.equal(snaps[4], Vfull)
    for k in (1, 3):
        Vk = V0.clone()
        ndk = eng.hals_sweeps(UtM, UtU, Vk, k)
        assert torch.equal(snaps[k - 1], Vk)                    # bitwise: the kernels are deterministic
        assert torch.equal(ndk, nd[:k])


def test_hals_resident_columns_and_blocked_chunks(eng, monkeypatch):
    """A factor with more columns than the resident sweep kernel holds runs its blind chunks block by block (dist.column_blocks:
    config E on one, two or four devices).  With the capacity forced down to 768 columns a 50 x 5000 solve must give the factor
    and the counts of the persistent solve bit for bit, in both forms of the stopping decision; the real capacity is what the
    occupancy of the kernels says (two 256-column workgroups per CU at ranks 50 and 100 on this device)."""
    from nn_fac_amd import dist as nd
    # (the subject is the one-lane-per-column kernel whose resident capacity the blocks exist for; 5000 columns alone would
    # take the few-column layouts, which agree with it to rounding, not bit for bit)
    _blocked_chunks_case(eng, monkeypatch, "lane", 50)


def test_hals_blocked_chunks_on_the_matrix_core_layout(eng, monkeypatch):
    """The same protocol on k_hals_mfma.hip (rank 64): the chunks hand the residual state on, per column block."""
    _blocked_chunks_case(eng, monkeypatch, "mfma", 64)


def _blocked_chunks_case(eng, monkeypatch, layout, r):
    from nn_fac_amd import dist as nd
    monkeypatch.setenv("NNF_HALS_FORCE", layout)
    for rr in (50, 100):
        cap = eng.hals_resident_columns(rr)
        assert cap % 256 == 0 and 65536 <= cap <= 2048 * 256
    rng = np.random.RandomState(12)
    n = 5000
    A = rng.rand(200, r)
    UtU, UtM, V0 = dev(A.T @ A), dev(A.T @ (A @ rng.rand(r, n) + 0.3 * rng.rand(200, n))), dev(rng.rand(r, n))
    want = V0.clone()
    st = torch.zeros(8, dtype=torch.float64, device="cuda")
    eng.hals_solve(UtM, UtU, want, 100, delta=0.01, status=st)
    cnt_want = int(st[1].item())
    assert 4 < cnt_want < 101
    monkeypatch.setattr(eng, "hals_resident_columns", lambda rank: 768, raising=False)
    assert nd.column_blocks(eng, V0) == [(0, 768), (768, 1536), (1536, 2304), (2304, 3072), (3072, 3840), (3840, 4608), (4608, 5000)]
    for first in (3, cnt_want + 2, 60):                        # chunk too short, window on the stop, stop before the window
        got = V0.clone()
        eps, cnt, eps0 = nd.sharded_hals_solve(eng, UtM, UtU, got, None, nd.SweepGuess(first=first, max_chunk=104, window=4),
                                               budget=100, delta=0.01)
        assert cnt == cnt_want and torch.equal(got, want)
    got = V0.clone()
    status = torch.zeros(8, dtype=torch.float64, device="cuda")
    nd.sharded_hals_solve_async(eng, UtM, UtU, got, None, nd.SweepGuess(first=cnt_want + 1, max_chunk=104, window=4), status,
                                budget=100, delta=0.01)
    assert int(status[1].item()) == cnt_want and int(status[3].item()) == 0 and torch.equal(got, want)


@pytest.mark.parametrize("beta", [0.5, 1, 2, 3])
def test_mu_right_accumulate_then_apply(eng, beta):
    """Row-sharded right update: per-block numerator / denominator, summed, then applied == the one-shot kernel."""
    rng = np.random.RandomState(5)
    m, n, r = 3000, 260, 33
    X, Ut, V = dev(rng.rand(m, n) + 0.1), dev(rng.rand(r, m) + 0.05), dev(rng.rand(r, n) + 0.05)
    want = eng.mu_right(X, Ut, V, beta)
    cut = 1111
    parts = [eng.mu_right_accum(X[:cut], Ut[:, :cut], V, beta), eng.mu_right_accum(X[cut:], Ut[:, cut:], V, beta)]
    num = parts[0][0] + parts[1][0]
    den = parts[0][1] + parts[1][1] if parts[0][1] is not None else None
    dvec = parts[0][2] + parts[1][2] if parts[0][2] is not None else None
    got = eng.mu_apply(V, num, den, dvec, beta)
    assert rel(got.cpu().numpy(), want.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("rule,beta,extra", [("hals", 2, []), ("mu", 1, []), ("mu", 2, []), ("hals", 2, ["normalize"]),
                                             ("hals", 2, ["-", "stop"])])
def test_row_sharded_two_ranks_on_one_gpu(built_lib, rule, beta, extra):
    """Two gloo ranks sharing the GPU run the real sharded step (tools/dist_gpu_check.py) against the single-process run;
    "normalize": the sharded factor is normalised (row norms across both ranks, nnf_hals_row_update_f32 / _scale_f32);
    "stop": a run with `tol` between two cost differences -- the identity cost's near-threshold switch in a sharded run."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", os.path.join(root, "tools", "dist_gpu_check.py"), rule,
                          str(beta)] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert "DIST_GPU_CHECK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    if "stop" in extra:     # + the stopping test under the identity cost: both ranks switch to the direct cost together
        assert "DIST_GPU_STOP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("layout", LAYOUTS + ["generic"])
@pytest.mark.parametrize("maxiter", [2500, 1001])
def test_hals_more_sweeps_than_one_launch_tags(layout, maxiter, monkeypatch, built_lib):
    """hals_nnls_acc(maxiter > 1000) -- the reference accepts any maxiter (nnls.py:156); one launch tags at most 1000 sweeps,
    longer solves are chained launches (nnf_hals_solve_continue_f32) that carry eps0 and the count in the status block.
    delta = 0 runs to the budget across two / three launches: sweep count equal to the oracle's, factors within tolerance."""
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    if layout != "generic":
        monkeypatch.setenv("NNF_HALS_FORCE", layout)
    rng = np.random.RandomState(maxiter)
    r, n = 6, 150
    U = rng.rand(40, r)
    M = U @ rng.rand(r, n) + 1e-2 * rng.rand(40, n)
    UtU, UtM, V0 = U.T @ U, U.T @ M, rng.rand(r, n)
    kw = dict(maxiter=maxiter, delta=0.0, alpha=math.inf, normalize=(layout == "generic"))
    want, eps_o, cnt_o, _ = orc.hals_nnls_acc(UtM, UtU, V0, **kw)
    V, eps, cnt, _ = hals_nnls_acc(UtM, UtU, V0, **kw)
    assert cnt == cnt_o == maxiter + 1
    assert rel(V, want) < 1e-3, rel(V, want)


@pytest.mark.parametrize("layout", LAYOUTS + ["generic"])
@pytest.mark.parametrize("slice_", [1, 3, 7])
def test_hals_chained_launches_equal_one_launch(layout, slice_, monkeypatch, built_lib):
    """The chaining protocol at small counts: with the per-launch slice forced down to 1 / 3 / 7 sweeps the same solves -- stop
    by the rule in the first launch, in a later one, exactly at a slice end, at the budget -- must give bit for bit the factor,
    eps and cnt of the single launch (launches queued behind the one that stopped leave V and the status block alone)."""
    from nn_fac_amd import engine as eng_mod
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    if layout != "generic":
        monkeypatch.setenv("NNF_HALS_FORCE", layout)
    rng = np.random.RandomState(11)
    r, n = 10, 300
    U = rng.rand(80, r)
    M = U @ rng.rand(r, n) + 1e-2 * rng.rand(80, n)
    UtU, UtM, V0 = U.T @ U, U.T @ M, rng.rand(r, n)
    for maxiter, delta in ((100, 0.01), (100, 1e-4), (21, 0.0), (14, 1e-3), (2, 0.5)):
        kw = dict(maxiter=maxiter, delta=delta, alpha=math.inf, normalize=(layout == "generic"))
        V1, eps1, cnt1, _ = hals_nnls_acc(UtM, UtU, V0, **kw)
        monkeypatch.setattr(eng_mod.Engine, "HALS_MAX_SWEEPS_PER_LAUNCH", slice_)
        V2, eps2, cnt2, _ = hals_nnls_acc(UtM, UtU, V0, **kw)
        monkeypatch.setattr(eng_mod.Engine, "HALS_MAX_SWEEPS_PER_LAUNCH", 1000)
        assert cnt1 == cnt2 and eps1 == eps2 and np.array_equal(V1, V2), (maxiter, delta, cnt1, cnt2)


def test_device_side_stop_decision_of_the_sharded_solve(built_lib):
    """dist.sharded_hals_solve_async on one rank (no group): a blind chunk + nnf_hals_stop_restore_f32 must leave the factor
    and status words of the persistent solve when the chunk's snapshot window contains the stopping sweep, and flag a missed
    guess otherwise (3: the rule fired before the window; 4: not within the chunk) -- never a wrong factor silently."""
    from nn_fac_amd import dist as nd
    from nn_fac_amd.engine import get_engine, ST_EPS, ST_CNT, ST_EPS0, ST_ERR
    eng = get_engine("cuda:0")
    rng = np.random.RandomState(5)
    r, n = 20, 70000                      # lane kernel
    U = rng.rand(300, r)
    M = U @ rng.rand(r, n) + 1e-2 * rng.rand(300, n)
    UtU, UtM, V0 = dev(U.T @ U), dev(U.T @ M), dev(rng.rand(r, n))
    ref = V0.clone()
    st = eng.hals_solve(UtM, UtU, ref, 100, delta=0.01).cpu()
    cnt = int(st[ST_CNT])
    assert 4 < cnt - 1 < 90
    for value, want_err in ((cnt - 1 + 4, 0), (cnt - 1, 0), (cnt - 1 + 7, 0), (cnt - 3, 4), (cnt - 1 + 30, 3)):
        F = V0.clone()
        status = torch.zeros(8, dtype=torch.float64, device="cuda")
        guess = nd.SweepGuess(first=value, max_chunk=104, window=8)
        nd.sharded_hals_solve_async(eng, UtM, UtU, F, None, guess, status, budget=100, delta=0.01)
        s = status.cpu()
        assert int(s[ST_ERR]) == want_err, (value, s)
        if want_err == 0:
            assert int(s[ST_CNT]) == cnt and float(s[ST_EPS]) == float(st[ST_EPS]) and float(s[ST_EPS0]) == float(st[ST_EPS0])
            assert torch.equal(F, ref)
    # at the sweep budget the stop is the last sweep by construction
    ref2 = V0.clone()
    st2 = eng.hals_solve(UtM, UtU, ref2, 6, delta=0.0).cpu()
    F = V0.clone()
    status = torch.zeros(8, dtype=torch.float64, device="cuda")
    nd.sharded_hals_solve_async(eng, UtM, UtU, F, None, nd.SweepGuess(first=50, max_chunk=104, window=8), status, budget=6, delta=0.0)
    assert int(status[ST_ERR]) == 0 and int(status[ST_CNT]) == int(st2[ST_CNT]) == 7 and torch.equal(F, ref2)


def test_c_abi_rccl_communicator_single_rank(built_lib):
    """nnf_comm_* / nnf_allreduce_f32|f64: the RCCL exchange of the row-sharded path behind the C ABI.  One GPU here, so a
    communicator of ONE rank: id, init, in-place sum all-reduce (the identity) of the V-side buffer UtM | UtU and of a
    float64 vector on the caller's stream, destroy."""
    from nn_fac_amd.engine import get_engine, Comm
    eng = get_engine("cuda:0")
    uid = Comm.unique_id()
    assert len(uid) == 128
    comm = Comm(eng, 1, 0, uid)
    assert comm.size() == 1 and comm.rank() == 0
    a = torch.rand(50 * 2000 + 50 * 50, device="cuda")
    b = torch.rand(104, dtype=torch.float64, device="cuda")
    a0, b0 = a.clone(), b.clone()
    comm.allreduce_(a)
    comm.allreduce_(b)
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)
    comm.close()


def test_hals_normalize_beyond_the_resident_column_limit(built_lib):
    """normalize=True needs the norm of a whole row after every row update (nnls.py:179-185); the generic kernel does that with
    every column resident (<= 131072).  Beyond, the rows are walked from the host (Engine._hals_solve_rowwalk: row update, norm,
    scaling -- the one-device form of the row-sharded protocol): same result, same sweep count as the oracle.  nonzero=True
    (nnls.py:172-177: a row left all zero is refilled with 1e-16 max(V); a zero Gram diagonal raises) walks the rows the same
    way since round 4, alone and together with normalize, and so does the wall-clock rule's one-sweep probe
    (deterministic=False)."""
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    from nn_fac_amd.utils import errors as err
    rng = np.random.RandomState(17)
    r, n = 5, 140001
    A = rng.rand(40, r)
    UtU = A.T @ A
    UtM = A.T @ (A @ rng.rand(r, n) + 0.1 * rng.rand(40, n))
    V0 = rng.rand(r, n)
    for sp in (None, 0.05):
        Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM, UtU, V0, maxiter=6, alpha=math.inf, delta=0.01, sparsity_coefficient=sp,
                                              normalize=True)
        V, eps, cnt, _ = hals_nnls_acc(UtM.astype(np.float32), UtU.astype(np.float32), V0.astype(np.float32), maxiter=6,
                                       alpha=math.inf, delta=0.01, sparsity_coefficient=sp, normalize=True)
        assert cnt == cnto
        assert rel(V, Vo) < 2e-4 and abs(eps - epso) <= 5e-3 * abs(epso)
        np.testing.assert_allclose(np.linalg.norm(np.asarray(V, dtype=np.float64), axis=1), 1.0, rtol=1e-5)
    # nonzero=True: row 2's right-hand side is pushed far below zero, so its update leaves the row all zero -> refilled
    UtM2 = UtM.copy()
    UtM2[2] = -5.0 - rng.rand(n)
    for nrm in (False, True):
        Vo, epso, cnto, _ = orc.hals_nnls_acc(UtM2, UtU, V0, maxiter=4, alpha=math.inf, delta=0.01, normalize=nrm, nonzero=True)
        V, eps, cnt, _ = hals_nnls_acc(UtM2.astype(np.float32), UtU.astype(np.float32), V0.astype(np.float32), maxiter=4,
                                       alpha=math.inf, delta=0.01, normalize=nrm, nonzero=True)
        assert cnt == cnto and rel(V, Vo) < 2e-4 and abs(eps - epso) <= 5e-3 * abs(epso)
        assert np.all(np.asarray(V)[2] > 0)
    G0 = UtU.copy()
    G0[3, 3] = 0.0
    with pytest.raises(err.ZeroColumnWhenUnautorized):
        hals_nnls_acc(UtM.astype(np.float32), G0.astype(np.float32), V0.astype(np.float32), maxiter=3, alpha=math.inf, nonzero=True)
    # the reference's default call (deterministic=False): its one-sweep timing probe runs through the row-walk too
    V, eps, cnt, rho = hals_nnls_acc(UtM.astype(np.float32), UtU.astype(np.float32), V0.astype(np.float32), maxiter=3, atime=1e-3,
                                     alpha=0.5, normalize=True)
    assert 2 <= cnt <= 4
