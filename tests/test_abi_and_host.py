"""C-ABI surface and host-side logic that need no GPU."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nnfac_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nnf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nnfac_hip.h but not exported"


def test_binding_table_matches_header(built_lib):
    from nn_fac_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    lib = _lib.load()
    assert lib.nnf_version() >= 100
    assert lib.nnf_status_string(0) == b"ok"
    assert b"not supported" in lib.nnf_status_string(-3)


def test_built_library_carries_the_default_build_switches(built_lib):
    """The timing-only ablation / A-B macros of the kernel sources (XHT_ABL, SEG_ABL, MTTKRP_ABL, HALS_LATE_ISSUE, ...): a stray
    -D in a build would silently ship a kernel that skips work.  Every translation unit records the values it was compiled
    with (nnf_build_flags); the library under test must carry the product defaults in every unit."""
    from nn_fac_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(8192)
    n = lib.nnf_build_flags(buf, 8192)
    assert 0 < n < 8192
    got = dict(u.split(": ", 1) for u in buf.value.decode().split("; "))
    hals = "HALS_LATE_ISSUE=1 HALS_MID_AT(R)=((R) - 1) HALS_DBG=0"
    quad = "QUAD_MID_SEL=1 HALS_LATE_ISSUE=1"
    mu = "MU_WG_PER_CU=2 MU_STEP_FENCE()=__builtin_amdgcn_sched_barrier(0)"
    want = {"k_stream": "XHT_ABL=0 XTY_BIG_WG=2", "k_mttkrp": "SEG_ABL=0 MTTKRP_ABL=0", "k_hals_wave": "WAVE_DBG=0 WAVE_REFRESH_V=8",
            "k_hals_mfma": "MFMA_NREF_V=32 MFMA_EARLY=1 MFMA_DBG=0",
            **{f"k_hals_fast{i}": hals for i in range(4)}, **{f"k_hals_quad{i}": quad for i in range(4)},
            **{f"k_mu{i}": mu for i in range(3)}}
    for unit, flags in want.items():
        assert got.get(unit) == flags, (unit, got.get(unit))


def test_tucker_rank_above_the_kernels_limit_is_refused_at_the_boundary():
    """The reference takes any rank up to min(shape) (nn_fac/nmf.py:175-178).  The matrix path and NTF follow it (rank chunks,
    generic sweep kernel: tests/test_gpu_bigrank.py); the TUCKER kernels (core contractions, core update) stop at 128 and every
    drop-in entry on that path says so before anything is uploaded or launched (also without a GPU), not as a bare status code."""
    from nn_fac_amd.utils.errors import EngineError
    from nn_fac_amd.update_rules.mu import mu_tensorial
    from nn_fac_amd.ntd import ntd
    rng = np.random.RandomState(0)
    r = 129
    calls = [lambda: ntd(rng.rand(130, 131, 132), [r, 4, 4], n_iter_max=1),
             lambda: mu_tensorial(rng.rand(r, 4, 4), [rng.rand(130, r), rng.rand(131, 4), rng.rand(132, 4)], rng.rand(130, 131, 132), 1)]
    for f in calls:
        with pytest.raises(EngineError, match="rank 129 is above the 128"):
            f()


def test_no_gpu_means_loud_failure(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nn_fac_amd.utils.errors import EngineError
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    from nn_fac_amd.nmf import nmf
    r = np.random.RandomState(0)
    with pytest.raises(EngineError):
        hals_nnls_acc(r.rand(4, 6), r.rand(4, 4), r.rand(4, 6))
    with pytest.raises(EngineError):
        nmf(r.rand(20, 10), 3, n_iter_max=2, deterministic=True)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nn_fac_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert "nnfac_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_argument_exceptions_are_raised_before_touching_the_device():
    from nn_fac_amd.utils import errors as err
    from nn_fac_amd.update_rules.nnls import hals_nnls_acc
    from nn_fac_amd.update_rules.mu import mu_betadivmin, switch_alternate_mu
    from nn_fac_amd.nmf import nmf
    r = np.random.RandomState(0)
    # tests/nnls_tests.py:21-28,46-47 of the reference
    with pytest.raises(err.ArgumentException):
        hals_nnls_acc(r.rand(8, 8), r.rand(8, 8), np.array([]))
    with pytest.raises(err.ArgumentException):
        hals_nnls_acc(r.rand(8), r.rand(8, 8), r.rand(8, 8))
    with pytest.raises(err.ArgumentException):
        hals_nnls_acc(r.rand(8, 8), r.rand(8), r.rand(8, 8))
    with pytest.raises(err.ArgumentException):
        hals_nnls_acc(r.rand(8), r.rand(15, 15), r.rand(15, 1), nonzero=True)
    with pytest.raises(err.InvalidArgumentValue):
        mu_betadivmin(r.rand(5, 2), r.rand(2, 4), r.rand(5, 4), -0.5)
    with pytest.raises(err.InvalidArgumentValue):
        switch_alternate_mu(r.rand(5, 4), r.rand(5, 2), r.rand(2, 4), 1, "Z")
    # tests/NMF_tests.py:45-54
    with pytest.raises(err.InvalidInitializationType):
        nmf(r.rand(20, 10), 3, init="invalid_init", n_iter_max=2, deterministic=True)
    with pytest.raises(err.CustomNotValidFactors):
        nmf(r.rand(20, 10), 3, init="custom", U_0=None, V_0=r.rand(3, 10), n_iter_max=2)
    assert issubclass(err.ArgumentException, BaseException) and not issubclass(err.ArgumentException, Exception)
    assert issubclass(err.ZeroColumnWhenUnautorized, err.OptimException)


def test_host_helpers():
    from nn_fac_amd.update_rules.nnls import sweep_budget
    from nn_fac_amd.utils.beta_divergence import gamma_beta
    from nn_fac_amd.utils.initialize_factors import nmf_initialization
    assert sweep_budget(100, math.inf, 100000) == 100
    assert sweep_budget(100, 0.5, 6.0) == 4          # cnt <= 1 + 0.5*6
    assert sweep_budget(100, 0.5, 100000) == 100
    assert sweep_budget(500, 0.5, 0.0) == 1
    assert gamma_beta(0) == 0.5 and gamma_beta(1) == 1 and gamma_beta(2) == 1 and gamma_beta(3) == 0.5
    U, V = nmf_initialization(np.zeros((73, 25)), 9, "random", deterministic=True, seed=0)
    assert abs(U[0][0] - 0.5488135) < 1e-7 and abs(V[0][0] - 1.15834001e-01) < 1e-7   # NMF_tests.py:40-41


def test_sweep_kernels_never_touch_a_load_destination_before_its_wait(built_lib):
    """tools/check_sweep_spills.py on the ISA the build kept next to the sweep objects (k_hals_fast: hand-issued s_load into
    SGPRs; k_hals_quad: hand-issued ds_read_b128 into VGPRs): no instruction of a sweep block may name a register with such a
    load in flight before the wait that covers it -- the abort class of round 1 (a rank instantiation over the SGPR budget
    made the allocator move an in-flight buffer).  Runs in seconds after `make`; compiles to assembly otherwise."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_sweep_spills", os.path.join(ROOT, "tools", "check_sweep_spills.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, blocks = mod.check_all()
    assert blocks >= 60, blocks          # every rank instantiation of both kernels was seen
    assert not bad, bad[:3]


def test_streaming_kernels_keep_their_prefetch_ring_in_flight(built_lib):
    """tools/check_loop_drains.py on the ISA the build kept for k_stream / k_mttkrp: the chunk loops of the streaming MFMA
    kernels of the BASELINE configurations hold no `s_waitcnt vmcnt(0)` -- a full drain of the X prefetch ring per trip is
    what hipcc emits when a load sits under a branch and its value is used at once (round 2: 30 of the mode-2 MTTKRP's
    125 us).  A regression shows up here, on the CPU, instead of as a slower bench line."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_loop_drains", os.path.join(ROOT, "tools", "check_loop_drains.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    build = os.path.join(ROOT, "nn_fac_amd", "csrc", "build")
    want = {"k_stream.s": ["nnf_xty_kernel<3, 2, true>", "nnf_xht_kernel<3, 2, true, 4>", "nnf_xty_kernel<6, 4, true>",
                           "nnf_xht_kernel<6, 4, true, 4>", "nnf_xty_kernel<2, 0, true>"],
            "k_mttkrp.s": ["nnf_mttkrp_rows_kernel<2, true, true>", "nnf_mttkrp_seg_kernel<2, true>",
                           "nnf_mttkrp_rows_kernel<4, true, true>", "nnf_mttkrp_seg_kernel<4, true>"]}
    for fname, kernels in want.items():
        path = os.path.join(build, fname)
        if not os.path.exists(path):     # a library that came without its build directory: make recreates objects and ISA
            import __graft_entry__
            __graft_entry__.build()
        assert os.path.exists(path), f"{path}: the Makefile keeps the ISA of the streaming kernels next to the objects"
        found = mod.scan(path)
        for k in kernels:
            hits = [loops for name, loops in found.items() if k + "(" in name]
            assert hits, f"{k} not found in {fname}"
            for loops in hits:
                # (X H^T: the one-tile loops of the k-split tail -- 16 MFMAs per rank tile and trip, two trips per wave -- are
                #  not the streaming loop)
                mt = int(k.split("<")[1].split(",")[0])
                main = [l for l in loops if l["mfma"] >= max(90, 16 * mt + 1)]
                assert main, (k, loops)
                assert all(l["vmcnt0"] == 0 for l in main), (k, main)
