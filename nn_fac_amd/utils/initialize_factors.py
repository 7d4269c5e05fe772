"""Initialisers -- mirror of nn_fac/utils/initialize_factors.py (random and NNDSVD branches).

The random branch reproduces the reference's legacy global NumPy stream bit for bit
(initialize_factors.py:40-46,90-96): ``np.random.seed(seed); random.seed(seed); rand(m, r); rand(r, n)``.
NNDSVD (initialize_factors.py:160-206) runs on the device: data of config-B/E size should not have
to exist on the host.  The Tucker initialiser (initialize_factors.py:68-81) restates tensorly 0.6.0's HOSVD + HOOI
`tucker` on the device in float64.
"""
import random

import numpy as np
import torch

from . import errors as err


def _thin_svd_top(X, rank):
    """First `rank` singular triplets of X (float64, device).  A strongly rectangular matrix goes through the
    eigen-decomposition of its small Gram matrix (n x n or m x m): one fp64 GEMM over X instead of a full
    bidiagonalisation of a 100000 x 2000 array; the reference's np.linalg.svd(full_matrices=True) could not even
    allocate its 100000 x 100000 U there.  Only triplets whose singular value is far above sqrt(eps)*sigma_0 are
    trustworthy on that route (the Gram squares the condition number): others fall back to the direct SVD."""
    m, n = X.shape
    small, big = (n, m) if m >= n else (m, n)
    if big >= 2 * small and small > rank:
        G = X.T @ X if m >= n else X @ X.T
        lam, Q = torch.linalg.eigh(G)
        lam, Q = lam.flip(0)[:rank], Q.flip(1)[:, :rank]
        if float(lam[-1]) > 1e-10 * float(lam[0]):
            S = lam.sqrt()
            if m >= n:
                Vr = Q
                Ur = (X @ Vr) / S
            else:
                Ur = Q
                Vr = (X.T @ Ur) / S
            return Ur, S, Vr
    U, S, Vh = torch.linalg.svd(X, full_matrices=False)
    return U[:, :rank], S[:rank], Vh[:rank].T


def nndsvd(V, rank):
    """NNDSVD start values (initialize_factors.py:160-206) computed on the device in float64.  NumPy in -> NumPy out
    (float64, as the reference); device tensor in -> device tensors out.  The outcome does not depend on the sign
    convention of the singular vectors: flipping a pair (u, v) swaps the two candidates of :195-200."""
    from .._convert import device_of
    dev = device_of(V)
    X = V.to(device=dev, dtype=torch.float64) if isinstance(V, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(V, dtype=np.float64)).to(dev)
    U, S, E = _thin_svd_top(X, rank)
    up, un = U.clamp(min=0), (-U).clamp(min=0)            # _pos / _neg, column by column (:164-168)
    vp, vn = E.clamp(min=0), (-E).clamp(min=0)
    n_up, n_un = up.norm(dim=0), un.norm(dim=0)
    n_vp, n_vn = vp.norm(dim=0), vn.norm(dim=0)
    termp, termn = n_up * n_vp, n_un * n_vn
    pos = termp >= termn                                   # :195
    term = torch.where(pos, termp, termn)
    cu = torch.where(pos, n_up, n_un)
    cv = torch.where(pos, n_vp, n_vn)
    W = torch.where(pos, up, un) * (torch.sqrt(S * term) / cu)
    H = (torch.where(pos, vp, vn) * (torch.sqrt(S * term) / cv)).T
    W[:, 0] = torch.sqrt(S[0]) * U[:, 0].abs()             # first triplet: absolute values (:179-180)
    H[0, :] = torch.sqrt(S[0]) * E[:, 0].abs()
    W, H = W.clamp(min=1e-12), H.clamp(min=1e-12).contiguous()   # :204-205
    if isinstance(V, torch.Tensor):
        return W.to(V.dtype), H.to(V.dtype)
    return W.cpu().numpy(), H.cpu().numpy()


def nmf_initialization(data, rank, init_type, deterministic=False, seed=0):
    kind = init_type.lower()
    if kind == "random":
        if deterministic:
            np.random.seed(seed)
            random.seed(seed)
        m, n = data.shape
        U_0 = np.random.rand(m, rank)
        V_0 = np.random.rand(rank, n)
        return U_0, V_0
    if kind == "nndsvd":
        return nndsvd(data, rank)
    raise err.InvalidInitializationType("Initialization type not understood.")


def ntf_initialization(tensor, rank, init_type, deterministic=False, seed=0):
    if deterministic:
        np.random.seed(seed)
        random.seed(seed)
    kind = init_type.lower()
    if kind == "random":
        return [np.random.rand(tensor.shape[mode], rank) for mode in range(len(tensor.shape))]
    if kind == "nndsvd":
        # initialize_factors.py:98-105: NNDSVD of every unfolding; modes shorter than the rank fall back to rand
        factors = []
        for mode in range(len(tensor.shape)):
            if tensor.shape[mode] < rank:
                factors.append(np.random.rand(tensor.shape[mode], rank))
            else:
                t = tensor if isinstance(tensor, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tensor))
                unf = torch.movedim(t, mode, 0).reshape(t.shape[mode], -1)      # tl.unfold
                W, _ = nndsvd(unf if isinstance(tensor, torch.Tensor) else unf.numpy(), rank)
                factors.append(W)
        return factors
    raise err.InvalidInitializationType("Initialization type not understood.")


def _mode_dot_t(T, F, mode):
    """T x_mode F^T for F of shape (n_mode, r): contracts mode `mode` of T with the rows of F."""
    return torch.movedim(torch.tensordot(T, F, dims=([mode], [0])), -1, mode)


def tucker_hooi(tensor, ranks, n_iter_max=100, tol=10e-5):
    """tensorly.decomposition.tucker(tensor, ranks) of tensorly 0.6.0 (third party, setup.py:30) restated on the device in
    float64: HOSVD start, then HOOI sweeps until the relative reconstruction error moves by less than tol (tested from the
    third sweep on).  The leading left singular vectors of an unfolding come from `_thin_svd_top` (Gram route for the
    strongly rectangular unfoldings of a big tensor).  Signs of singular vectors are arbitrary; the caller takes absolute
    values.  NumPy in -> NumPy out; device tensor in -> device tensors (of the input dtype) out.
    Checked against the oracle's tucker_hooi (itself pinned by NTD_tests.py:157-175,197-215) in tests/test_gpu_ntd.py."""
    from .._convert import device_of
    dev = device_of(tensor)
    T = tensor.to(device=dev, dtype=torch.float64) if isinstance(tensor, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(tensor, dtype=np.float64)).to(dev)
    N = T.dim()

    def leading(X, mode):
        unf = torch.movedim(X, mode, 0).reshape(X.shape[mode], -1)
        return _thin_svd_top(unf, ranks[mode])[0].contiguous()
    factors = [leading(T, m) for m in range(N)]
    norm_t2 = float((T * T).sum())
    errs = []
    core = None
    for it in range(n_iter_max):
        for m in range(N):
            approx = T
            for i in range(N):
                if i != m:
                    approx = _mode_dot_t(approx, factors[i], i)
            factors[m] = leading(approx, m)
        core = T
        for i in range(N):
            core = _mode_dot_t(core, factors[i], i)
        errs.append(np.sqrt(abs(norm_t2 - float((core * core).sum()))) / np.sqrt(norm_t2))
        if it > 1 and abs(errs[-2] - errs[-1]) < tol:
            break
    if isinstance(tensor, torch.Tensor):
        return core.to(tensor.dtype), [f.to(tensor.dtype) for f in factors]
    return core.cpu().numpy(), [f.cpu().numpy() for f in factors]


def ntd_initialization(tensor, ranks, init_type, deterministic=False, seed=0):
    """initialize_factors.py:50-83."""
    kind = init_type.lower()
    if kind == "random":
        factors = []
        if deterministic:
            np.random.seed(seed)
            random.seed(seed)
        for mode in range(len(tensor.shape)):
            one_factor = np.random.rand(tensor.shape[mode], ranks[mode])
            one_factor[one_factor < 1e-12] = 1e-12   # To avoid zeros
            factors.append(one_factor)
        the_core = np.random.rand(int(np.prod(ranks))).reshape(tuple(ranks))
        the_core[the_core < 1e-12] = 1e-12
        return the_core, factors
    if kind == "tucker":        # :68-75 (tensorly's `random_state` only seeds ARPACK's start vector: the converged leading subspaces do not depend on it)
        init_core, init_factors = tucker_hooi(tensor, list(ranks))
        if isinstance(tensor, torch.Tensor):
            return init_core.abs() + 1e-12, [f.abs() + 1e-12 for f in init_factors]
        return np.abs(init_core) + 1e-12, [np.abs(f) + 1e-12 for f in init_factors]
    if kind == "chromas":       # :77-80 -- Tucker where W is fixed to I12
        core, factors = ntd_initialization(tensor, ranks, "tucker", deterministic=deterministic, seed=seed)
        factors[0] = np.identity(12)
        return core, factors
    raise err.InvalidInitializationType("Initialization type not understood.")
