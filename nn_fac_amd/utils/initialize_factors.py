"""Initialisers -- mirror of nn_fac/utils/initialize_factors.py (random and NNDSVD branches).

The random branch reproduces the reference's legacy global NumPy stream bit for bit
(initialize_factors.py:40-46,90-96): ``np.random.seed(seed); random.seed(seed); rand(m, r); rand(r, n)``.
NNDSVD (initialize_factors.py:160-206) runs on the device: data of config-B/E size should not have
to exist on the host.  Tucker (HOSVD) initialisers stay out of scope and raise.
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


def ntd_initialization(tensor, ranks, init_type, deterministic=False, seed=0):
    """initialize_factors.py:50-83; 'tucker' / 'chromas' need a Tucker decomposition (HOSVD), outside the hot path."""
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
    if kind in ("tucker", "chromas"):
        raise NotImplementedError("tucker (HOSVD) initialisation is outside the accelerated hot path; pass init='custom'")
    raise err.InvalidInitializationType("Initialization type not understood.")
