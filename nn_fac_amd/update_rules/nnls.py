"""hals_nnls_acc on the MI355X engine -- drop-in for nn_fac/update_rules/nnls.py:24-198.

Same signature, argument meaning, return tuple ``(V, eps, cnt, rho)`` and exceptions as the reference.  The sweep
loop, the sum of squared steps and the ``eps >= delta*eps0`` stopping rule run inside ONE persistent HIP launch
(``nnf_hals_solve_f32``); only the wall-clock rule ``cnt <= 1 + alpha*rho`` (nnls.py:156,190-194) stays on the host,
because it is defined by host time: with a finite ``alpha`` and a truthy ``atime`` one probe sweep is timed on a
scratch copy to obtain ``btime``, exactly the quantity the reference measures.

Inputs may be NumPy arrays (uploaded, computed in fp32, answered as NumPy in the input dtype) or torch tensors
(device tensors are used in place; the result is a new device tensor).  ``in_V`` is never modified (nnls.py:147).
"""
import math
import time

import numpy as np
import torch

from ..utils import errors as err
from .. import engine as _engine
from .._convert import device_of, to_dev, like_input


def _ndim(a):
    return len(np.shape(a)) if not isinstance(a, torch.Tensor) else a.dim()


def _size(a):
    return a.numel() if isinstance(a, torch.Tensor) else np.size(a)


def hals_solve_device(eng, UtM, UtU, V, max_sweeps, delta, sparsity_coefficient, normalize, nonzero, status=None):
    """Device-level call used by the drivers: in place on V (r x n), no host synchronisation."""
    return eng.hals_solve(UtM, UtU, V, max_sweeps, delta=delta, sparsity=sparsity_coefficient, normalize=normalize,
                          nonzero=nonzero, status=status)


def sweep_budget(maxiter, alpha, rho):
    """Largest cnt allowed by ``cnt <= 1 + alpha*rho and cnt <= maxiter`` (nnls.py:156)."""
    lim = 1 + alpha * rho
    if math.isinf(lim) or lim >= maxiter:
        return int(maxiter)
    return int(max(0, math.floor(lim)))


def hals_nnls_acc(UtM, UtU, in_V, maxiter=500, atime=None, alpha=0.5, delta=0.01,
                  sparsity_coefficient=None, normalize=False, nonzero=False):
    """Accelerated HALS NNLS (Gillis & Glineur 2012).  See the reference docstring, nnls.py:28-129."""
    if _ndim(UtM) != 2:
        raise err.ArgumentException(f"Argument UtM is an array of {np.shape(UtM)} dimensions when it should be a matrix.")
    if _ndim(UtU) != 2:
        raise err.ArgumentException(f"Argument UtU is an array of {np.shape(UtU)} dimensions when it should be a matrix.")
    if _ndim(in_V) != 2:
        raise err.ArgumentException(f"Argument in_V is an array of {np.shape(in_V)} dimensions when it should be a matrix.")

    dev = device_of(UtM, UtU, in_V)
    eng = _engine.get_engine(dev)
    M = to_dev(UtM, dev)
    G = to_dev(UtU, dev)
    r, n = M.shape
    if not _size(in_V):
        # nnls.py:138-145: unconstrained least squares, clipped and rescaled (host-side plumbing, runs once)
        V = torch.linalg.solve(G.double(), M.double())
        V[V < 0] = 0
        V = (torch.sum(M.double() * V) / torch.sum(G.double() * (V @ V.T))) * V
        V = V.float().contiguous()
    else:
        V = to_dev(in_V, dev)
        if isinstance(in_V, torch.Tensor) and V.data_ptr() == in_V.data_ptr():
            V = V.clone()   # never touch the caller's array (nnls.py:147)
    R = V.shape[0]
    if V.shape[1] != n or G.shape[0] < R or G.shape[1] < R or R < r:
        raise err.ArgumentException(f"Inconsistent shapes: UtM {tuple(M.shape)}, UtU {tuple(G.shape)}, V {tuple(V.shape)}.")
    if R != r:
        # the reference takes r from UtM and sweeps rows < r only, while UtU[k,:]@V runs over every row of V
        # (nnls.py:137,158,167; tests/nnls_tests.py:44-45): sweep all R rows with the extra ones frozen (zero diagonal)
        G = G[:R, :R].clone()
        idx = torch.arange(r, R, device=dev)
        G[idx, idx] = 0.0
        M = torch.cat([M, torch.zeros((R - r, n), dtype=torch.float32, device=dev)], dim=0)
        if nonzero:
            raise err.ArgumentException("nonzero=True needs UtM, UtU and V of consistent rank.")

    rho = 100000
    budget = sweep_budget(maxiter, alpha, rho)
    if atime and not math.isinf(alpha):
        # probe: wall time of the first sweep (nnls.py:155,190), measured on a scratch copy
        probe = V.clone()
        torch.cuda.synchronize(dev)
        t0 = time.time()
        eng.hals_sweeps(M, G, probe, 1, sparsity=sparsity_coefficient, normalize=normalize, nonzero=nonzero)
        torch.cuda.synchronize(dev)
        btime = max(time.time() - t0, 10e-7)
        rho = atime / btime
        budget = max(1, sweep_budget(maxiter, alpha, rho)) if maxiter >= 1 else 0

    st = eng.hals_solve(M, G, V, budget, delta=delta, sparsity=sparsity_coefficient, normalize=normalize,
                        nonzero=nonzero)
    st = st.cpu()
    code = int(st[_engine.ST_ERR])
    if code == 2:
        k = int(torch.nonzero(torch.diagonal(G)[:r] == 0)[0])
        raise err.ZeroColumnWhenUnautorized("Column " + str(k) + " of U is zero with nonzero condition")
    if code != 0:
        raise err.EngineError("hals grid barrier timed out; result invalid")
    eps, cnt = float(st[_engine.ST_EPS]), int(st[_engine.ST_CNT])
    if maxiter < 1:
        eps, cnt = 1, 1
    return like_input(V, in_V), eps, cnt, rho


def hals_coupling_nnls_acc(UtM, UtU, in_V, Vtarget, mu, maxiter=500, atime=None, alpha=0.5, delta=0.01,
                           normalize=False, nonzero=False):
    """HALS NNLS coupled to a target matrix, min_{V>=0} ||M-UV||_F^2 + mu ||V-Vtarget||_F^2 -- drop-in for
    nn_fac/update_rules/nnls.py:204-352 (PARAFAC2's caller of the sweep, parafac2.py:548,581).

    The coupled row update (nnls.py:317)
        (UtM[k] - UtU[k]@V + mu (Vtarget[k] - V[k])) / (UtU[k,k] + mu)
    is the plain update of hals_nnls_acc on the shifted operands  UtM + mu Vtarget  and  UtU + mu I , so the same
    persistent device solve runs it (two tiny element-wise preparations, r x n and r x r).  What does NOT carry over is
    the zero-diagonal test, which the reference makes on UtU[k,k] alone (nnls.py:316): such rows keep a zero diagonal
    in the shifted Gram, which is how the kernels recognise a frozen row.  Same return tuple ``(V, eps, cnt, rho)``;
    ``nonzero`` with a zero diagonal raises the reference's plain ValueError (nnls.py:331-332).  The reference does
    not validate its arguments here (no ArgumentException checks); shapes are checked only as far as the device needs."""
    dev = device_of(UtM, UtU, in_V)
    eng = _engine.get_engine(dev)
    M = to_dev(UtM, dev)
    G = to_dev(UtU, dev)
    T = to_dev(Vtarget, dev)
    r, n = M.shape
    if G.shape[0] < r or G.shape[1] < r or tuple(T.shape) != (r, n):
        raise err.ArgumentException(f"Inconsistent shapes: UtM {tuple(M.shape)}, UtU {tuple(G.shape)}, "
                                    f"Vtarget {tuple(T.shape)}.")
    if not _size(in_V):
        # nnls.py:297-303: unconstrained least squares on the UNshifted operands, clipped and rescaled (runs once)
        V = torch.linalg.solve(G.double(), M.double())
        V[V < 0] = 0
        V = (torch.sum(M.double() * V) / torch.sum(G.double() * (V @ V.T))) * V
        V = V.float().contiguous()
    else:
        V = to_dev(in_V, dev)
        if isinstance(in_V, torch.Tensor) and V.data_ptr() == in_V.data_ptr():
            V = V.clone()   # never touch the caller's array (nnls.py:305)
    if tuple(V.shape) != (r, n):
        raise err.ArgumentException(f"Inconsistent shapes: UtM {tuple(M.shape)}, V {tuple(V.shape)}.")
    mu = float(mu)
    Ms = torch.add(M, T, alpha=mu)                       # UtM + mu Vtarget
    Gs = G[:r, :r].clone()
    d = torch.diagonal(Gs)
    frozen = d == 0
    d.add_(mu)
    d[frozen] = 0.0                                      # rows the reference skips stay skipped (nnls.py:316)

    rho = 100000
    budget = sweep_budget(maxiter, alpha, rho)
    if atime and not math.isinf(alpha):
        probe = V.clone()
        torch.cuda.synchronize(dev)
        t0 = time.time()
        eng.hals_sweeps(Ms, Gs, probe, 1, normalize=normalize, nonzero=nonzero)
        torch.cuda.synchronize(dev)
        btime = max(time.time() - t0, 10e-7)
        rho = atime / btime
        budget = max(1, sweep_budget(maxiter, alpha, rho)) if maxiter >= 1 else 0

    st = eng.hals_solve(Ms, Gs, V, budget, delta=delta, normalize=normalize, nonzero=nonzero).cpu()
    code = int(st[_engine.ST_ERR])
    if code == 2:
        k = int(torch.nonzero(frozen)[0])
        raise ValueError("Column " + str(k) + " is zero with nonzero condition")
    if code != 0:
        raise err.EngineError("hals grid barrier timed out; result invalid")
    eps, cnt = float(st[_engine.ST_EPS]), int(st[_engine.ST_CNT])
    if maxiter < 1:
        eps, cnt = 1, 1
    return like_input(V, in_V), eps, cnt, rho
