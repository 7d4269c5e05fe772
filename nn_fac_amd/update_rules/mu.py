"""beta-divergence multiplicative updates on the MI355X engine -- drop-in for nn_fac/update_rules/mu.py:18-97.

``mu_betadivmin(U, V, M, beta)`` and ``switch_alternate_mu(data, U, V, beta, matrix)`` keep the reference signatures;
the fused kernels (``nnf_mu_left_f32`` / ``nnf_mu_right_f32``) stream M once and never materialise U@V.
"""
import torch

from ..utils import errors as err
from ..utils.beta_divergence import gamma_beta  # noqa: F401  (re-exported like the reference module does)
from .. import engine as _engine
from .._convert import device_of, to_dev, to_dev_t, like_input

epsilon = 1e-12  # mu.py:18


def mu_betadivmin(U, V, M, beta):
    """U <- max(U * ((K^(beta-2) .* M) V^T / (K^(beta-1) V^T))^gamma(beta), 1e-12), K = U V   (mu.py:79-97)."""
    if beta < 0:
        raise err.InvalidArgumentValue("Invalid value for beta: negative one.") from None
    dev = device_of(U, V, M)
    eng = _engine.get_engine(dev)
    X = to_dev(M, dev)
    Ut = to_dev_t(U, dev)
    Vd = to_dev(V, dev)
    out = eng.mu_left(X, Ut, Vd, beta)
    return like_input(out.t(), U)


def switch_alternate_mu(data, U, V, beta, matrix):
    """mu.py:20-29: 'U'/'W' updates the left factor, 'V'/'H' the right one (same rule on the transposed problem)."""
    if matrix in ["U", "W"]:
        return mu_betadivmin(U, V, data, beta)
    elif matrix in ["V", "H"]:
        if beta < 0:
            raise err.InvalidArgumentValue("Invalid value for beta: negative one.") from None
        dev = device_of(U, V, data)
        eng = _engine.get_engine(dev)
        out = eng.mu_right(to_dev(data, dev), to_dev_t(U, dev), to_dev(V, dev), beta)
        return like_input(out, V)
    else:
        raise err.InvalidArgumentValue(f"Invalid value for matrix: got {matrix}, but it must be 'U' or 'W' for the first matrix, and 'V' or 'H' for the second one.") from None


def mu_tensorial(G, factors, tensor, beta):
    """Core update of NTD-MU (reference mu.py:99-159): max(G * (L2 x F^T / L1 x F^T)^gamma, epsilon).
    Same arguments as the reference (core, list of factors I_n x r_n, tensor, beta); 3-way tensors."""
    from .. import engine as _engine
    from .._convert import device_of, to_dev, to_dev_t, like_input
    from ..ntd import _NtdState, _mu_tensorial_dev
    if beta < 0:
        raise err.InvalidArgumentValue("Invalid value for beta: negative one.") from None
    _engine.check_rank(max(G.shape), "mu_tensorial")
    dev = device_of(tensor, G, *factors)
    eng = _engine.get_engine(dev)
    st = _NtdState(eng, to_dev(tensor, dev))
    core = _mu_tensorial_dev(st, to_dev(G, dev).contiguous(), [to_dev_t(f, dev) for f in factors], beta)
    return like_input(core, G)
