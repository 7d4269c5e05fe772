"""beta-divergence cost and MU exponent -- mirror of nn_fac/utils/beta_divergence.py:17-80.

``beta_divergence(a, b, beta)`` keeps the reference signature for the generic two-array form (device element-wise
kernels are not needed there: it is only called by the drivers through the fused ``Engine.betadiv`` which never
materialises b = U@V).  ``gamma_beta`` is host arithmetic.
"""
import torch

from . import errors as err
from .._convert import device_of, to_dev


def gamma_beta(beta):
    """Fevotte-Idier exponent (beta_divergence.py:75-80)."""
    if beta < 1:
        return 1 / (2 - beta)
    if beta > 2:
        return 1 / (beta - 1)
    return 1


def beta_divergence(a, b, beta):
    """Sum of the element-wise beta-divergence d(a|b) (beta_divergence.py:42-52) for two explicit arrays.

    Runs as torch device ops in fp64 (this generic form is off the hot path; the drivers use the fused kernel).
    Inputs must be strictly positive for beta in {0, 1}, as in the reference (its masked entries are undefined).
    """
    if beta < 0:
        raise err.InvalidArgumentValue("Invalid value for beta: negative one.") from None
    dev = device_of(a, b)
    A, B = to_dev(a, dev).double(), to_dev(b, dev).double()
    if beta == 1:
        return float(torch.sum(A * torch.log(A / B) - A + B))
    if beta == 0:
        q = A / B
        return float(torch.sum(q - torch.log(q) - 1))
    return float(torch.sum((A ** beta + (beta - 1) * B ** beta - beta * A * B ** (beta - 1)) / (beta * (beta - 1))))


def kl_divergence(a, b):
    return beta_divergence(a, b, beta=1)
