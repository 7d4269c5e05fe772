"""Host-side initialisers -- mirror of nn_fac/utils/initialize_factors.py (random branches only).

The random branch reproduces the reference's legacy global NumPy stream bit for bit
(initialize_factors.py:40-46,90-96): ``np.random.seed(seed); random.seed(seed); rand(m, r); rand(r, n)``.
NNDSVD / Tucker initialisers are out of the hot-path scope and raise.
"""
import random

import numpy as np

from . import errors as err


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
        raise NotImplementedError("nndsvd initialisation is outside the accelerated hot path; pass init='custom'")
    raise err.InvalidInitializationType("Initialization type not understood.")


def ntf_initialization(tensor, rank, init_type, deterministic=False, seed=0):
    if deterministic:
        np.random.seed(seed)
        random.seed(seed)
    kind = init_type.lower()
    if kind == "random":
        return [np.random.rand(tensor.shape[mode], rank) for mode in range(len(tensor.shape))]
    if kind == "nndsvd":
        raise NotImplementedError("nndsvd initialisation is outside the accelerated hot path; pass init='custom'")
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
