"""Multilayer beta-NMF on the MI355X engine -- same call signature and results as nn_fac/multilayer_nmf.py:7-51 (a caller
that inherits the hot path through the unchanged ``nmf()`` signature).

A stack of L factorisations: the first one factorises the data, every further one the (rescaled) left factor of the one
before it.  A layer is one MU run of ``nmf`` at the given beta (NNDSVD start values by default, computed on the device),
after which the rows of H are scaled to unit sum and the scales moved into W (``normalize_WH(., ., "H")``).
"""
import warnings

import numpy as np

from .nmf import nmf
from .utils.normalize_wh import normalize_WH


def _clip_ranks(all_ranks, limit):
    """Ranks larger than the smaller data dimension are lowered to it, in place like the reference (multilayer_nmf.py:14-22)."""
    too_big = [i for i, rk in enumerate(all_ranks) if rk > limit]
    if not too_big:
        return
    for i in too_big:
        all_ranks[i] = limit
    print(f"The ranks are too high for the input matrix. The {len(too_big)} larger ranks were set to {limit} instead.")
    warnings.warn("Ranks have been changed.")


def one_layer_update(data, rank, beta, delta, init_each_nmf, n_iter_max_each_nmf, verbose, deterministic=False, seed=0):
    """One layer (multilayer_nmf.py:46-51): MU NMF of `data`, then H rows scaled to sum one.  `delta` is accepted and unused,
    as in the reference.  Returns (W, H, cost per iteration, total time)."""
    W, H, costs, times = nmf(data, rank, init=init_each_nmf, n_iter_max=n_iter_max_each_nmf, tol=1e-8, update_rule="mu",
                             beta=beta, normalize=[False, True], verbose=verbose, return_costs=True,
                             deterministic=deterministic, seed=seed)
    W, H = normalize_WH(W, H, matrix="H")
    return W, H, np.asarray(costs), np.sum(times)


def multilayer_beta_NMF(data, all_ranks, beta=1, delta=1e-6, n_iter_max_each_nmf=100, init_each_nmf="nndsvd",
                        return_errors=False, verbose=False, deterministic=False, seed=0):
    """Returns the lists W, H (one entry per layer) [, the L x n_iter_max_each_nmf table of costs, the time per layer].
    Like the reference, a layer that stops before n_iter_max_each_nmf iterations does not fit its row of the table
    (NumPy's broadcasting error, multilayer_nmf.py:32,37)."""
    if deterministic:
        np.random.seed(seed)
    n_layers = len(all_ranks)
    assert n_layers > 1, "The number of layers must be at least 2. Otherwise, ou should just use NMF"
    _clip_ranks(all_ranks, min(data.shape))
    if any(a < b for a, b in zip(all_ranks, all_ranks[1:])):
        raise ValueError("The ranks of deep NMF should be decreasing.")

    W, H, toc = [], [], []
    errors = np.full((n_layers, n_iter_max_each_nmf), np.nan)
    current = data
    for layer, rank in enumerate(all_ranks):
        W_l, H_l, errors[layer], t_l = one_layer_update(current, rank, beta, delta, init_each_nmf, n_iter_max_each_nmf,
                                                        verbose, deterministic=deterministic, seed=seed)
        W.append(W_l)
        H.append(H_l)
        toc.append(t_l)
        current = W_l
        if verbose and layer > 0:
            print(f'Layer {layer} done.')
    if return_errors:
        return W, H, errors, toc
    return W, H
