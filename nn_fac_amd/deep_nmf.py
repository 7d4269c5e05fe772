"""Deep KL-NMF on the MI355X engine -- same call signature and results as nn_fac/deep_nmf.py:13-113 (a caller that inherits
the MU hot path: SURVEY.md 8f row 4).

L stacked factorisations X ~ W0 H0, W0 ~ W1 H1, ..., started from ``multilayer_beta_NMF`` (beta = 1) and refined by
alternating, per layer, the KL multiplicative update of H (``switch_alternate_mu(..., "H")`` = nnf_mu_right_f32), the
rescaling of H's rows to unit sum (``normalize_WH``), and -- for every layer but the last -- the Lambert-W update of W that
balances the layer's own KL error against the next layer's (``deep_KL_mu``: nnf_mu_left_num_f32 + nnf_deep_kl_apply_f32;
the product W_{l+1} H_{l+1} is one nnf_small_gemm_f32); the last layer's W takes the plain KL update (nnf_mu_left_f32).
Layer errors are nnf_betadiv_f32 calls.  Factors live on the device for the whole run (W transposed, r x m), fp32.
"""
import time
import warnings

import numpy as np
import torch

from . import engine as _engine
from . import multilayer_nmf as multi_nmf
from ._convert import device_of, to_dev, to_dev_t, like_input
from .update_rules.deep_mu import _deep_kl_mu_dev


def deep_KL_NMF(data, all_ranks, n_iter_max_each_nmf=100, n_iter_max_deep_loop=100, init="multilayer_nmf",
                init_multi_layer="nndsvd", W_0=None, H_0=None, delta=1e-6, tol=1e-6, return_errors=False, verbose=False,
                deterministic=False, seed=0):
    L = len(all_ranks)
    assert L > 1, "The number of layers must be at least 2. Otherwise, you should just use NMF."
    multi_nmf._clip_ranks(all_ranks, min(data.shape))            # deep_nmf.py:17-26

    reconstruction_errors = np.full((L, n_iter_max_deep_loop + 1), np.nan)
    toc = []
    global_errors = []

    if sorted(all_ranks, reverse=True) != all_ranks:
        raise ValueError("The ranks of deep NMF should be decreasing.")

    dev = device_of(data)
    eng = _engine.get_engine(dev)
    X = to_dev(data, dev)

    if init == "multilayer_nmf":
        W, H, e, _ = multi_nmf.multilayer_beta_NMF(X, all_ranks, beta=1, n_iter_max_each_nmf=n_iter_max_each_nmf,
                                                   init_each_nmf=init_multi_layer, delta=delta, return_errors=True,
                                                   verbose=False, deterministic=deterministic, seed=seed)
        reconstruction_errors[:, 0] = e[:, -1]
        Wt = [to_dev_t(w, dev).clone() for w in W]
        Hd = [to_dev(h, dev).clone() for h in H]
    elif init == "custom":
        Wt = [to_dev_t(w, dev).clone() for w in W_0]
        Hd = [to_dev(h, dev).clone() for h in H_0]
        for i in range(L):
            reconstruction_errors[i, 0] = float(eng.betadiv(_layer_data(X, Wt, i), Wt[i], Hd[i], 1))   # deep_nmf.py:50-52
    else:
        raise ValueError("The init method is not supported.")

    lambda_ = 1 / np.array(reconstruction_errors[:, 0])
    global_errors.append(lambda_.T @ reconstruction_errors[:, 0])

    for deep_iteration in range(n_iter_max_deep_loop):
        tic = time.time()
        Wt, Hd, errors = _one_step_dev(eng, X, Wt, Hd, lambda_)
        toc.append(time.time() - tic)

        reconstruction_errors[:, deep_iteration + 1] = lambda_ * errors
        global_errors.append(lambda_.T @ errors)

        if verbose:
            if global_errors[-2] - global_errors[-1] > 0:
                print(f'Normalized sum of errors through layers={global_errors[-1]}, variation={global_errors[-2] - global_errors[-1]}.')
            else:
                print(f'\033[91m Normalized sum of errors through layers={global_errors[-1]}, variation={global_errors[-2] - global_errors[-1]}. \033[0m')

        if deep_iteration > 1 and abs(global_errors[-2] - global_errors[-1]) < tol:
            if verbose:
                print(f'Converged in {deep_iteration} iterations.')
            break

    W_out = [like_input(w.t(), data) for w in Wt]
    H_out = [like_input(h, data) for h in Hd]
    if return_errors:
        return W_out, H_out, reconstruction_errors, toc
    return W_out, H_out


def _layer_data(X, Wt, layer):
    """What layer `layer` factorises: the data for layer 0, else W[layer-1] as an m x r_{layer-1} row-major matrix."""
    return X if layer == 0 else Wt[layer - 1].t().contiguous()


def _normalize_h(Wt, H):
    """normalize_WH(W, H, "H") (normalize_wh.py:8-11) on the transposed storage: rows of H to unit sum, scales into W."""
    s = H.sum(dim=1)
    return Wt * s[:, None], H / s[:, None]


def _one_step_dev(eng, X, Wt, Hd, lambda_):
    """one_step_deep_KL_nmf (deep_nmf.py:84-113) on device factors; returns the new lists and the layer errors (host)."""
    L = len(Wt)
    errs = torch.empty(L, dtype=torch.float64, device=X.device)
    for layer in range(L):
        D = _layer_data(X, Wt, layer)
        Hd[layer] = eng.mu_right(D, Wt[layer], Hd[layer], 1)                        # switch_alternate_mu(..., "H")
        Wt[layer], Hd[layer] = _normalize_h(Wt[layer], Hd[layer])
        if layer == L - 1:
            Wt[layer] = eng.mu_left(D, Wt[layer], Hd[layer], 1)                     # switch_alternate_mu(..., "W")
        else:
            lam = lambda_[layer + 1] / lambda_[layer]
            # (W_{l+1} H_{l+1})^T = H_{l+1}^T W_{l+1}^T : r_l x m, a rank-sized left operand against the long factor
            WHn_t = eng.small_gemm(Hd[layer + 1].t().contiguous(), Wt[layer + 1])
            Wt[layer] = _deep_kl_mu_dev(eng, D, Wt[layer], Hd[layer], WHn_t, lam)
        eng.betadiv(D, Wt[layer], Hd[layer], 1, out=errs[layer:layer + 1])          # kl_divergence(data_l, W_l H_l)
    return Wt, Hd, errs.cpu().numpy()


def one_step_deep_KL_nmf(data, W, H, all_ranks, lambda_, delta):
    """deep_nmf.py:84-113 with the reference's signature (lists of NumPy arrays or tensors in, same kind out)."""
    dev = device_of(data, *W, *H)
    eng = _engine.get_engine(dev)
    Wt = [to_dev_t(w, dev).clone() for w in W]
    Hd = [to_dev(h, dev).clone() for h in H]
    Wt, Hd, errors = _one_step_dev(eng, to_dev(data, dev), Wt, Hd, np.asarray(lambda_, dtype=np.float64))
    return [like_input(w.t(), W[i]) for i, w in enumerate(Wt)], [like_input(h, H[i]) for i, h in enumerate(Hd)], list(errors)
