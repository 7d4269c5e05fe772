"""deep_KL_mu on the MI355X engine -- same signature as nn_fac/update_rules/deep_mu.py:8-14.

    a = ONES @ H_L^T - lambda * log(WH_Lp1)                  (m x r; every row of ONES @ H_L^T is the row sums of H_L)
    b = W_L * ((W_Lm1 / (W_L H_L)) @ H_L^T)                  (the KL numerator of the multiplicative update)
    W_L <- max(1e-12, (b / lambda) / (lambertw(b exp(a / lambda) / lambda).real + 1e-12))

Device pieces (include/nnfac_hip.h): nnf_mu_left_num_f32 (the fused two-MFMA kernel of mu_betadivmin's left update, numerator
only), nnf_deep_kl_apply_f32 (element-wise tail; principal Lambert W in fp64 from the logarithm of its argument).
"""
import torch

from .. import engine as _engine
from .._convert import device_of, to_dev, to_dev_t, like_input

eps = 1e-12


def deep_KL_mu(W_Lm1, W_L, H_L, WH_Lp1, lambda_):
    dev = device_of(W_Lm1, W_L, H_L, WH_Lp1)
    eng = _engine.get_engine(dev)
    out = _deep_kl_mu_dev(eng, to_dev(W_Lm1, dev), to_dev_t(W_L, dev), to_dev(H_L, dev), to_dev_t(WH_Lp1, dev), lambda_)
    return like_input(out.t(), W_L)


def _deep_kl_mu_dev(eng, X, Ut, V, WHnext_t, lambda_):
    """Device form: X = W_{L-1} (m x n), Ut = W_L^T (r x m), V = H_L (r x n), WHnext_t = (W_{L+1} H_{L+1})^T (r x m)."""
    num = eng.mu_left_num(X, Ut, V)
    hsum = V.sum(dim=1, dtype=torch.float64)          # ONES @ H_L^T (deep_mu.py:9)
    return eng.deep_kl_apply(Ut, num, hsum, WHnext_t, lambda_)
