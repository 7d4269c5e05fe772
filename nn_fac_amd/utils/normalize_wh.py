"""normalize_WH -- the scaling multilayer NMF applies between its layers (nn_fac/utils/normalize_wh.py:6-22).

Only this function of the reference module is on a path that reaches the accelerated NMF (multilayer_nmf.py:47-51); the
simplex-projection helpers of the same file (:24-162) belong to simplex / deep NMF, which are out of scope.
The product W H is unchanged: one factor is divided by the sums of the chosen factor, the other multiplied by them.
NumPy in -> NumPy out; torch tensors (host or device) in -> tensors out.
"""
import numpy as np
import torch


def normalize_WH(W, H, matrix):
    if matrix not in ("W", "H"):
        raise ValueError(f"Matrix must be either 'W' or 'H', but it is {matrix}")
    is_t = isinstance(W, torch.Tensor) or isinstance(H, torch.Tensor)
    if matrix == "H":                       # rows of H sum to one (normalize_wh.py:8-10)
        scale = H.sum(dim=1) if is_t else np.sum(H, axis=1)
        return W * scale[None, :], H / scale[:, None]
    scale = W.sum(dim=0) if is_t else np.sum(W, axis=0)      # columns of W sum to one (:13-15)
    return W / scale[None, :], H * scale[:, None]
