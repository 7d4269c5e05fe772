"""normalize_WH -- mirror of nn_fac/utils/normalize_wh.py:6-22 (the scaling used between the layers of multilayer NMF).

Only this function of the reference module is on a path that reaches the accelerated NMF (multilayer_nmf.py:47-51); the
simplex-projection helpers of the same file (:24-162) belong to simplex / deep NMF, which are out of scope.
NumPy in -> NumPy out; torch tensors (host or device) in -> tensors out, same arithmetic (a row / column scaling).
"""
import numpy as np
import torch


def normalize_WH(W, H, matrix):
    if matrix == "H":
        if isinstance(H, torch.Tensor):
            scalH = H.sum(dim=1)
            return W * scalH.unsqueeze(0), H / scalH.unsqueeze(1)
        scalH = np.sum(H, axis=1)
        H = np.diag(1 / scalH) @ H            # normalize_wh.py:9-10
        W = W @ np.diag(scalH)
    elif matrix == "W":
        if isinstance(W, torch.Tensor):
            scalW = W.sum(dim=0)
            return W / scalW.unsqueeze(0), H * scalW.unsqueeze(1)
        scalW = np.sum(W, axis=0)
        H = np.diag(scalW) @ H                # normalize_wh.py:14-15
        W = W @ np.diag(1 / scalW)
    else:
        raise ValueError(f"Matrix must be either 'W' or 'H', but it is {matrix}")
    return W, H
