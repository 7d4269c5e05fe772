"""Boundary conversions: the drop-in functions accept NumPy arrays or torch tensors and answer in kind."""
import numpy as np
import torch

from .utils.errors import EngineError


def device_of(*xs):
    for x in xs:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return x.device
    if not torch.cuda.is_available():
        raise EngineError("no ROCm device available: the nn_fac_amd engine is GPU-only (no CPU fallback)")
    return torch.device(f"cuda:{torch.cuda.current_device()}")


def to_dev(x, device):
    """float32, device-resident, unit inner stride.  NumPy (any float dtype) and CPU tensors are uploaded;
    a conforming device tensor is returned as is (no copy)."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=torch.float32)
    else:
        a = np.asarray(x)
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    if t.dim() >= 1 and t.stride(-1) != 1:
        t = t.contiguous()
    return t


def to_dev_t(x, device):
    """Device float32 copy/view of x^T with unit inner stride (x: m x r -> r x m).  No copy if x is already a
    transposed view of such a tensor."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=torch.float32).t()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x).T, dtype=np.float32)).to(device)
    if t.stride(-1) != 1:
        t = t.contiguous()
    return t


def like_input(t, proto):
    """Return device tensor `t` in the type/dtype of the caller's `proto` (ndarray -> ndarray of its dtype)."""
    if isinstance(proto, torch.Tensor):
        return t if proto.is_cuda else t.to(proto.device, proto.dtype)
    dt = np.asarray(proto).dtype if proto is not None else np.float64
    if not np.issubdtype(dt, np.floating):
        dt = np.float64
    return t.detach().cpu().numpy().astype(dt, copy=False)
