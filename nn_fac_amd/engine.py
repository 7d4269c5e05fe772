"""Thin tensor-level wrapper over the C ABI: torch tensors in, raw device pointers out.

PyTorch is plumbing here (device memory, streams); every arithmetic statement of the hot path runs in
libnnfac_hip.so.  Factors are kept "transposed" on the device (Ut: r x m, V: r x n), the layout hals_nnls_acc works on.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from .utils import errors as err
from .utils.errors import EngineError

HALS_SPARSITY, HALS_NORMALIZE, HALS_NONZERO = 1, 2, 4
ST_EPS, ST_CNT, ST_EPS0, ST_ERR, ST_WORDS = 0, 1, 2, 3, 8

_engines = {}
_lock = threading.Lock()

MAX_RANK = 128      # NNF_MAX_RANK of include/nnfac_hip.h: up to here one launch per product / cost pass and the register- or
#                     LDS-resident sweep kernels; above, matrix factorisations go on in rank chunks (see check_rank)


def check_rank(r, where):
    """The reference accepts any rank up to min(shape) (nn_fac/nmf.py:175-178; nnls.py:156-170 loops `range(r)`).  The matrix
    path (nmf, hals_nnls_acc, mu_betadivmin) and NTF (ntf, compute_ntf, one_ntf_step) follow it: above 128 the contractions,
    MTTKRPs and cost passes walk the rank in chunks of 128 and the sweeps run in the generic kernel ("Ranks
    above 128") -- callers on those paths do not call this.  The TUCKER kernels (core contractions along the middle axis, the
    projected-gradient core update) are built for ranks <= 128: said at the boundary, before anything is uploaded or launched,
    instead of an NNF_ERR_UNSUPPORTED status from deep inside an iteration."""
    r = int(r)
    if r > MAX_RANK:
        raise EngineError(f"{where}: rank {r} is above the {MAX_RANK} the Tucker kernels (core contractions, core update) of "
                          f"nn_fac_amd are built for; nmf, ntf, hals_nnls_acc and mu_betadivmin take any rank")


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _ld(t):
    """Leading dimension (elements) of a 2-D tensor for the C ABI.  PyTorch leaves the stride of a size-1 dimension
    arbitrary (a 1 x m factor -- rank 1 -- can report stride(0) == 1); with a single row it is never used to address
    anything, so any value >= the row length is valid."""
    return t.stride(0) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


def _chk2d(t, name):
    # (the stride of a size-1 dimension is arbitrary in PyTorch: a k x 1 tensor need not report stride(1) == 1)
    if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or (t.shape[1] > 1 and t.stride(1) != 1):
        raise EngineError(f"{name}: expected a 2-D float32 device tensor with unit inner stride, got "
                          f"{tuple(t.shape)} {t.dtype} {t.device} strides {t.stride()}")
    return t


class KernelTime(float):
    """Mean launch duration in ms (the float) + the spread of the samples it came from."""

    @classmethod
    def of(cls, samples):
        ts = sorted(float(t) for t in samples)
        k = cls(sum(ts) / len(ts))
        k.n, k.median, k.min, k.max = len(ts), ts[len(ts) // 2], ts[0], ts[-1]
        return k

    def stats(self):
        return {"samples": self.n, "mean_ms": float(self), "median_ms": self.median, "min_ms": self.min, "max_ms": self.max}


class Engine:
    """One context (workspace + device properties) per device."""

    def __init__(self, device, workspace_bytes=0):
        if not torch.cuda.is_available():
            raise EngineError("no ROCm device available: the nn_fac_amd engine is GPU-only (no CPU fallback)")
        self.device = torch.device(device)
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.nnf_ctx_create(C.byref(h), self.device.index or 0, workspace_bytes), "nnf_ctx_create")
        self.ctx = h
        self._resident_cols = {}

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.nnf_ctx_destroy(self.ctx)
        except Exception:  # interpreter shutdown
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- contractions -------------------------------------------------------------------------------
    def gram(self, A, out=None, out64=None):
        """A (r x K) -> A A^T (r x r).  out64 (optional, contiguous r x r float64 device tensor): also receives the sums before
        they are rounded to fp32 (nnf_gram_f64_f32), for gram_cost(..., UtU64=...)."""
        _chk2d(A, "gram A")
        r, K = A.shape
        G = out if out is not None else torch.empty((r, r), dtype=torch.float32, device=A.device)
        if out64 is not None:
            if out64.dtype != torch.float64 or not out64.is_contiguous() or out64.numel() < r * r or out64.device != A.device:
                raise EngineError("gram: out64 must be a contiguous float64 device tensor of r*r elements")
            _lib.check(self.lib.nnf_gram_f64_f32(self.ctx, _ptr(A), r, K, _ld(A), _ptr(G), _ld(G), _ptr(out64), self._stream()),
                       "nnf_gram_f64_f32")
            return G
        _lib.check(self.lib.nnf_gram_f32(self.ctx, _ptr(A), r, K, _ld(A), _ptr(G), _ld(G), self._stream()),
                   "nnf_gram_f32")
        return G

    def xht(self, X, V, out=None):
        """V (r x n), X (m x n) -> V X^T (r x m)."""
        _chk2d(X, "xht X"), _chk2d(V, "xht V")
        m, n = X.shape
        r = V.shape[0]
        if V.shape[1] != n:
            raise EngineError("xht: shape mismatch")
        O = out if out is not None else torch.empty((r, m), dtype=torch.float32, device=X.device)
        _lib.check(self.lib.nnf_xht_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(V), r, _ld(V), _ptr(O),
                                        _ld(O), self._stream()), "nnf_xht_f32")
        return O

    def xty(self, X, Ut, out=None):
        """Ut (r x m), X (m x n) -> Ut X (r x n)."""
        _chk2d(X, "xty X"), _chk2d(Ut, "xty Ut")
        m, n = X.shape
        r = Ut.shape[0]
        if Ut.shape[1] != m:
            raise EngineError("xty: shape mismatch")
        O = out if out is not None else torch.empty((r, n), dtype=torch.float32, device=X.device)
        _lib.check(self.lib.nnf_xty_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), r, _ld(Ut), _ptr(O),
                                        _ld(O), self._stream()), "nnf_xty_f32")
        return O

    def frob_resid(self, X, Ut, V, out=None):
        """sum (X - Ut^T V)^2 as a 1-element float64 device tensor."""
        _chk2d(X, "frob X"), _chk2d(Ut, "frob Ut"), _chk2d(V, "frob V")
        m, n = X.shape
        r = Ut.shape[0]
        if Ut.shape[1] != m or V.shape != (r, n):
            raise EngineError("frob_resid: shape mismatch")
        o = out if out is not None else torch.empty(1, dtype=torch.float64, device=X.device)
        self._model_scratch(m, n, r)
        _lib.check(self.lib.nnf_frob_resid_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V),
                                               _ld(V), r, _ptr(o), self._stream()), "nnf_frob_resid_f32")
        return o

    def _model_scratch(self, m, n, r):
        """Ranks above 128: the cost passes build the model U V over rank chunks in an m x n float32 buffer (nnf_ctx_set_scratch);
        kept between calls, grown when a larger one is asked for."""
        if r <= MAX_RANK:
            return
        need = int(m) * ((int(n) + 3) // 4 * 4) * 4
        cur = getattr(self, "_scratch", None)
        if cur is None or cur.numel() < need:
            torch.cuda.current_stream(self.device).synchronize()      # (earlier calls may still read the old buffer)
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.nnf_ctx_set_scratch(self.ctx, _ptr(self._scratch), need), "nnf_ctx_set_scratch")

    def gram_cost(self, V, UtM, UtU, normx2, out, UtU_b=None, rounding=None, UtU64=None):
        """||X - U V||^2 through the Gram identity (nnf_nmf_gram_cost_cal_f32): `normx2` a 1-element float64 device tensor holding
        ||X||^2, `out` >= 3 float64 on the device: {cost, 1 if the fp32 operands do not carry it to 5e-4, error estimate}.
        rounding = (relative rms, |relative mean|) of the rounding error of a UtM entry (default: 6e-8, 0)."""
        _chk2d(V, "gram_cost V"), _chk2d(UtM, "gram_cost UtM"), _chk2d(UtU, "gram_cost UtU")
        r, n = V.shape
        if UtM.shape != (r, n) or UtU.shape[0] < r or UtU.shape[1] < r or normx2.dtype != torch.float64 or out.dtype != torch.float64 \
                or out.numel() < 3:
            raise EngineError("gram_cost: shape / dtype mismatch")
        if UtU_b is not None and (UtU_b.shape != UtU.shape or _ld(UtU_b) != _ld(UtU)):
            raise EngineError("gram_cost: the two Grams of a Hadamard pair must share shape and leading dimension")
        sa, ba = (6e-8, 0.0) if rounding is None else (float(rounding[0]), float(rounding[1]))
        if UtU64 is not None and UtU_b is None:
            # the quadratic form on the Gram before its rounding to fp32; rounding[2] = relative rms error of a UtU64 entry
            sg = float(rounding[2]) if rounding is not None and len(rounding) > 2 else 1e-8
            _lib.check(self.lib.nnf_nmf_gram_cost_g64_f32(self.ctx, _ptr(V), _ld(V), _ptr(UtM), _ld(UtM), _ptr(UtU), _ptr(UtU64),
                                                          _ld(UtU), r, n, _ptr(normx2), sa, ba, sg, _ptr(out), self._stream()),
                       "nnf_nmf_gram_cost_g64_f32")
            return out
        _lib.check(self.lib.nnf_nmf_gram_cost_cal_f32(self.ctx, _ptr(V), _ld(V), _ptr(UtM), _ld(UtM), _ptr(UtU),
                                                      _ptr(UtU_b) if UtU_b is not None else None, _ld(UtU), r, n,
                                                      _ptr(normx2), sa, ba, _ptr(out), self._stream()), "nnf_nmf_gram_cost_cal_f32")
        return out

    def cross_rounding(self, X, Ut, blocks=16):
        """(relative rms, |relative mean|) of the rounding error the W^T X kernel leaves in an entry of U^T X at THIS shape: the
        product summed in one piece (row splits of the launch plan, fp32 accumulators per workgroup) against the same product
        summed over `blocks` row blocks in float64 -- whose fp32 chains are `blocks` times shorter, so that the difference is
        the long chains' error to ~1/sqrt(blocks).  Two passes over X, once per run (nmf.run_steps)."""
        m = int(X.shape[0])
        full = self.xty(X, Ut).double()
        acc = torch.zeros_like(full)
        step = -(-m // int(blocks))
        step = -(-step // 256) * 256
        for lo in range(0, m, step):
            hi = min(m, lo + step)
            acc += self.xty(X[lo:hi], Ut[:, lo:hi]).double()
        rel = (full - acc) / acc.clamp_min(1e-300)
        rel = torch.where(acc > 0, rel, torch.zeros_like(rel))
        return float(rel.pow(2).mean().sqrt()), abs(float(rel.mean()))

    def gram_rounding(self, Ut, blocks=16):
        """Relative rms error of an entry of the fp64 Gram of nnf_gram_f64_f32 at THIS shape (the fp32 accumulation inside a
        split of the Gram kernel): the Gram in one piece against the sum of the Grams of `blocks` column blocks, whose chains
        are `blocks` times shorter.  r x r work, once per run."""
        r, m = Ut.shape
        full = torch.empty((r, r), dtype=torch.float64, device=Ut.device)
        self.gram(Ut, out64=full)
        acc, part = torch.zeros_like(full), torch.empty_like(full)
        step = -(-m // int(blocks))
        step = -(-step // 256) * 256
        for lo in range(0, m, step):
            self.gram(Ut[:, lo:min(m, lo + step)], out64=part)
            acc += part
        rel = (full - acc) / acc.abs().clamp_min(1e-300)
        rel = torch.where(acc != 0, rel, torch.zeros_like(rel))
        return float(rel.pow(2).mean().sqrt())

    def dot(self, A, B):
        _chk2d(A, "dot A"), _chk2d(B, "dot B")
        o = torch.empty(1, dtype=torch.float64, device=A.device)
        _lib.check(self.lib.nnf_dot_f32(self.ctx, _ptr(A), _ld(A), _ptr(B), _ld(B), A.shape[0], A.shape[1],
                                        _ptr(o), self._stream()), "nnf_dot_f32")
        return o

    def hadamard(self, A, B, out=None):
        A, B = A.contiguous(), B.contiguous()
        Cc = out if out is not None else torch.empty_like(A)
        _lib.check(self.lib.nnf_hadamard_f32(self.ctx, _ptr(A), _ptr(B), _ptr(Cc), A.numel(), self._stream()),
                   "nnf_hadamard_f32")
        return Cc

    # ---- HALS ---------------------------------------------------------------------------------------
    @staticmethod
    def _hals_flags(sparsity, normalize, nonzero):
        f = 0
        if sparsity is not None:
            f |= HALS_SPARSITY
        if normalize:
            f |= HALS_NORMALIZE
        if nonzero:
            f |= HALS_NONZERO
        return f

    ROWSYNC_MAX_COLUMNS = 131072          # normalize / nonzero: one column per resident thread (k_hals.hip generic path)

    def _check_rowsync_columns(self, rowsync, ncols):
        """normalize=True / nonzero=True need a grid-wide reduction per ROW update, which the generic kernel does with every
        column resident (one per thread, 4 x 128-thread workgroups per CU): say so here instead of a bare status code.
        (persistent solves go on beyond that limit, row by row from the host: _hals_solve_rowwalk; fixed-count sweeps do not.)"""
        if rowsync and ncols > self.ROWSYNC_MAX_COLUMNS:
            raise EngineError(f"fixed-count sweeps with normalize=True / nonzero=True are built for at most "
                              f"{self.ROWSYNC_MAX_COLUMNS} columns (got {ncols}): every row update needs all columns "
                              f"resident on the device at once; normalise the shorter factor, or split the columns and "
                              f"normalise on the host between outer iterations")

    def _hals_solve_rowwalk(self, UtM, UtU, V, max_sweeps, delta, sparsity, st, normalize=True, nonzero=False):
        """hals_nnls_acc(..., normalize=True) on more columns than the generic kernel keeps resident: the rows are walked from
        the host -- row update over all columns (nnf_hals_row_update_f32), row norm, scaling (nnf_hals_row_scale_f32), r x 2
        launches per sweep and one host round trip per sweep for the stopping rule of nnls.py:156 -- the one-device form of the
        row-sharded protocol (dist.sharded_hals_solve_rownorm, which this calls without a group).  Correct and slow (a sweep of a
        rank-50 factor is 100 launches): the option is on no BASELINE configuration."""
        from . import dist as _dist
        eps, cnt, eps0 = _dist.sharded_hals_solve_rownorm(self, UtM, UtU, V, None, budget=int(max_sweeps), delta=float(delta),
                                                          sparsity=sparsity, normalize=normalize, nonzero=nonzero)
        st[:4] = torch.tensor([eps, float(cnt), eps0, 0.0], dtype=torch.float64)
        return st

    HALS_MAX_SWEEPS_PER_LAUNCH = 1000     # NNF_HALS_MAX_SWEEPS (exchange tags hold the sweep index in 10 bits)

    def hals_solve(self, UtM, UtU, V, max_sweeps, delta=0.01, sparsity=None, normalize=False, nonzero=False,
                   status=None):
        """In-place accelerated HALS on V (r x ncols); returns the 8-double status tensor (device, not synced)."""
        _chk2d(UtM, "hals UtM"), _chk2d(UtU, "hals UtU"), _chk2d(V, "hals V")
        r, ncols = V.shape
        if UtM.shape != (r, ncols) or UtU.shape[0] < r or UtU.shape[1] < r:
            raise EngineError("hals_solve: shape mismatch")
        st = status if status is not None else torch.empty(ST_WORDS, dtype=torch.float64, device=V.device)
        flags = self._hals_flags(sparsity, normalize, nonzero)
        if (normalize or nonzero or r > MAX_RANK) and ncols > self.ROWSYNC_MAX_COLUMNS and int(max_sweeps) > 0:
            # (ranks above 128 run in the generic kernel too: one column per resident thread for a persistent solve)
            return self._hals_solve_rowwalk(UtM, UtU, V, max_sweeps, delta, sparsity, st, normalize=normalize, nonzero=nonzero)
        self._check_rowsync_columns(normalize or nonzero, ncols)
        total, first = int(max_sweeps), min(int(max_sweeps), self.HALS_MAX_SWEEPS_PER_LAUNCH)
        _lib.check(self.lib.nnf_hals_solve_f32(self.ctx, _ptr(UtM), _ld(UtM), _ptr(UtU), _ld(UtU), _ptr(V),
                                               _ld(V), r, ncols, first, float(delta), float(sparsity or 0.0), flags,
                                               _ptr(st), self._stream()), "nnf_hals_solve_f32")
        done = first
        while done < total:      # maxiter beyond one launch's tag range: chained launches, no host round trip in between
            step = min(total - done, self.HALS_MAX_SWEEPS_PER_LAUNCH)
            _lib.check(self.lib.nnf_hals_solve_continue_f32(self.ctx, _ptr(UtM), _ld(UtM), _ptr(UtU), _ld(UtU), _ptr(V),
                                                            _ld(V), r, ncols, done, step, float(delta),
                                                            float(sparsity or 0.0), flags, _ptr(st), self._stream()),
                       "nnf_hals_solve_continue_f32")
            done += step
        return st

    def hals_solve_cross(self, UtM, Ga, Gb, V_in, V_out, max_sweeps, delta=0.01, sparsity=None, normalize=False, status=None):
        """hals_solve with the Gram Ga .* Gb (Gb may be None), start values V_in and the result in V_out (ntf.py:442-456)."""
        _chk2d(UtM, "hals UtM"), _chk2d(Ga, "hals Ga"), _chk2d(V_in, "hals V_in"), _chk2d(V_out, "hals V_out")
        r, ncols = V_out.shape
        if UtM.shape != (r, ncols) or V_in.shape != (r, ncols) or Ga.shape[0] < r or (Gb is not None and
                                                                                       (Gb.shape != Ga.shape or _ld(Gb) != _ld(Ga))):
            raise EngineError("hals_solve_cross: shape mismatch")
        st = status if status is not None else torch.empty(ST_WORDS, dtype=torch.float64, device=V_out.device)
        if normalize and ncols > self.ROWSYNC_MAX_COLUMNS and int(max_sweeps) > 0:
            if V_out.data_ptr() != V_in.data_ptr():
                V_out.copy_(V_in)
            return self._hals_solve_rowwalk(UtM, Ga if Gb is None else self.hadamard(Ga[:r, :r], Gb[:r, :r]), V_out, max_sweeps,
                                            delta, sparsity, st)
        self._check_rowsync_columns(normalize, ncols)
        _lib.check(self.lib.nnf_hals_solve_cross_f32(self.ctx, _ptr(UtM), _ld(UtM), _ptr(Ga), _ptr(Gb) if Gb is not None else None,
                                                     _ld(Ga), _ptr(V_in), _ld(V_in), _ptr(V_out), _ld(V_out), r, ncols,
                                                     int(max_sweeps), float(delta), float(sparsity or 0.0),
                                                     self._hals_flags(sparsity, normalize, False), _ptr(st), self._stream()),
                   "nnf_hals_solve_cross_f32")
        return st

    def hals_sweeps(self, UtM, UtU, V, nsweeps, sparsity=None, normalize=False, nonzero=False, snapshots=None, snap_first=0,
                    sweeps_done=0, resid_in=None, resid_out=None):
        """Exactly `nsweeps` in-place sweeps; returns the per-sweep LOCAL sum of squared steps (float64, device).
        snapshots (optional, contiguous float32 [>= nsweeps - snap_first, r, ncols]): block j receives V after sweep
        snap_first + j + 1.  sweeps_done / resid_in / resid_out: this call continues a solve of which `sweeps_done` sweeps have
        run (nnf_hals_sweeps_ex_f32): with the residual state handed on (float32 tensors of hals_resid_floats() elements) the
        chunks of a solve are bit for bit one launch of all its sweeps."""
        _chk2d(UtM, "hals UtM"), _chk2d(UtU, "hals UtU"), _chk2d(V, "hals V")
        r, ncols = V.shape
        if (normalize or nonzero) and ncols > self.ROWSYNC_MAX_COLUMNS and snapshots is None and int(nsweeps) > 0:
            # more columns than the generic kernel keeps resident: the rows are walked from the host, one sweep per call of the
            # row-walk (the wall-clock rule's one-sweep probe of nmf() / hals_nnls_acc(deterministic=False) lands here)
            from . import dist as _dist
            vals = [_dist.sharded_hals_solve_rownorm(self, UtM, UtU, V, None, budget=1, delta=0.0, sparsity=sparsity,
                                                     normalize=normalize, nonzero=nonzero)[0] for _ in range(int(nsweeps))]
            return torch.tensor(vals, dtype=torch.float64, device=V.device)
        self._check_rowsync_columns(normalize or nonzero, ncols)
        nd = torch.zeros(max(int(nsweeps), 1), dtype=torch.float64, device=V.device)
        sp, ss = C.c_void_p(0), 0
        if snapshots is not None:
            if (snapshots.dtype != torch.float32 or not snapshots.is_contiguous() or snapshots.dim() != 3
                    or snapshots.shape[0] < nsweeps - int(snap_first) or tuple(snapshots.shape[1:]) != (r, ncols)):
                raise EngineError("hals_sweeps: snapshots must be a contiguous float32 [>= nsweeps - snap_first, r, ncols] tensor")
            sp, ss = _ptr(snapshots), snapshots.stride(0)
        need = self.hals_resid_floats(r, ncols) if (resid_in is not None or resid_out is not None) else 0
        for t in (resid_in, resid_out):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < need or t.device != V.device):
                raise EngineError("hals_sweeps: a residual-state buffer must be a contiguous float32 tensor of hals_resid_floats() elements")
        ri = _ptr(resid_in) if (resid_in is not None and need > 0) else C.c_void_p(0)
        ro = _ptr(resid_out) if (resid_out is not None and need > 0) else C.c_void_p(0)
        _lib.check(self.lib.nnf_hals_sweeps_ex_f32(self.ctx, _ptr(UtM), _ld(UtM), _ptr(UtU), _ld(UtU), _ptr(V),
                                                   _ld(V), r, ncols, int(nsweeps), int(sweeps_done), float(sparsity or 0.0),
                                                   self._hals_flags(sparsity, normalize, nonzero), _ptr(nd), sp, ss,
                                                   int(snap_first), ri, ro, self._stream()), "nnf_hals_sweeps_ex_f32")
        return nd[:int(nsweeps)]

    def hals_resid_floats(self, r, ncols):
        """Elements of one residual-state buffer for blind chunks on an r x ncols factor (0: the layout that runs keeps no state)."""
        out = C.c_int64(0)
        _lib.check(self.lib.nnf_hals_resid_floats(self.ctx, int(r), int(ncols), C.byref(out)), "nnf_hals_resid_floats")
        return int(out.value)

    def hals_row_update(self, UtM, UtU, V, k, sparsity=None, out=None):
        """Row k of V (this rank's columns) gets the update of nnls.py:162-170, in place; returns a 2-element float64 device
        tensor {sum of squared steps, sum of squares of the updated row} (row-sharded solves with normalize, dist.py)."""
        _chk2d(UtM, "hals UtM"), _chk2d(UtU, "hals UtU"), _chk2d(V, "hals V")
        r, ncols = V.shape
        o = out if out is not None else torch.empty(2, dtype=torch.float64, device=V.device)
        _lib.check(self.lib.nnf_hals_row_update_f32(self.ctx, _ptr(UtM), _ld(UtM), _ptr(UtU), _ld(UtU), _ptr(V), _ld(V), r, ncols,
                                                    int(k), float(sparsity or 0.0), self._hals_flags(sparsity, False, False), _ptr(o),
                                                    self._stream()), "nnf_hals_row_update_f32")
        return o

    def hals_row_scale(self, V, k, normsq, ncols_total):
        """Row k of V /= sqrt(normsq[0]) (a float64 device scalar), or := 1 / sqrt(ncols_total) when it is 0 (nnls.py:181-185)."""
        _chk2d(V, "hals V")
        _lib.check(self.lib.nnf_hals_row_scale_f32(self.ctx, _ptr(V), _ld(V), V.shape[1], int(k), _ptr(normsq), int(ncols_total),
                                                   self._stream()), "nnf_hals_row_scale_f32")
        return V

    def hals_resident_columns(self, r):
        """Columns the register-resident sweep kernel of rank r holds on this device (blind chunks with snapshots, and fast
        sweeps at all, need column blocks of at most this size: dist.sharded_hals_solve)."""
        hit = self._resident_cols.get(int(r))
        if hit is None:
            out = C.c_int64(0)
            _lib.check(self.lib.nnf_hals_resident_columns(self.ctx, int(r), C.byref(out)), "nnf_hals_resident_columns")
            hit = self._resident_cols[int(r)] = int(out.value)
        return hit

    def hals_stop_restore(self, sums, head, budget, delta, V, snapshots, status):
        """Device-side replay of the stopping rule over the all-reduced per-sweep sums of a blind chunk (dist.py)."""
        _chk2d(V, "hals V")
        r, ncols = V.shape
        n = int(sums.numel())
        if sums.dtype != torch.float64 or not sums.is_contiguous() or status.dtype != torch.float64:
            raise EngineError("hals_stop_restore: sums / status must be float64 device tensors")
        sp = _ptr(snapshots) if snapshots is not None else C.c_void_p(0)
        ss = snapshots.stride(0) if snapshots is not None else 0
        _lib.check(self.lib.nnf_hals_stop_restore_f32(self.ctx, _ptr(sums), n, int(head), int(budget), float(delta), _ptr(V),
                                                      _ld(V), r, ncols, sp, ss, _ptr(status), self._stream()),
                   "nnf_hals_stop_restore_f32")
        return status

    # ---- MU / beta-divergence -----------------------------------------------------------------------
    MU_FUSED_MAX_RANK = 64   # the fused two-MFMA kernels are built for r <= 64 (beta = 2 has no limit: Gram form)

    def _mu_large_rank(self, X, Ut, V, beta, side):
        """r > 64 (beta != 2) or r > 128 (any beta; the model then builds up over rank chunks inside R1): one pass writes the element-wise operands R1 = X.*(UV)^(beta-2), R2 = (UV)^(beta-1)
        (m x n device scratch each), then the numerator / denominator are plain X H^T ('left') or W^T X ('right') products.
        Returns (num, den or None, den_vec or None) like mu_right_accum."""
        m, n = X.shape
        r = Ut.shape[0]
        kl = float(beta) == 1.0
        R1 = torch.empty((m, n), dtype=torch.float32, device=X.device)
        R2 = None if kl else torch.empty((m, n), dtype=torch.float32, device=X.device)
        _lib.check(self.lib.nnf_mu_ratio_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V),
                                             _ld(V), r, float(beta), _ptr(R1), _ptr(R2) if R2 is not None else None,
                                             _ld(R1), self._stream()), "nnf_mu_ratio_f32")
        if side == "left":
            num = self.xht(R1, V)
            den = None if kl else self.xht(R2, V)
            dvec = V.sum(dim=1, dtype=torch.float64) if kl else None          # mu.py:86-87
        else:
            num = self.xty(R1, Ut)
            den = None if kl else self.xty(R2, Ut)
            dvec = Ut.sum(dim=1, dtype=torch.float64) if kl else None
        return num, den, dvec

    def mu_left(self, X, Ut, V, beta, out=None, cost_out=None):
        """`cost_out` (1-element float64 device tensor; beta = 1, r <= 64 only): also receives beta_divergence(X, U V, 1) of
        the INPUT factors (nnf_mu_left_kl_cost_f32)."""
        _chk2d(X, "mu X"), _chk2d(Ut, "mu Ut"), _chk2d(V, "mu V")
        m, n = X.shape
        r = Ut.shape[0]
        if cost_out is not None:
            if float(beta) != 1.0 or r > self.MU_FUSED_MAX_RANK:
                raise EngineError("mu_left: the fused cost is built for beta = 1 and r <= 64")
            O = out if out is not None else torch.empty_like(Ut)
            _lib.check(self.lib.nnf_mu_left_kl_cost_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V), _ld(V),
                                                        r, _ptr(O), _ld(O), _ptr(cost_out), self._stream()),
                       "nnf_mu_left_kl_cost_f32")
            return O
        if (r > self.MU_FUSED_MAX_RANK and float(beta) != 2.0) or r > MAX_RANK:
            num, den, dvec = self._mu_large_rank(X, Ut, V, beta, "left")
            return self.mu_apply(Ut, num, den, dvec, beta, out=out)
        O = out if out is not None else torch.empty_like(Ut)
        _lib.check(self.lib.nnf_mu_left_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V),
                                            _ld(V), r, float(beta), _ptr(O), _ld(O), self._stream()),
                   "nnf_mu_left_f32")
        return O

    def mu_right(self, X, Ut, V, beta, out=None):
        _chk2d(X, "mu X"), _chk2d(Ut, "mu Ut"), _chk2d(V, "mu V")
        m, n = X.shape
        r = Ut.shape[0]
        if (r > self.MU_FUSED_MAX_RANK and float(beta) != 2.0) or r > MAX_RANK:
            num, den, dvec = self._mu_large_rank(X, Ut, V, beta, "right")
            return self.mu_apply(V, num, den, dvec, beta, out=out)
        O = out if out is not None else torch.empty_like(V)
        _lib.check(self.lib.nnf_mu_right_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V),
                                             _ld(V), r, float(beta), _ptr(O), _ld(O), self._stream()),
                   "nnf_mu_right_f32")
        return O

    def mu_right_accum(self, X, Ut, V, beta):
        """This row block's numerator / denominator of the right update (row-sharded runs): returns (num r x n,
        den r x n or None, den_vec r doubles or None); all three are sums over the rows and all-reduce additively."""
        _chk2d(X, "mu X"), _chk2d(Ut, "mu Ut"), _chk2d(V, "mu V")
        m, n = X.shape
        r = Ut.shape[0]
        if (r > self.MU_FUSED_MAX_RANK and float(beta) != 2.0) or r > MAX_RANK:
            return self._mu_large_rank(X, Ut, V, beta, "right")
        num = torch.empty((r, n), dtype=torch.float32, device=X.device)
        den = torch.empty((r, n), dtype=torch.float32, device=X.device) if float(beta) != 1.0 else None
        dvec = torch.empty(r, dtype=torch.float64, device=X.device) if float(beta) == 1.0 else None
        _lib.check(self.lib.nnf_mu_right_accum_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V),
                                                   _ld(V), r, float(beta), _ptr(num), _ld(num),
                                                   _ptr(den) if den is not None else None,
                                                   _ld(den) if den is not None else 0,
                                                   _ptr(dvec) if dvec is not None else None, self._stream()),
                   "nnf_mu_right_accum_f32")
        return num, den, dvec

    def mu_apply(self, F, num, den, den_vec, beta, out=None):
        """out = max(F * (num/den)^gamma(beta), 1e-12) (mu.py:84-97) from already reduced numerator / denominator."""
        _chk2d(F, "mu F"), _chk2d(num, "mu num")
        r, cols = F.shape
        O = out if out is not None else torch.empty_like(F)
        _lib.check(self.lib.nnf_mu_apply_f32(self.ctx, _ptr(F), _ld(F), r, cols, _ptr(num), _ld(num),
                                             _ptr(den) if den is not None else None,
                                             _ld(den) if den is not None else 0,
                                             _ptr(den_vec) if den_vec is not None else None, float(beta), _ptr(O),
                                             _ld(O), self._stream()), "nnf_mu_apply_f32")
        return O

    # ---- deep KL-NMF (deep_mu.py:8-14) -------------------------------------------------------------
    def mu_left_num(self, X, Ut, V, out=None):
        """Raw KL numerator of the left update: num[k,i] = sum_j (X[i,j] / (UV)[i,j]) V[k,j]  (r x m)."""
        _chk2d(X, "mu X"), _chk2d(Ut, "mu Ut"), _chk2d(V, "mu V")
        m, n = X.shape
        r = Ut.shape[0]
        if Ut.shape[1] != m or V.shape != (r, n):
            raise EngineError("mu_left_num: shape mismatch")
        O = out if out is not None else torch.empty((r, m), dtype=torch.float32, device=X.device)
        _lib.check(self.lib.nnf_mu_left_num_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V), _ld(V), r,
                                                _ptr(O), _ld(O), self._stream()), "nnf_mu_left_num_f32")
        return O

    def small_gemm(self, A, B, out=None):
        """A (p x q, q <= 2048) @ B (q x cols) -> p x cols: a rank-sized left operand against a wide matrix."""
        _chk2d(A, "small_gemm A"), _chk2d(B, "small_gemm B")
        p, q = A.shape
        if B.shape[0] != q:
            raise EngineError("small_gemm: shape mismatch")
        cols = B.shape[1]
        O = out if out is not None else torch.empty((p, cols), dtype=torch.float32, device=A.device)
        _lib.check(self.lib.nnf_small_gemm_f32(self.ctx, _ptr(A), _ld(A), p, q, _ptr(B), _ld(B), cols, _ptr(O), _ld(O),
                                               self._stream()), "nnf_small_gemm_f32")
        return O

    def deep_kl_apply(self, Ft, num, hsum, WHnext_t, lam, out=None):
        """max(1e-12, (b/lam) / (W0(b exp(a/lam)/lam) + 1e-12)) with b = Ft .* num, a = hsum[k] - lam log(WHnext_t)."""
        _chk2d(Ft, "deep F"), _chk2d(num, "deep num"), _chk2d(WHnext_t, "deep WHnext")
        r, cols = Ft.shape
        if num.shape != (r, cols) or WHnext_t.shape != (r, cols) or hsum.dtype != torch.float64 or hsum.numel() != r:
            raise EngineError("deep_kl_apply: shape mismatch")
        O = out if out is not None else torch.empty_like(Ft)
        _lib.check(self.lib.nnf_deep_kl_apply_f32(self.ctx, _ptr(Ft), _ld(Ft), r, cols, _ptr(num), _ld(num), _ptr(hsum),
                                                  _ptr(WHnext_t), _ld(WHnext_t), float(lam), _ptr(O), _ld(O),
                                                  self._stream()), "nnf_deep_kl_apply_f32")
        return O

    PROBE_KERNELS = {"xty": 0, "xht": 1, "cost": 2, "hals": 3, "mu_left": 4, "mu_right": 5, "mttkrp": 6}

    def set_probe(self, ev_begin=None, ev_end=None, kernel="xty"):
        """Measurement hook (bench.py): two torch.cuda.Event(enable_timing=True) that the library records on the launch
        stream right before and right after the main kernel named by `kernel` (NNF_PROBE_* of include/nnfac_hip.h); call
        with no arguments to remove them.  The events must have been recorded once already (torch creates the underlying
        hipEvent_t lazily)."""
        b = C.c_void_p(ev_begin.cuda_event) if ev_begin is not None else None
        e = C.c_void_p(ev_end.cuda_event) if ev_end is not None else None
        _lib.check(self.lib.nnf_ctx_set_probe_kernel(self.ctx, self.PROBE_KERNELS[kernel]), "nnf_ctx_set_probe_kernel")
        _lib.check(self.lib.nnf_ctx_set_probe(self.ctx, b, e), "nnf_ctx_set_probe")

    def set_probe_ring(self, pairs=None, kernel="xty"):
        """Measurement hook for a whole timed region (bench.py): `pairs` = [(begin, end), ...] of already recorded
        torch.cuda.Event(enable_timing=True); the i-th launch of `kernel` from now on records pair i on its launch stream
        (launches beyond the list record nothing).  None removes the ring (nnf_ctx_set_probe_ring)."""
        _lib.check(self.lib.nnf_ctx_set_probe_kernel(self.ctx, self.PROBE_KERNELS[kernel]), "nnf_ctx_set_probe_kernel")
        if not pairs:
            _lib.check(self.lib.nnf_ctx_set_probe_ring(self.ctx, None, 0), "nnf_ctx_set_probe_ring")
            return
        arr = (C.c_void_p * (2 * len(pairs)))(*[C.c_void_p(e.cuda_event) for pr in pairs for e in pr])
        _lib.check(self.lib.nnf_ctx_set_probe_ring(self.ctx, arr, len(pairs)), "nnf_ctx_set_probe_ring")

    def probe_ring_count(self):
        """Pairs of the current ring recorded so far."""
        return int(self.lib.nnf_ctx_probe_ring_count(self.ctx))

    @staticmethod
    def probe_pairs(n, stream):
        """n (begin, end) pairs of timing events, each recorded once on `stream` (torch creates the hipEvent_t lazily)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record(stream)
            b.record(stream)
        return evs

    def time_kernel(self, kernel, fn, reps=20):
        """Duration (ms) of the main kernel `kernel` over `reps` calls of `fn` (which must launch it exactly once on the
        current stream), measured with HIP events recorded by the library immediately around that kernel.  Returns a
        `KernelTime`: the plain MEAN over all samples as a float (nothing is dropped), with .median / .min / .max / .n."""
        stream = torch.cuda.current_stream(self.device)
        evs = self.probe_pairs(reps, stream)
        fn()
        stream.synchronize()
        try:
            self.set_probe_ring(evs, kernel)
            for _ in range(reps):
                fn()
        finally:
            self.set_probe_ring(None, kernel)
        stream.synchronize()
        return KernelTime.of([a.elapsed_time(b) for a, b in evs])

    # ---- NTD -----------------------------------------------------------------------------------------
    def mttkrp3_from_partial(self, Y, Ft, axis, out=None):
        """Y (R x A x B, contiguous) contracted with the rows of Ft over `axis` (1: A, 2: B) -> R x (the other extent)."""
        if Y.dim() != 3 or Y.dtype != torch.float32 or not Y.is_contiguous():
            raise EngineError("mttkrp3_from_partial: Y must be a contiguous 3-way float32 tensor")
        _chk2d(Ft, "partial factor")
        R, A, B = Y.shape
        other = B if axis == 1 else A
        if Ft.shape != (R, A if axis == 1 else B):
            raise EngineError("mttkrp3_from_partial: shape mismatch")
        O = out if out is not None else torch.empty((R, other), dtype=torch.float32, device=Y.device)
        _lib.check(self.lib.nnf_mttkrp3_from_partial_f32(self.ctx, _ptr(Y), A, B, _ptr(Ft), _ld(Ft), R, int(axis), _ptr(O),
                                                         _ld(O), self._stream()), "nnf_mttkrp3_from_partial_f32")
        return O

    def ttm3(self, T, Ft, mode, out=None):
        """T x_mode F^T for a contiguous 3-way tensor and a transposed factor Ft (r x I_mode).
        Result layout: mode 0 -> (r, J, K); mode 1 -> (I, r, K); mode 2 -> (r, I, J)  (include/nnfac_hip.h)."""
        if T.dim() != 3 or not T.is_contiguous():
            raise err.ArgumentException("ttm3 needs a contiguous 3-way tensor")
        _chk2d(Ft, "ttm3 Ft")
        I, J, K = T.shape
        r = Ft.shape[0]
        shape = (r, J, K) if mode == 0 else ((I, r, K) if mode == 1 else (r, I, J))
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=T.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise EngineError("ttm3: `out` must be a contiguous float32 tensor of shape " + str(shape))
        _lib.check(self.lib.nnf_ttm3_f32(self.ctx, _ptr(T), I, J, K, _ptr(Ft), _ld(Ft), r, int(mode), _ptr(out),
                                         self._stream()), "nnf_ttm3_f32")
        return out

    def ntd_core_pg(self, core, MtX, grams, sparse, delta, max_iter, norm_sq, status=None):
        """Projected-gradient core update (ntd.py:588-619), in place on `core`; returns the 6-double status block
        {iterations, last update, first update, step, reconstruction error, 0} (device tensor, no sync)."""
        if core.dim() != 3 or not core.is_contiguous() or core.dtype != torch.float32:
            raise err.ArgumentException("ntd_core_pg updates a contiguous float32 3-way core in place")
        d0, d1, d2 = core.shape
        st = status if status is not None else torch.empty(6, dtype=torch.float64, device=core.device)
        MtX = MtX.contiguous()
        M = [g.contiguous() for g in grams]
        _lib.check(self.lib.nnf_ntd_core_pg_f32(self.ctx, _ptr(core), _ptr(MtX), _ptr(M[0]), _ptr(M[1]), _ptr(M[2]), d0, d1,
                                                d2, float(sparse), float(delta), int(max_iter), float(norm_sq), _ptr(st),
                                                self._stream()), "nnf_ntd_core_pg_f32")
        return st

    def betadiv(self, X, Ut, V, beta, out=None):
        _chk2d(X, "betadiv X"), _chk2d(Ut, "betadiv Ut"), _chk2d(V, "betadiv V")
        m, n = X.shape
        r = Ut.shape[0]
        o = out if out is not None else torch.empty(1, dtype=torch.float64, device=X.device)
        self._model_scratch(m, n, r)
        _lib.check(self.lib.nnf_betadiv_f32(self.ctx, _ptr(X), m, n, _ld(X), _ptr(Ut), _ld(Ut), _ptr(V),
                                            _ld(V), r, float(beta), _ptr(o), self._stream()), "nnf_betadiv_f32")
        return o

    def mttkrp3(self, T, Ft, mode, out=None):
        """T (I x J x K, contiguous), Ft = [F0^T, F1^T, F2^T] (each R x dim) -> R x dim_mode."""
        if T.dim() != 3 or T.dtype != torch.float32 or not T.is_contiguous():
            raise EngineError("mttkrp3: T must be a contiguous 3-way float32 tensor")
        for f in Ft:
            _chk2d(f, "mttkrp factor")
        I, J, K = T.shape
        R = Ft[0].shape[0]
        O = out if out is not None else torch.empty((R, T.shape[mode]), dtype=torch.float32, device=T.device)
        _lib.check(self.lib.nnf_mttkrp3_f32(self.ctx, _ptr(T), I, J, K, _ptr(Ft[0]), _ld(Ft[0]), _ptr(Ft[1]),
                                            _ld(Ft[1]), _ptr(Ft[2]), _ld(Ft[2]), R, int(mode), _ptr(O),
                                            _ld(O), self._stream()), "nnf_mttkrp3_f32")
        return O


    CP3_FUSED_MAX_RANK = 64

    def cp3_partial_cost(self, T, Ft, Y, cost):
        """One pass over T: cost[0] = ||T - [[F0,F1,F2]]||^2 (float64 device scalar) and Y[r][i][j] = sum_k T[i,j,k] F2[k,r]."""
        if T.dim() != 3 or T.dtype != torch.float32 or not T.is_contiguous():
            raise EngineError("cp3_partial_cost: T must be a contiguous 3-way float32 tensor")
        I, J, K = T.shape
        R = Ft[0].shape[0]
        if tuple(Y.shape) != (R, I, J) or Y.dtype != torch.float32 or not Y.is_contiguous():
            raise EngineError("cp3_partial_cost: Y must be a contiguous float32 R x I x J tensor")
        _lib.check(self.lib.nnf_cp3_partial_cost_f32(self.ctx, _ptr(T), I, J, K, _ptr(Ft[0]), _ld(Ft[0]), _ptr(Ft[1]),
                                                     _ld(Ft[1]), _ptr(Ft[2]), _ld(Ft[2]), R, _ptr(Y), _ptr(cost),
                                                     self._stream()), "nnf_cp3_partial_cost_f32")
        return Y

    def cp3_betadiv(self, T, Ft, beta, out=None):
        """beta-divergence between T (I x J x K) and the CP model of Ft = [F0^T, F1^T, F2^T]; float64 device scalar."""
        if T.dim() != 3 or T.dtype != torch.float32 or not T.is_contiguous():
            raise EngineError("cp3_betadiv: T must be a contiguous 3-way float32 tensor")
        I, J, K = T.shape
        R = Ft[0].shape[0]
        o = out if out is not None else torch.empty(1, dtype=torch.float64, device=T.device)
        self._model_scratch(I * J, K, R)
        _lib.check(self.lib.nnf_cp3_betadiv_f32(self.ctx, _ptr(T), I, J, K, _ptr(Ft[0]), _ld(Ft[0]), _ptr(Ft[1]),
                                                _ld(Ft[1]), _ptr(Ft[2]), _ld(Ft[2]), R, float(beta), _ptr(o),
                                                self._stream()), "nnf_cp3_betadiv_f32")
        return o


class Comm:
    """RCCL communicator of the C ABI (nnf_comm_*): what a C-ABI consumer uses for the row-sharded exchanges.  The Python
    drivers go through torch.distributed (the same RCCL underneath); this wrapper exists for the ABI tests and for hosts
    that do not initialise torch.distributed."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().nnf_comm_unique_id(buf), "nnf_comm_unique_id")
        return buf.raw

    def __init__(self, engine, nranks, rank, unique_id):
        self.engine = engine
        h = C.c_void_p()
        _lib.check(engine.lib.nnf_comm_create(C.byref(h), engine.ctx, int(nranks), int(rank),
                                              C.create_string_buffer(unique_id, 128)), "nnf_comm_create")
        self.h = h

    def size(self):
        return self.engine.lib.nnf_comm_size(self.h)

    def rank(self):
        return self.engine.lib.nnf_comm_rank(self.h)

    def allreduce_(self, t):
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.float64):
            raise EngineError("Comm.allreduce_: contiguous float32 / float64 device tensor expected")
        fn = self.engine.lib.nnf_allreduce_f32 if t.dtype == torch.float32 else self.engine.lib.nnf_allreduce_f64
        _lib.check(fn(self.h, _ptr(t), t.numel(), self.engine._stream()), "nnf_allreduce")
        return t

    def close(self):
        if getattr(self, "h", None):
            self.engine.lib.nnf_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_engine(device=None):
    """Process-wide engine for `device` (default: current device)."""
    if not torch.cuda.is_available():
        raise EngineError("no ROCm device available: the nn_fac_amd engine is GPU-only (no CPU fallback)")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    if dev.index is None:
        dev = torch.device(f"cuda:{torch.cuda.current_device()}")
    with _lock:
        e = _engines.get(dev.index)
        if e is None:
            # 1 GiB of scratch for the main context (NNF_WORKSPACE_MB overrides): the split-K slabs of W^T X at 10^6 x 4000 rank
            # 100 are 780 MB when a workgroup sums at most 2048 rows in fp32 (k_stream.hip: launch_xty) -- with the C default of
            # 256 MiB the plan falls back to longer chains.  Side contexts keep the default.
            mb = int(os.environ.get("NNF_WORKSPACE_MB", "1024"))
            e = Engine(dev, workspace_bytes=mb << 20)
            _engines[dev.index] = e
        return e


_side = {}


def get_side_engine(device, role="gram"):
    """Process-wide extra context + stream of `device` per role, for work that overlaps with a launch on the main stream (a
    context's workspace serves one stream at a time).  Created once: context creation allocates and clears its workspace."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        e = _side.get((idx, role))
        if e is None:
            e = (Engine(torch.device(f"cuda:{idx}"), workspace_bytes=64 << 20), torch.cuda.Stream(device=idx))
            _side[(idx, role)] = e
        return e
