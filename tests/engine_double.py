"""Engine test double: the method names / in-place semantics of nn_fac_amd.engine.Engine on fp64 CPU tensors, backed by the
CPU oracle.  Used where the HIP engine cannot run (no GPU in the build container): the gloo world-size-2 tests of the
row-sharded step and the multi-rank launch test of bench.py (NNF_BENCH_ENGINE=engine_double:OracleEngine).
Test infrastructure only -- never imported by the product."""
import math
import os
import sys

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import nnfac_oracle as orc  # noqa: E402


class OracleEngine:
    """Engine double: same method names / in-place semantics as nn_fac_amd.engine.Engine, fp64 CPU tensors."""

    def gram(self, A, out=None):
        G = A @ A.T
        return G if out is None else out.copy_(G)

    def xht(self, X, V, out=None):
        O = V @ X.T
        return O if out is None else out.copy_(O)

    def xty(self, X, Ut, out=None):
        O = Ut @ X
        return O if out is None else out.copy_(O)

    def frob_resid(self, X, Ut, V, out=None):
        c = torch.sum((X - Ut.T @ V) ** 2).reshape(1)
        return c if out is None else out.copy_(c)

    def hals_sweeps(self, UtM, UtU, V, nsweeps, sparsity=None, normalize=False, nonzero=False, snapshots=None, snap_first=0,
                    sweeps_done=0, resid_in=None, resid_out=None):
        log = []
        cur = V.numpy().copy()
        for s in range(nsweeps):   # one sweep at a time so that every intermediate V can be snapshotted
            cur, *_ = orc.hals_nnls_acc(UtM.numpy(), UtU.numpy(), cur, maxiter=1, alpha=math.inf, delta=0.0,
                                        sparsity_coefficient=sparsity, sweep_log=log)
            if snapshots is not None and s >= snap_first:
                snapshots[s - snap_first].copy_(torch.from_numpy(cur))
        V.copy_(torch.from_numpy(cur))
        return torch.tensor(log, dtype=torch.float64)

    def hals_solve(self, UtM, UtU, V, max_sweeps, delta=0.01, sparsity=None, normalize=False, nonzero=False,
                   status=None):
        Vn, eps, cnt, _ = orc.hals_nnls_acc(UtM.numpy(), UtU.numpy(), V.numpy(), maxiter=max_sweeps, alpha=math.inf,
                                            delta=delta, sparsity_coefficient=sparsity, normalize=normalize)
        V.copy_(torch.from_numpy(Vn))
        st = status if status is not None else torch.zeros(8, dtype=torch.float64)
        st[0], st[1], st[3] = float(eps), float(cnt), 0.0
        return st


    def mu_left(self, X, Ut, V, beta, out=None):
        O = torch.from_numpy(orc.mu_betadivmin(Ut.numpy().T, V.numpy(), X.numpy(), beta).T.copy())
        return O if out is None else out.copy_(O)

    def mu_right_accum(self, X, Ut, V, beta):
        U, K = Ut.T, Ut.T @ V
        if beta == 1:
            return U.T @ (X / K), None, U.sum(dim=0).double()
        if beta == 2:
            return U.T @ X, (U.T @ U) @ V, None
        return U.T @ (K ** (beta - 2) * X), U.T @ K ** (beta - 1), None

    def mu_apply(self, F, num, den, den_vec, beta, out=None):
        d = den if den is not None else den_vec.reshape(-1, 1)
        O = torch.clamp(F * (num / d) ** orc.gamma_beta(beta), min=1e-12)
        return O if out is None else out.copy_(O)

    def betadiv(self, X, Ut, V, beta, out=None):
        c = torch.tensor([orc.beta_divergence(X.numpy(), (Ut.T @ V).numpy(), beta)], dtype=torch.float64)
        return c if out is None else out.copy_(c)

    # ---- NTF (3-way, factors transposed R x dim like the HIP engine) ----
    def dot(self, A, B):
        return torch.sum(A * B).reshape(1).double()

    def hadamard(self, A, B, out=None):
        O = A * B
        return O if out is None else out.copy_(O)

    def mttkrp3(self, T, Ft, mode, out=None):
        F = [f.numpy().T for f in Ft]
        O = torch.from_numpy((orc.unfold(T.numpy(), mode) @ orc.khatri_rao(F, skip_matrix=mode)).T.copy())
        return O if out is None else out.copy_(O)

    def cp3_betadiv(self, T, Ft, beta, out=None):
        F = [f.numpy().T for f in Ft]
        model = (F[0] @ orc.khatri_rao(F, skip_matrix=0).T).reshape(T.shape)
        c = torch.tensor([orc.beta_divergence(T.numpy(), model, beta)], dtype=torch.float64)
        return c if out is None else out.copy_(c)

    def ttm3(self, T, Ft, mode, out=None):
        t = T.numpy()
        O = torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.tensordot(Ft.numpy(), t, axes=(1, mode)), 0, 0 if mode != 1 else 1)))
        return O if out is None else out.copy_(O)

    def mttkrp3_from_partial(self, Y, Ft, axis, out=None):
        y, f = Y.numpy(), Ft.numpy()
        O = torch.from_numpy(np.einsum('rab,ra->rb', y, f) if axis == 1 else np.einsum('rab,rb->ra', y, f))
        return O if out is None else out.copy_(O)

    CP3_FUSED_MAX_RANK = 64

    def cp3_partial_cost(self, T, Ft, Y, cost):
        F = [f.numpy().T for f in Ft]
        model = (F[0] @ orc.khatri_rao(F, skip_matrix=0).T).reshape(T.shape)
        cost.copy_(torch.tensor([np.sum((T.numpy() - model) ** 2)], dtype=torch.float64))
        Y.copy_(torch.from_numpy(np.einsum('ijk,kr->rij', T.numpy(), F[2])))
        return Y

    def hals_row_update(self, UtM, UtU, V, k, sparsity=None, out=None):
        sp = 0.0 if sparsity is None else sparsity
        d = float(UtU[k, k])
        nd = 0.0
        if d != 0:
            step = torch.maximum((UtM[k] - UtU[k] @ V - sp) / d, -V[k])
            V[k] += step
            nd = float(step @ step)
        o = torch.tensor([nd, float(V[k] @ V[k])], dtype=torch.float64)
        return o if out is None else out.copy_(o)

    def hals_row_scale(self, V, k, normsq, ncols_total):
        nsq = float(normsq[0])
        if nsq != 0:
            V[k] /= math.sqrt(nsq)
        else:
            V[k] = 1.0 / math.sqrt(ncols_total)
        return V

    def hals_stop_restore(self, sums, head, budget, delta, V, snapshots, status):
        s = sums.tolist()
        stop = next((j for j, v in enumerate(s) if not (v >= delta * s[0]) or j + 1 >= budget), None)
        status[2] = s[0]
        if stop is None:
            status[0], status[1], status[3] = s[-1], float(len(s) + 1), 4.0
        else:
            status[0], status[1], status[3] = s[stop], float(stop + 2), (3.0 if stop < head else 0.0)
            if head <= stop < len(s) - 1:
                V.copy_(snapshots[stop - head])
        return status
