"""Row-sharded NMF over torch.distributed (RCCL over xGMI on the GPU box; gloo in the CPU tests).

Partition: X (m x n) and U are split by rows across the ranks (contiguous blocks), V (r x n), both
Grams and every scalar are replicated.  Per outer iteration the only exchanges are

    U update : the stopping scalar sum(step^2) of the HALS sweeps over the local columns of U^T
    V update : all-reduce of UtU (r x r) and UtM (r x n); every rank then runs the identical r x n solve redundantly
               (identical inputs + deterministic kernels -> bitwise identical V, no broadcast)
    cost     : one f64 scalar

The U-side stopping rule (nnls.py:156) couples all shards once per sweep.  Exchanging one scalar per sweep would put
a collective and a host round trip (tens of microseconds) behind every ~10 us sweep, so the sweeps are run in chunks:
a chunk of C sweeps is executed blind (nnf_hals_sweeps_f32 records the local sum of every sweep), ONE all-reduce of
the C partials follows, and the first sweep at which the reference would have stopped is located.  If that is the
last sweep of the chunk we are done; if it is earlier the chunk is replayed from a saved copy for exactly that many
sweeps (the kernels are deterministic, so the replay reproduces the straight run bit for bit).  The chunk length is
the sweep count of the previous outer iteration, which changes slowly.
"""
import torch
import torch.distributed as dist


def world(group):
    return dist.get_world_size(group) if group is not None else 1


def allreduce_(t, group):
    if group is not None and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_rows(m, rank, nranks):
    """Contiguous row block [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, extra = divmod(m, nranks)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class SweepGuess:
    """Chunk length memory for the sharded solve (one per factor)."""

    def __init__(self, first=8):
        self.value = first


def sharded_hals_solve(eng, cross, gram, F, group, guess, budget=100, delta=0.01, sparsity=None):
    """In-place HALS on the local columns F (r x m_local) with the GLOBAL stopping rule.  Returns (eps, cnt, eps0) of
    the reference (cnt = sweeps + 1).  normalize / nonzero need row-level reductions across shards: not supported."""
    done, eps0, eps = 0, 0.0, 1.0
    if budget < 1:
        return 1.0, 1, 0.0
    while done < budget:
        C = max(1, min(int(guess.value), budget - done))
        backup = F.clone()
        nd = eng.hals_sweeps(cross, gram, F, C, sparsity=sparsity)
        allreduce_(nd, group)
        ndh = nd.cpu().tolist()                    # one host round trip per chunk
        stop = None
        for j, v in enumerate(ndh):
            if done + j == 0:
                eps0 = v
            eps = v
            if not (v >= delta * eps0) or done + j + 1 >= budget:
                stop = j
                break
        if stop is None:
            done += C
            continue
        if stop < C - 1:                           # overshoot: replay exactly stop+1 sweeps
            F.copy_(backup)
            eng.hals_sweeps(cross, gram, F, stop + 1, sparsity=sparsity)
        done += stop + 1
        break
    guess.value = max(1, done)
    return eps, done + 1, eps0
