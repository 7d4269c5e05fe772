"""Row-sharded NMF over torch.distributed (RCCL over xGMI on the GPU box; gloo in the CPU tests).

Partition: X (m x n) and U are split by rows across the ranks (contiguous blocks), V (r x n), both
Grams and every scalar are replicated.  Per outer iteration the only exchanges are

    U update : the stopping scalar sum(step^2) of the HALS sweeps over the local columns of U^T
    V update : all-reduce of UtU (r x r) and UtM (r x n); every rank then runs the identical r x n solve redundantly
               (identical inputs + deterministic kernels -> bitwise identical V, no broadcast)
    cost     : one f64 scalar

The U-side stopping rule (nnls.py:156) couples all shards once per sweep.  Exchanging one scalar per sweep would put
a collective and a host round trip (tens of microseconds) behind every ~10 us sweep, so the sweeps are run in chunks:
a chunk of C sweeps is executed blind (nnf_hals_sweeps_f32 records the local sum of every sweep and a snapshot of the
factor after every sweep), ONE all-reduce of the C partials follows, and the first sweep at which the reference would
have stopped is located.  If that is not the last sweep of the chunk, the factor is taken from that sweep's snapshot
(bitwise what a straight run would hold; the kernels are deterministic).  The chunk length follows the sweep count of
the previous outer iteration, which changes slowly, so a solve is normally one chunk and its stopping sweep falls in the
chunk's last few sweeps.  Only those (`SweepGuess.window`, 8) are run with snapshots -- writing the factor out costs
1.65 us per sweep next to an 8.6 us sweep at config B (tools/snap_probe.py) --; the sweeps before them run as a plain
launch, with one copy of the factor taken at the start of the chunk.  A stop before the window (rare: the count dropped by
more than the window since the previous outer iteration) restores that copy and re-runs exactly the right number of
sweeps -- the kernels are deterministic, so the result is bitwise what a straight run would hold.
"""
import torch
import torch.distributed as dist


import os

# NNF_FORCE_SHARDED=1: a ONE-rank group runs the row-sharded protocol too (its chunked solves, collectives and host decisions,
# with nobody to exchange with) -- rehearsal and measurement of what the protocol costs a rank on a one-GPU box
FORCE_SHARDED = os.environ.get("NNF_FORCE_SHARDED") == "1"


def world(group):
    return dist.get_world_size(group) if group is not None else 1


def is_sharded(group):
    return group is not None and (dist.get_world_size(group) > 1 or FORCE_SHARDED)


def opt_in(name, group):
    """Switch of the two row-sharded optimisations that need ONE PROCESS PER DEVICE to pay off (device-side stopping decision,
    NNF_SHARDED_ASYNC; cost under the V-side solve, NNF_SHARDED_OVERLAP): "1" in the environment turns one on, anything else
    (or unset) leaves it OFF.  Both are validated for correctness (gloo world-size-2 tests, a one-rank RCCL group on the GPU)
    and measured on one rank (-2.5 % instead of -8 % for the protocol at config B), but no run with more than
    one rank on real GPUs has exercised them yet -- the overlap path issues collectives from a second stream next to a
    resident persistent kernel -- so the default is the host-synchronous protocol until such a run exists.  bench.py records
    which mode ran and the hit / miss counts of the device-side decision in its JSON line."""
    return os.environ.get(name) == "1"


def allreduce_(t, group):
    if is_sharded(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max_(t, group):
    if is_sharded(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def allreduce_cost_(block, group):
    """The cost word of an iteration's 24-double status block ([16]) summed over the row blocks -- and, in the same
    collective, copies of the error words of the iteration's two solves ([3], [11] -> [17], [18]).  A solve's error word is
    identical on every rank except for one code: 1, "the persistent kernel gave up waiting for its workgroups", which is a
    rank-local event.  The sums let every rank decode the same code (`agreed_code`) and take the same branch of
    nmf.run_steps -- a rank falling back to chunked solves alone would stop matching its peers' collectives."""
    if is_sharded(group):
        block[17:19].copy_(block[3:12:8])
        dist.all_reduce(block[16:19], op=dist.ReduceOp.SUM, group=group)
    return block


def allreduce_errs_(block, group):
    """The error words alone (HALS cost through the Gram identity: every operand of the cost is replicated, nothing else of
    the block has to cross ranks)."""
    if is_sharded(group):
        block[17:19].copy_(block[3:12:8])
        dist.all_reduce(block[17:19], op=dist.ReduceOp.SUM, group=group)
    return block


def agreed_code(host, i, nranks):
    """Error code of solve i from the summed copies made by allreduce_cost_: 0 if no rank reported anything; the common code
    when every rank reported the same one (3 / 4: the device-side stopping decision missed -- a function of all-reduced sums;
    2: a zero row of the replicated factor); otherwise some ranks timed out and others did not: 1 for everybody."""
    total = int(round(float(host[17 + i])))
    if total == 0:
        return 0
    return total // nranks if total % nranks == 0 else 1


def agree_int(value, group, device=None):
    """Rank 0's `value` on every rank (a broadcast): quantities derived from the wall clock -- the sweep budget of
    nnls.py:156,190-194 -- differ from rank to rank, and the replicated solves must take the same decisions."""
    if group is None or dist.get_world_size(group) == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0, group=group)
    return int(t.item())


def shard_rows(m, rank, nranks):
    """Contiguous row block [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, extra = divmod(m, nranks)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def column_blocks(eng, F):
    """[(lo, hi)] column blocks of the r x m factor F, each small enough for the register-resident sweep kernel (the only one
    that writes snapshots, and the fast one: Engine.hals_resident_columns): one block for everything the bench's B / C / D
    shapes need, 8 for the 10^6-column factor of config E on one device, 4 / 2 on two / four.  Block edges are multiples of
    256 columns (whole workgroups, 1 KB-aligned rows)."""
    m = int(F.shape[1])
    cap = getattr(eng, "hals_resident_columns", None)
    cap = int(cap(F.shape[0])) if cap is not None else 0
    if cap <= 0 or m <= cap:
        return [(0, m)]
    nb = -(-m // cap)
    size = -(-m // nb)
    size = -(-size // 256) * 256
    return [(lo, min(lo + size, m)) for lo in range(0, m, size)]


class SweepGuess:
    """Per-factor state of the sharded solve: chunk length memory and the reusable snapshot buffers (one per column block)."""

    def __init__(self, first=16, max_chunk=104, window=8):
        self.value = first
        self.max_chunk = max_chunk
        self.window = window
        self.snap = None
        self.state = None

    def snapshots(self, F, W, blocks=None):
        """One (>= W) x r x (block columns) contiguous buffer per column block (a single-block F: one buffer shaped like F)."""
        blocks = blocks or [(0, int(F.shape[1]))]
        shapes = [(int(F.shape[0]), hi - lo) for lo, hi in blocks]
        ok = self.snap is not None and len(self.snap) == len(shapes) and all(
            s.shape[0] >= W and tuple(s.shape[1:]) == sh and s.device == F.device and s.dtype == F.dtype
            for s, sh in zip(self.snap, shapes))
        if not ok:
            self.snap = None          # release before allocating: eight 400 MB windows at config E's full size
            self.snap = [torch.empty((max(W, min(self.max_chunk, self.window)),) + sh, dtype=F.dtype, device=F.device)
                         for sh in shapes]
        return self.snap


    def residual_state(self, eng, F, blocks=None):
        """Two residual-state buffers per column block (Engine.hals_resid_floats elements each; None when the sweep kernel that
        runs keeps no state): a chunk reads one and leaves the other, so the state at the START of a chunk survives it (the
        re-run of an overshoot that ended before the snapshot window starts from it)."""
        blocks = blocks or [(0, int(F.shape[1]))]
        fn = getattr(eng, "hals_resid_floats", None)
        sizes = [int(fn(int(F.shape[0]), hi - lo)) if fn is not None else 0 for lo, hi in blocks]
        if not any(sizes):
            self.state = None
            return None
        ok = self.state is not None and len(self.state) == len(sizes) and all(
            a.numel() >= n and a.device == F.device for (a, b), n in zip(self.state, sizes))
        if not ok:
            self.state = None
            self.state = [(torch.empty(max(n, 1), dtype=torch.float32, device=F.device),
                           torch.empty(max(n, 1), dtype=torch.float32, device=F.device)) for n in sizes]
        return self.state


def _blind_sweeps(eng, cross, gram, F, blocks, nsweeps, sparsity, snaps=None, snap_first=0, done=0, state=None, flip=0):
    """`nsweeps` blind sweeps over every column block (ONE launch per block: `snap_first` sweeps without and the rest with a
    snapshot); the blocks' per-sweep sums of squared steps added in block order.  done: sweeps of this solve run before;
    state / flip: the residual-state pairs of SweepGuess.residual_state -- read from state[b][flip] (when done > 0), left in
    state[b][1 - flip]."""
    nd = None
    for bi, (lo, hi) in enumerate(blocks):
        whole = lo == 0 and hi == F.shape[1]
        kw = {}
        if snaps is not None:
            kw["snapshots"], kw["snap_first"] = snaps[bi], snap_first
        if state is not None:
            kw["sweeps_done"], kw["resid_out"] = done, state[bi][1 - flip]
            if done > 0:
                kw["resid_in"] = state[bi][flip]
        part = eng.hals_sweeps(cross if whole else cross[:, lo:hi], gram, F if whole else F[:, lo:hi], nsweeps,
                               sparsity=sparsity, **kw)
        nd = part if nd is None else nd + part
    return nd


def sharded_hals_solve(eng, cross, gram, F, group, guess, budget=100, delta=0.01, sparsity=None):
    """In-place HALS on the local columns F (r x m_local) with the GLOBAL stopping rule.  Returns (eps, cnt, eps0) of
    the reference (cnt = sweeps + 1).  normalize / nonzero need row-level reductions across shards: not supported.

    Chunks of C sweeps run blind; the kernel records every sweep's local sum of squared steps and, for the last
    `guess.window` sweeps of the chunk, a snapshot of F after the sweep.  ONE all-reduce per chunk locates the sweep at which
    the reference stops; if that is before the end of the chunk, F is restored from that sweep's snapshot, or -- before the
    window -- from the copy taken at the start of the chunk followed by a re-run of exactly that many sweeps (both bitwise
    what a straight run would hold)."""
    done, eps0, eps = 0, 0.0, 1.0
    if budget < 1:
        return 1.0, 1, 0.0
    blocks = column_blocks(eng, F)
    state = guess.residual_state(eng, F, blocks)
    flip = 0                                       # state[b][flip]: the residual state after `done` sweeps
    C = max(1, min(int(guess.value), guess.max_chunk, budget))
    while done < budget:
        C = max(1, min(C, budget - done))
        W = max(1, min(C, int(guess.window)))
        head = C - W
        F0 = F.clone() if head > 0 else None
        snap = guess.snapshots(F, W, blocks)
        nd = _blind_sweeps(eng, cross, gram, F, blocks, C, sparsity, snap, head, done, state, flip)
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
            flip = 1 - flip
            C = min(guess.max_chunk, 2 * C)        # keep going with longer chunks
            continue
        if stop < C - 1:                           # overshoot
            if stop >= head:                       # inside the window: the snapshot after sweep stop+1
                for (lo, hi), s in zip(blocks, snap):
                    F[:, lo:hi].copy_(s[stop - head])
            else:                                  # before it: start of the chunk + stop+1 sweeps again, from the chunk's start state
                F.copy_(F0)
                _blind_sweeps(eng, cross, gram, F, blocks, stop + 1, sparsity, None, 0, done, state, flip)
        done += stop + 1
        break
    guess.value = max(8, min(done + 4, guess.max_chunk))
    return eps, done + 1, eps0


def sharded_hals_solve_rownorm(eng, cross, gram, F, group, budget=100, delta=0.01, sparsity=None, normalize=True, nonzero=False):
    """hals_nnls_acc(..., normalize=True and / or nonzero=True) on the local columns F (r x m_local) of a factor whose columns
    are spread over the ranks -- or are more than the generic kernel keeps resident on one device: the row norm of
    nnls.py:179-185 and the all-zero test / max(V) of nnls.py:172-177 run over ALL columns, once per row update, so the rows
    are walked from the host -- row update on the local columns (Engine.hals_row_update), ONE all-reduce of {sum of squared
    steps, sum of squares of the row}, then the non-zero guard (a row left all zero is refilled with 1e-16 max(V), the maximum
    all-reduced; a zero Gram diagonal raises ZeroColumnWhenUnautorized, nnls.py:176-177) and the scaling
    (Engine.hals_row_scale) -- and the stopping rule of nnls.py:156 is applied once per sweep.  r collectives per sweep:
    correct and slow; the options are on no BASELINE configuration.  Returns (eps, cnt, eps0) like sharded_hals_solve."""
    from .utils import errors as err
    r = int(F.shape[0])
    ncols_total = torch.tensor([int(F.shape[1])], dtype=torch.int64, device=F.device)
    if group is not None and dist.get_world_size(group) > 1:
        dist.all_reduce(ncols_total, op=dist.ReduceOp.SUM, group=group)
    ncols_total = int(ncols_total.item())
    eps0, eps, done = 0.0, 1.0, 0
    acc = torch.zeros(1, dtype=torch.float64, device=F.device)
    diag = torch.diagonal(gram)[:r].cpu() if nonzero else None      # (replicated: the same on every rank)
    while done < budget:
        acc.zero_()
        for k in range(r):
            st = eng.hals_row_update(cross, gram, F, k, sparsity=sparsity)
            allreduce_(st, group)
            if nonzero:
                if float(diag[k]) == 0.0:
                    raise err.ZeroColumnWhenUnautorized("Column " + str(k) + " of U is zero with nonzero condition")
                if float(st[1].item()) == 0.0:                      # the updated row is all zero on every rank (nnls.py:173-174)
                    vmax = F.max().reshape(1).double()
                    allreduce_max_(vmax, group)
                    fill = float(1e-16 * vmax.item())
                    F[k].fill_(fill)
                    st[1] = fill * fill * ncols_total
            if normalize:
                eng.hals_row_scale(F, k, st[1:2], ncols_total)
            acc.add_(st[0:1])
        eps = float(acc.item())                    # one host round trip per sweep (nnls.py:187-196)
        if done == 0:
            eps0 = eps
        done += 1
        if not (eps >= delta * eps0):
            break
    return eps, done + 1, eps0


ERR_BEFORE_WINDOW, ERR_NOT_STOPPED = 3, 4      # status codes of nnf_hals_stop_restore_f32


def sharded_hals_solve_async(eng, cross, gram, F, group, guess, status, budget=100, delta=0.01, sparsity=None):
    """The same solve with NO host round trip: one blind chunk of C = guess.value sweeps (snapshots for the last
    guess.window of them), one all-reduce of the C per-sweep sums, and the stopping rule replayed on the device
    (Engine.hals_stop_restore), which restores F from the right snapshot and fills `status` (8 float64 on the device) with
    {eps, cnt, eps0, err}.  err = ERR_BEFORE_WINDOW / ERR_NOT_STOPPED mean the guess was off: the caller -- who reads the
    status block an iteration or two later -- redoes that iteration with `sharded_hals_solve` (host-synchronous, exact),
    which also re-centres the guess.  With sweep counts that drift slowly (and saturate at the budget, where the stop is the
    last sweep by construction) that is rare; a wrong guess costs a redo, never a wrong factor."""
    blocks = column_blocks(eng, F)
    C = max(1, min(int(guess.value), guess.max_chunk, budget))
    W = max(1, min(C, int(guess.window)))
    head = C - W
    snap = guess.snapshots(F, W, blocks)
    nd = _blind_sweeps(eng, cross, gram, F, blocks, C, sparsity, snap, head)      # head blind sweeps + the window: one launch
    allreduce_(nd, group)
    for (lo, hi), s in zip(blocks, snap):          # the same decision for every block (it is a function of `nd` alone)
        whole = lo == 0 and hi == F.shape[1]
        eng.hals_stop_restore(nd, head, budget, delta, F if whole else F[:, lo:hi], s, status)


def sharded_random_init(m, n, rank, group, seed=0, device=None, exact_stream=False):
    """Random start values of a row-sharded factorisation (initialize_factors.py:40-46 per shard).

    Returns (U0_block, V0, (lo, hi)): this rank's rows [lo, hi) of U_0 (m x rank) and the whole V_0 (rank x n), V_0 identical
    on every rank.  Default: drawn on the device -- V_0 from `seed`, the row block from `seed * nranks + 1 + rank` -- so that
    neither factor ever exists on the host (config E: U_0 alone is 800 MB in float64).  `exact_stream=True` reproduces the
    reference's legacy NumPy stream instead (np.random.seed(seed); rand(m, rank); rand(rank, n)): every rank skips to its
    rows in that stream, so the concatenated blocks ARE the unsharded reference start (host memory: one block + V_0)."""
    import numpy as np
    nranks, r_ = world(group), (dist.get_rank(group) if group is not None else 0)
    lo, hi = shard_rows(m, r_, nranks)
    if exact_stream:
        rng = np.random.RandomState(seed)
        done = 0
        while done < lo * rank:                      # skip the rows of the ranks before this one, a bounded piece at a time
            k = min(lo * rank - done, 1 << 24)
            rng.random_sample(k)
            done += k
        U0 = rng.random_sample((hi - lo, rank))
        rest = (m - hi) * rank
        while rest > 0:
            k = min(rest, 1 << 24)
            rng.random_sample(k)
            rest -= k
        V0 = rng.random_sample((rank, n))
        U0, V0 = torch.from_numpy(U0), torch.from_numpy(V0)
        if device is not None:
            U0, V0 = U0.to(device=device, dtype=torch.float32), V0.to(device=device, dtype=torch.float32)
        return U0, V0, (lo, hi)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    gv = torch.Generator(device=dev).manual_seed(int(seed))
    gu = torch.Generator(device=dev).manual_seed(int(seed) * nranks + 1 + r_)
    V0 = torch.rand(rank, n, device=dev, generator=gv)
    U0 = torch.rand(hi - lo, rank, device=dev, generator=gu)
    return U0, V0, (lo, hi)
