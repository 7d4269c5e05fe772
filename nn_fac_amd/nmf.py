"""NMF driver on the MI355X engine -- drop-in for nn_fac/nmf.py (nmf :19-193, compute_nmf :196-329, one_nmf_step :332-458).

The outer alternating loop, its stopping test and the call signatures are the reference's; what changes is where the
arithmetic runs: data and factors live on the device for the whole run (factors transposed: Ut r x m, V r x n) and
each statement of one_nmf_step maps to one call of libnnfac_hip.so:

    VVt = V V^T, VMt = V X^T      (nmf.py:407-408)  -> nnf_gram_f32, nnf_xht_f32
    hals_nnls_acc(VMt, VVt, U^T)  (nmf.py:415/418)  -> nnf_hals_solve_f32   (one persistent launch)
    UtU = U^T U, UtM = U^T X      (nmf.py:432-433)  -> nnf_gram_f32, nnf_xty_f32
    hals_nnls_acc(UtM, UtU, V)    (nmf.py:440/443)  -> nnf_hals_solve_f32
    ||X - U V||_F^2               (nmf.py:452)      -> nnf_frob_resid_f32   (product never materialised)
    mu / beta-divergence          (nmf.py:422,447,455) -> nnf_mu_left_f32, nnf_mu_right_f32, nnf_betadiv_f32

Inputs may be NumPy arrays or torch tensors; results come back in kind.  Arithmetic is fp32 on the device
(fp64 for every long reduction), whatever the input dtype.  Row-sharded multi-GPU runs go through
``nn_fac_amd.dist`` (same step, RCCL all-reduce of the Gram / cross terms).
"""
import math
import os
import time
import warnings

import numpy as np
import torch

from .utils import errors as err
from .utils import initialize_factors as init_factors
from . import engine as _engine
from ._convert import device_of, to_dev, to_dev_t, like_input
from . import dist as _dist


def nmf(data, rank, init="random", U_0=None, V_0=None, n_iter_max=100, tol=1e-8,
        update_rule="hals", beta=2,
        sparsity_coefficients=[None, None], fixed_modes=[], normalize=[False, False],
        verbose=False, return_costs=False, deterministic=False, seed=0):
    """Nonnegative matrix factorisation  data ~= U V  (reference docstring: nmf.py:23-174)."""
    if min(data.shape) < rank:
        min_data = min(data.shape)
        rank = min_data
        warnings.warn(f"The rank is too high for the input matrix. It was set to {min_data} instead.")

    if deterministic:
        np.random.seed(seed)

    if init.lower() == "custom":
        if U_0 is None or V_0 is None:
            raise err.CustomNotValidFactors("Custom initialization, but (at least) one factor is set to 'None'")
    else:
        U_0, V_0 = init_factors.nmf_initialization(data, rank, init, deterministic=deterministic, seed=seed)

    return compute_nmf(data, rank, U_0, V_0, n_iter_max=n_iter_max, tol=tol,
                       update_rule=update_rule, beta=beta,
                       sparsity_coefficients=sparsity_coefficients, fixed_modes=fixed_modes, normalize=normalize,
                       verbose=verbose, return_costs=return_costs, deterministic=deterministic)


def compute_nmf(data, rank, U_in, V_in, n_iter_max=100, tol=1e-8,
                update_rule="hals", beta=2,
                sparsity_coefficients=[None, None], fixed_modes=[], normalize=[False, False],
                verbose=False, return_costs=False, deterministic=False, sweep_log=None, group=None):
    """Outer loop of nmf.py:284-329.  ``sweep_log`` (extension): list receiving the inner sweep counts.
    ``group`` (extension): a torch.distributed process group -- `data` and `U_in` are then THIS RANK'S row block of a
    row-sharded problem (contiguous blocks, nn_fac_amd.dist.shard_rows), `V_in` is replicated; the Gram / cross terms, the
    stopping scalars and the cost are all-reduced over the group (RCCL over xGMI) and every rank returns its
    block of U, the whole V and the global costs (start values: nn_fac_amd.dist.sharded_random_init)."""
    dev = device_of(data, U_in, V_in)
    eng = _engine.get_engine(dev)
    X = to_dev(data, dev)
    Ut = to_dev_t(U_in, dev).clone()
    V = to_dev(V_in, dev).clone()
    cost_fct_vals = []
    tic = time.time()
    toc = []

    if sparsity_coefficients is None:
        sparsity_coefficients = [None, None]
    if fixed_modes is None:
        fixed_modes = []
    if normalize is None or normalize is False:
        normalize = [False, False]

    ws = _StepBuffers(X, Ut.shape[0])

    def retired(iteration, cost, sweeps):
        """Host side of one finished iteration (nmf.py:315-324); True = the stopping test fired."""
        if sweep_log is not None:
            sweep_log.extend(sweeps)
        toc.append(time.time() - tic)
        cost_fct_vals.append(cost)

        if verbose:
            if iteration == 0:
                print('Normalized cost function value={}'.format(cost))
            else:
                if cost_fct_vals[-2] - cost_fct_vals[-1] > 0:
                    print('Normalized cost function value={}, variation={}.'.format(
                        cost_fct_vals[-1], cost_fct_vals[-2] - cost_fct_vals[-1]))
                else:
                    print('\033[91m' + 'Normalized cost function value={}, variation={}.'.format(
                        cost_fct_vals[-1], cost_fct_vals[-2] - cost_fct_vals[-1]) + '\033[0m')

        if iteration > 0 and abs(cost_fct_vals[-2] - cost_fct_vals[-1]) < tol:
            if verbose:
                print('Converged in {} iterations.'.format(iteration))
            return True
        return False

    def revise_last(cost):
        # the loop switched from the Gram-identity cost to the streaming kernel (run_steps): the last value is re-evaluated the
        # same way, so that the variation printed next -- and the stopping test -- compare two costs of one kind
        if verbose:
            print('(cost evaluation switched to the pass over the data; last value {} re-evaluated: {})'.format(
                cost_fct_vals[-1], cost))
        cost_fct_vals[-1] = cost
    retired.revise_last = revise_last

    Ut, V = run_steps(eng, ws, X, rank, Ut, V, n_iter_max, update_rule, beta, sparsity_coefficients, fixed_modes,
                      normalize, deterministic, retired, group=group, tol=tol)

    U_out, V_out = like_input(Ut.t(), U_in), like_input(V, V_in)
    if return_costs:
        return U_out, V_out, cost_fct_vals, toc
    return U_out, V_out


def one_nmf_step(data, rank, U_in, V_in, norm_data, update_rule, beta,
                 sparsity_coefficients, fixed_modes, normalize, deterministic):
    """One pass of updates on U then V, then the cost (nmf.py:387-458).  Returns (U, V, cost)."""
    dev = device_of(data, U_in, V_in)
    eng = _engine.get_engine(dev)
    X = to_dev(data, dev)
    Ut, V = to_dev_t(U_in, dev), to_dev(V_in, dev)
    ws = _StepBuffers(X, Ut.shape[0])
    Ut2, V2, nstat = _one_nmf_step_dev(eng, ws, X, rank, Ut, V, update_rule, beta, sparsity_coefficients,
                                       fixed_modes, normalize, deterministic)
    host = ws.block.cpu()
    cost = float(host[16])
    _raise_on_status(host, nstat)
    return like_input(Ut2.t(), U_in), like_input(V2, V_in), cost


# ------------------------------------------------------------------------------------------------------------
PIPELINE_DEPTH = 1   # outer iterations enqueued ahead of the one whose cost the host is looking at


class _StepBuffers:
    """Device scratch reused across iterations (cross terms, Grams, status words)."""

    def select(self, slot):
        self.slot = slot
        self.block = self.blocks[slot]
        self.cost = self.block[16:17]

    def __init__(self, X, r, dtype=torch.float32):
        m, n = X.shape
        f32 = dict(dtype=dtype, device=X.device)     # (float64 only in the CPU tests' engine double)
        self.VMt = torch.empty((r, m), **f32)
        # UtM (r x n) and UtU (r x r) share one allocation: row-sharded runs sum both with ONE all-reduce
        self.v_terms = torch.empty(r * n + r * r, **f32)
        self.UtM = self.v_terms[:r * n].view(r, n)
        self.G2 = self.v_terms[r * n:].view(r, r)
        self.G = torch.empty((r, r), **f32)
        # U^T U before its rounding to fp32 (nnf_gram_f64_f32), for the Gram-identity cost (row-sharded: summed over the ranks by
        # a collective of its own, r x r doubles)
        self.G64 = torch.empty((r, r), dtype=torch.float64, device=X.device) if X.is_cuda else None
        self.g64_ok = False
        # one block read back per iteration: HALS status of the first / second solve at [0:8] / [8:16], cost at [16].
        # A ring of PIPELINE_DEPTH + 1 blocks with pinned host mirrors: run_steps enqueues iteration i+1 before it reads
        # the block of iteration i, so the device never waits for the host between iterations.
        self.blocks = torch.zeros((PIPELINE_DEPTH + 2, 24), dtype=torch.float64, device=X.device)
        self.host = torch.zeros((PIPELINE_DEPTH + 2, 24), dtype=torch.float64)
        if X.is_cuda:
            self.host = self.host.pin_memory()
        self.select(0)
        self.guess_u = _dist.SweepGuess()
        self.guess_v = _dist.SweepGuess()
        # Row-sharded U-side solve with the device-side stopping decision (dist.sharded_hals_solve_async): on by default over
        # RCCL, NNF_SHARDED_ASYNC=0/1 forces it (dist.opt_in).  A missed guess costs a pipeline drain + a redone iteration, and the sweep counts of the first
        # outer iterations jump by tens (33, 52, 67, 38, ... at config B), so it is engaged only once two consecutive solves
        # differ by <= 4 sweeps (`async_ready`).  Validated for correctness (gloo world-size-2 tests, one-GPU kernel test);
        # its gain needs one process per GPU to show -- with two ranks time-slicing ONE GPU (the only rehearsal available
        # here) the unsynchronised ranks starve each other's persistent V-side solves -- hence off over gloo.
        self.async_sharded = None         # decided by run_steps from the group (dist.opt_in)
        self.async_ready = False
        self.last_u_count = None
        self.sync_next = False            # row-sharded: the next step uses the host-synchronous U-side protocol (after a redo)
        self.last_step_async = False
        self.async_hits = self.async_misses = 0
        self.safe_solve = False           # set by run_steps after a persistent solve timed out (chunked launches from then on)
        self.direct_cost = False          # set by run_steps when the Gram-identity cost said it cannot carry the residual
        self.normx2 = None                # ||X||^2 (float64 device scalar; row-sharded: summed over the ranks), on first use
        self.cross_rounding = None        # (rms, |mean|) of the relative rounding of a U^T X entry at this shape, on first use
        # the r x r Gram of an update is independent of its cross product (nmf.py:407-408, :432-433): it runs on a side
        # stream, with its own context (a context's workspace serves one stream at a time), under the streaming kernel
        self.side_eng, self.side_stream = _engine.get_side_engine(X.device) if X.is_cuda else (None, None)
        # a third context + stream: the cost of one iteration runs under the V-side solve of the next (run_steps)
        self.cost_eng, self.cost_stream = _engine.get_side_engine(X.device, "cost") if X.is_cuda else (None, None)



def _gram_on_side(ws, eng, A, out):
    """eng.gram(A, out=out) overlapped with whatever the caller launches next on the current stream (the r x r Gram of an
    update is independent of its cross product, nmf.py:407-408 / :432-433).  Returns the event the current stream has to wait
    on before `out` is used -- None: it ran inline (CPU test doubles, buffers without a side stream).  Only the SHORT factor's
    Gram (V V^T: r x n) goes to the side stream: the Gram of the long one (U^T U, 20 MB at config B) took 82 us there next to
    W^T X instead of 5.4 us alone and cost that kernel 5 % (profiles/r02_B_kernel_stats.txt) -- it is launched in line, in
    front of W^T X (`inline=True`)."""
    side = getattr(ws, "side_stream", None)
    if side is None or not isinstance(eng, _engine.Engine):
        eng.gram(A, out=out)
        return None
    main = torch.cuda.current_stream(A.device)
    ready = main.record_event()
    with torch.cuda.stream(side):
        side.wait_event(ready)
        ws.side_eng.gram(A, out=out)
        return side.record_event()


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


class _SolveTimedOut(Exception):
    """A persistent HALS solve gave up waiting for its other workgroups (status word 1)."""


class _IdentityUnreliable(Exception):
    """HALS cost through the Gram identity (nnf_nmf_gram_cost_f32): the kernel's own error estimate is above 5e-4 of the
    cost -- the residual is too small next to ||X||^2 for fp32 cross terms (an almost exact fit).  The iteration is redone
    with the streaming cost kernel, and so is the rest of the run."""


class _IdentityNearStop(_IdentityUnreliable):
    """Two consecutive identity costs differ by the caller's `tol` give or take their error estimates."""


class _GuessMissed(Exception):
    """Row-sharded run: the blind chunk of the device-side protocol did not contain the stopping sweep in its snapshot window
    (status words 3 / 4 of nnf_hals_stop_restore_f32): the iteration is redone with the host-synchronous protocol."""


def _raise_on_status(host, nstat, timeout_ok=False, nranks=0):
    """`nranks` > 0: a row-sharded run -- the error words are read from the copies that travelled with the cost's all-reduce
    (dist.allreduce_cost_), so every rank sees the same code for the same iteration and takes the same branch: a time-out
    is a rank-local event (the replicated V-side solve of ONE rank found the chip shared), and a rank that fell back to
    chunked solves alone would issue a different sequence of collectives than its peers (a hang over RCCL)."""
    for i in range(nstat):
        code = _dist.agreed_code(host, i, nranks) if nranks else int(host[8 * i + _engine.ST_ERR])
        if code == 2:
            raise err.ZeroColumnWhenUnautorized("A column of U is zero with nonzero condition")
        if code in (_dist.ERR_BEFORE_WINDOW, _dist.ERR_NOT_STOPPED):
            raise _GuessMissed()
        if code != 0:
            if timeout_ok:
                raise _SolveTimedOut()
            raise err.EngineError("hals grid barrier timed out; result invalid")


def run_steps(eng, ws, X, rank, Ut, V, n_iter, update_rule, beta, sparsity_coefficients, fixed_modes, normalize,
              deterministic, retired, group=None, tol=None):
    """The `for iteration` loop of compute_nmf (nmf.py:298-324) with the device running ahead of the host.

    * Iteration i+1 is enqueued BEFORE the host reads the cost of iteration i (its 24-double block arrives through an
      asynchronous copy into pinned memory + an event), so the per-iteration round trip -- D2H copy, stopping test in
      Python, ~15 launches -- no longer leaves the GPU idle.
    * HALS: the cost of iteration i (nmf.py:452: one MFMA/HBM-bound pass over X, ~290 us at B) is not launched at the end
      of iteration i but next to the V-side solve of iteration i+1, on its own stream and context: that solve is a
      persistent kernel of ceil(n/16) single-wave workgroups (125 at n = 2000, ~270 us) which leaves the chip all but
      empty, and everything else of iteration i+1 depends on it.  The cost kernel's waves are small enough (<= 136 VGPRs)
      for a sweep wave to fit on a SIMD they fill, so the order in which the two get their CUs does not matter.  The main
      stream waits for that cost before it launches the next U-side solve (whose 1563 waves should find the chip free).
      The host then looks at costs two iterations behind the device.  Row-sharded runs do the same over RCCL (one process
      per device by construction; NNF_SHARDED_OVERLAP=0/1 forces it, dist.opt_in) and keep the cost inside the step over
      gloo: two processes sharing ONE GPU -- the only multi-rank rehearsal a one-GPU box offers -- run two persistent sweep
      kernels (one per process) plus the extra cost kernels, which can each end up partially resident and wait for the
      other's workgroups until the bounded spins expire (0.5 s per step).  Measured on ONE rank running the whole sharded
      protocol over a 1-rank RCCL group (NNF_FORCE_SHARDED=1, 40 steady-state iterations at config B): 590 iterations/s
      unsharded, 543 host-synchronous, 552 with the device-side decision, 575 with both.

    `retired(iteration, cost, sweeps)` is called once per iteration, in order, and returns True when the loop has to stop
    (nmf.py:320-324); the factors returned are those of the iteration that stopped it -- each step writes fresh factor
    tensors, so the speculative iterations in flight are simply dropped.  Paths that synchronise inside a step anyway
    (wall-clock rule, row-sharded solve) run through the same code."""
    cuda = X.is_cuda
    # HALS cost (nmf.py:452) WITHOUT a pass over X: ||X||^2 - 2 <V, U^T X> + sum_j v_j^T (U^T U) v_j from the operands of the V
    # update (they belong to the iteration's final U) and its result, inner products in fp64 (Engine.gram_cost).  Measured
    # against the fp64 residual on config B's own iterates: 7e-8 ... 1.2e-5 relative (the streaming kernel: 1e-9), absolute
    # error ~1e-9 ||X||^2 -- tools/probes/gram_cost_probe.py.  The kernel estimates its error from the operands and flags
    # an iterate it cannot carry to 5e-4 (an almost exact fit): run_steps then redoes that iteration, and runs the rest, with
    # the streaming kernel.  Row-sharded: every operand is replicated (||X||^2 summed once), so the cost needs no collective.
    # NNF_COST=direct in the environment forces the streaming kernel.
    # `tol` given (the caller stops on |cost[i-1] - cost[i]| < tol, nmf.py:320): an iterate whose difference to its predecessor
    # is within the two error estimates of `tol` is treated the same way.  At either switch the predecessor's cost is re-evaluated
    # by the streaming kernel too and handed to `retired.revise_last` -- the stopping test never compares costs of two kinds.
    ident = (cuda and update_rule == "hals" and 1 not in fixed_modes and isinstance(eng, _engine.Engine)
             and not ws.direct_cost and os.environ.get("NNF_COST") != "direct")
    if ident and ws.normx2 is None:
        ws.normx2 = eng.dot(X, X)
        _dist.allreduce_(ws.normx2, group)
    if ident and ws.cross_rounding is None:
        # what the cross-product kernel's fp32 accumulation leaves in U^T X at this shape (it grows with the rows a workgroup
        # sums: 5.7e-8 rms at config B, 9.5e-7 rms with a -2.2e-7 mean at 1e6 x 4000 rank 100): measured once per run, never
        # below the figures the estimate was calibrated with; row-sharded: the largest over the ranks, so that every rank's
        # (replicated) cost kernel flags the same iterates
        sa, ba = eng.cross_rounding(X, Ut)
        # (and what the Gram kernel's fp32 accumulation inside a split leaves in the fp64 copy of U^T U the cost is taken on)
        sg = eng.gram_rounding(Ut) if ws.G64 is not None else 0.0
        cal = torch.tensor([max(1.5 * sa, 6e-8), 1.5 * ba, max(1.5 * sg, 2e-9)], dtype=torch.float64, device=X.device)
        _dist.allreduce_max_(cal, group)
        ws.cross_rounding = tuple(float(v) for v in cal.cpu())
    overlap = (cuda and update_rule == "hals" and 1 not in fixed_modes and isinstance(eng, _engine.Engine) and not ident
               and ws.cost_stream is not None
               and (not _dist.is_sharded(group) or _dist.opt_in("NNF_SHARDED_OVERLAP", group)))
    if ws.async_sharded is None:
        ws.async_sharded = _dist.is_sharded(group) and _dist.opt_in("NNF_SHARDED_ASYNC", group)
    # MU, beta = 1: the left update of iteration i+1 forms U_i V_i entry by entry -- the KL cost of iteration i rides along
    # (nnf_mu_left_kl_cost_f32) and the separate cost pass over X (a quarter of a KL iteration at config C) is only run after
    # the last iteration.  Like the overlapped HALS cost, the host then looks at costs two iterations behind the device.
    # (row-sharded: the kernel's sum covers this rank's rows; one scalar all-reduce before the block goes to the host)
    fused_mu = (cuda and update_rule == "mu" and float(beta) == 1.0 and 0 not in fixed_modes
                and isinstance(eng, _engine.Engine) and Ut.shape[0] <= eng.MU_FUSED_MAX_RANK)
    depth = PIPELINE_DEPTH + (1 if (overlap or fused_mu) else 0)
    assert ws.blocks.shape[0] > depth
    pending = []          # steps not yet handed to `retired`: dicts {it, slot, Ut, V, nstat, ev}
    result = (Ut, V)
    stop = False
    main = torch.cuda.current_stream(X.device) if cuda else None

    def cost_of(step, stream):
        """Launch the cost of `step` (+ the copy of its status block to the host) on `stream`."""
        block = ws.blocks[step["slot"]]
        with torch.cuda.stream(stream):
            if stream is not main:
                stream.wait_event(main.record_event())       # factors, status words of `step`: all enqueued on main
            if fused_mu:
                # the last step's cost from the SAME kernel as every other cost of the run (an update whose output is
                # dropped): a run stopped early and a run of exactly that many iterations give bitwise equal costs
                eng.mu_left(X, step["Ut"], step["V"], beta, cost_out=block[16:17])
                _dist.allreduce_cost_(block, group)
            else:
                _step_cost(ws.cost_eng if stream is not main else eng, X, step["Ut"], step["V"], update_rule, beta,
                           sparsity_coefficients, block, group)
            ws.host[step["slot"]].copy_(block, non_blocking=True)
            step["ev"] = stream.record_event()

    last = None           # (cost, error estimate) of the last retired iterate while both came from the identity

    def retire():
        nonlocal result, stop, last
        step = pending[0]
        if step["ev"] is not None:
            step["ev"].synchronize()
        host = ws.host[step["slot"]]
        _raise_on_status(host, step["nstat"], timeout_ok=not getattr(ws, "safe_solve", False),
                         nranks=_dist.world(group) if _dist.is_sharded(group) else 0)
        if step.get("ident"):
            if float(host[20]) != 0.0:
                raise _IdentityUnreliable()
            c, e = float(host[19]), float(host[21])
            if tol is not None and tol > 0 and last is not None and abs(last[0] - c) < tol + e + last[1]:
                raise _IdentityNearStop()
            last = (c, e)
        pending.pop(0)                     # (a step that timed out stays at the head: run_steps resumes from it)
        result = (step["Ut"], step["V"])
        if group is not None and update_rule == "hals" and step["nstat"] >= 1 and 0 not in fixed_modes:
            cnt_u = int(host[_engine.ST_CNT]) - 1
            ws.async_ready = ws.last_u_count is not None and abs(cnt_u - ws.last_u_count) <= 4
            ws.last_u_count = cnt_u
        if step.get("async_u"):            # row-sharded, device-side protocol: centre the next blind chunk on this count
            ws.async_hits += 1
            ws.guess_u.value = max(8, min(int(host[_engine.ST_CNT]) - 1 + 4, ws.guess_u.max_chunk))
        stop = bool(retired(step["it"], float(host[19 if step.get("ident") else 16]),
                            [int(host[8 * i + _engine.ST_CNT]) - 1 for i in range(step["nstat"])]))

    def drain():
        if cuda:
            main.synchronize()
            if ws.cost_stream is not None:
                ws.cost_stream.synchronize()

    def fall_back():
        """A persistent solve timed out: its workgroups were not all resident at once (another process's kernels hold CUs --
        on a GPU this process owns alone that does not happen).  Everything in flight is dropped, and the loop resumes from
        the last retired factors with every HALS solve going through the chunked fixed-count launches of dist.py -- no
        workgroup of those waits for another, the stopping rule is applied between chunks (bitwise the same factors,
        tests/test_dist_gloo.py) -- and the cost inside the step.  Slower (one host round trip per chunk), never wrong."""
        nonlocal overlap, depth, owed, costed, stop
        drain()
        pending.clear()
        ws.safe_solve = True
        overlap, depth, owed, costed = False, PIPELINE_DEPTH, None, None
        warnings.warn("nn_fac_amd: a persistent HALS solve timed out waiting for its workgroups (GPU shared with another "
                      "process?); falling back to chunked launches for the rest of this run")

    owed = None           # overlap: the step whose cost has not been launched yet
    costed = None         # overlap: the step whose cost was launched during the previous step
    iteration = 0
    while iteration < n_iter:
        ws.select(iteration % ws.blocks.shape[0])
        hooks = {}
        if fused_mu and owed is not None:
            hooks["mu_cost_out"] = ws.blocks[owed["slot"]][16:17]        # cost of the previous step, by-product of this left update
        if overlap and owed is not None:
            hooks["before_v_solve"] = lambda prev=owed: cost_of(prev, ws.cost_stream)
        if overlap and costed is not None:
            # the cost launched during the previous step must be out of the way before this step's U-side solve
            hooks["before_u_solve"] = lambda ev=costed["ev"]: main.wait_event(ev)
        Ut, V, nstat = _one_nmf_step_dev(eng, ws, X, rank, Ut, V, update_rule, beta, sparsity_coefficients,
                                         fixed_modes, normalize, deterministic, group=group,
                                         skip_cost=overlap or fused_mu or ident, **hooks)
        step = dict(it=iteration, slot=ws.slot, Ut=Ut, V=V, nstat=nstat, ev=None, async_u=ws.last_step_async, ident=ident)
        ws.sync_next = False
        if ident:
            # words 19..21 of the block: {cost, 1 = not reliable, error estimate}; the V update's operands are still in place
            eng.gram_cost(V, ws.UtM, ws.G2, ws.normx2, ws.block[19:22], rounding=ws.cross_rounding,
                          UtU64=ws.G64 if ws.g64_ok else None)
            _add_sparsity_terms(Ut, V, sparsity_coefficients, ws.block[19:20], group)
            _dist.allreduce_errs_(ws.block, group)
            ws.host[ws.slot].copy_(ws.block, non_blocking=True)
            step["ev"] = main.record_event()
        elif fused_mu:
            if owed is not None:          # its cost has just been enqueued with this step's left update
                _dist.allreduce_cost_(ws.blocks[owed["slot"]], group)
                ws.host[owed["slot"]].copy_(ws.blocks[owed["slot"]], non_blocking=True)
                owed["ev"] = main.record_event()
            owed = step
        elif overlap:
            costed = owed
            owed = step
        elif cuda:
            ws.host[ws.slot].copy_(ws.block, non_blocking=True)
            step["ev"] = main.record_event()
        else:
            ws.host[ws.slot].copy_(ws.block)
        pending.append(step)
        iteration += 1
        try:
            if len(pending) > depth:
                retire()
                if stop:
                    break
            if iteration == n_iter:
                if (overlap or fused_mu) and not stop and owed is not None and owed["ev"] is None:
                    cost_of(owed, main)               # the last step has no V-side solve behind it to hide under
                while pending and not stop:
                    retire()
        except _SolveTimedOut:
            failed = pending[0]["it"]             # the step being retired is still at the head of the list
            fall_back()
            Ut, V = result                        # factors of the last iteration that retired cleanly
            iteration = failed
        except _IdentityUnreliable:
            failed = pending[0]["it"]
            drain()
            pending.clear()
            if last is not None and hasattr(retired, "revise_last"):
                # whichever test failed: the iterate before it was costed by the identity -- re-evaluate it too, so that the
                # stopping test never compares a cost of one kind with a cost of the other
                scratch = torch.zeros_like(ws.block)
                _step_cost(eng, X, result[0], result[1], update_rule, beta, sparsity_coefficients, scratch, group)
                retired.revise_last(float(scratch[16]))
            last = None
            ws.direct_cost = True                 # this iteration again, and every later one, with the streaming cost kernel
            ident = False
            overlap = (ws.cost_stream is not None
                       and (not _dist.is_sharded(group) or _dist.opt_in("NNF_SHARDED_OVERLAP", group)))
            depth = PIPELINE_DEPTH + (1 if overlap else 0)
            owed = costed = None
            Ut, V = result
            iteration = failed
        except _GuessMissed:
            failed = pending[0]["it"]
            drain()
            pending.clear()
            owed = costed = None
            ws.sync_next = True                   # redo this iteration with the exact, host-synchronous protocol
            ws.async_misses += 1
            Ut, V = result
            iteration = failed
    if cuda and (pending or overlap):     # dropped speculative iterations still use the shared scratch: let them drain
        main.synchronize()
        if overlap:
            ws.cost_stream.synchronize()
    return result


# inner-solve settings of one_nmf_step (nmf.py:415-419,440-444: maxiter=100, delta=0.01).  bench.py's fixed-work line
# (delta=0, maxiter=10, so that runs are comparable whatever the data) overrides them through this dict.
HALS_INNER = {"maxiter": 100, "delta": 0.01}


def _timed_budget(eng, cross, gram, F, sparsity, normalize, timer, group=None):
    """Sweep budget of the wall-clock rule: rho = atime / btime with btime = the time of one sweep, measured on a scratch copy
    (nnls.py:155,190-194), cnt <= 1 + 0.5 rho (nnls.py:156).  Row-sharded runs take rank 0's figure on every rank."""
    from .update_rules.nnls import sweep_budget
    probe = F.clone()
    _sync(F.device)
    t0 = time.time()
    eng.hals_sweeps(cross, gram, probe, 1, sparsity=sparsity, normalize=normalize)
    _sync(F.device)
    btime = max(time.time() - t0, 10e-7)
    rho = timer / btime if timer else 100000
    budget = max(1, sweep_budget(HALS_INNER["maxiter"], 0.5, rho))
    return _dist.agree_int(budget, group, F.device)


def _hals_call(eng, cross, gram, F_in, sparsity, normalize, deterministic, timer, status, safe=None, group=None, guess=None):
    """hals_nnls_acc(..., maxiter=100, atime=timer, alpha=inf|0.5, delta=0.01) of nmf.py:415-419,440-444.  Returns the updated
    factor, a NEW tensor: the solve reads its start values from F_in and writes its result elsewhere (the V_in / V_out form of
    nnf_hals_solve_cross_f32 -- the reference works on `in_V.copy()`, nnls.py:147; the 20 MB copy of U^T that used to run on
    the side stream next to X H^T is gone).
    `safe` (a dist.SweepGuess): the solve runs as chunked fixed-count launches whose workgroups never wait for each other
    (run_steps' fall-back after a persistent solve timed out); row normalisation needs the persistent kernel and cannot.
    `guess` (a dist.SweepGuess): the same chunked form is ALSO the fast one for a factor with more columns than the resident
    sweep kernel holds (config E on one device: 10^6 columns, rank 100) -- blind chunks over register-resident column blocks
    instead of a kernel that streams the whole factor through HBM every sweep (935 -> ~350 us per sweep there)."""
    from .update_rules.nnls import sweep_budget
    budget = HALS_INNER["maxiter"]
    if safe is None and guess is not None and deterministic and not normalize:
        cap = getattr(eng, "hals_resident_columns", None)
        if cap is not None and F_in.shape[1] > cap(F_in.shape[0]):
            safe = guess
    if safe is not None and deterministic and not normalize:
        F = F_in.clone()
        eps, cnt, eps0 = _dist.sharded_hals_solve(eng, cross, gram, F, None, safe, budget=budget, delta=HALS_INNER["delta"],
                                                  sparsity=sparsity)
        status[:4] = torch.tensor([eps, cnt, eps0, 0.0], dtype=torch.float64)
        return F
    if not deterministic:
        budget = _timed_budget(eng, cross, gram, F_in, sparsity, normalize, timer, group)
    if hasattr(eng, "hals_solve_cross"):
        F = torch.empty_like(F_in)
        eng.hals_solve_cross(cross, gram, None, F_in, F, budget, delta=HALS_INNER["delta"], sparsity=sparsity, normalize=normalize,
                             status=status)
        return F
    F = F_in.clone()                       # (CPU test doubles: in place on a copy)
    eng.hals_solve(cross, gram, F, budget, delta=HALS_INNER["delta"], sparsity=sparsity, normalize=normalize, nonzero=False,
                   status=status)
    return F


def _step_cost_local(eng, X, Ut, V, update_rule, beta, out):
    """The tensor-sized part of the cost line (nmf.py:452 / :455) over this rank's rows, into the 1-element float64 `out`."""
    if update_rule == "hals":
        eng.frob_resid(X, Ut, V, out=out)                         # nmf.py:452
    else:
        eng.betadiv(X, Ut, V, beta, out=out)                      # nmf.py:455


def _step_cost_finish(Ut, V, update_rule, sparsity_coefficients, block, group=None):
    """Sum over the row blocks (the error words of the iteration's solves ride along: dist.allreduce_cost_) and the sparsity
    terms of nmf.py:452.  `block`: the iteration's 24-double status block, cost at [16]."""
    sharded = _dist.is_sharded(group)
    out = block[16:17]
    if sharded:
        _dist.allreduce_cost_(block, group)
    if update_rule == "hals":
        _add_sparsity_terms(Ut, V, sparsity_coefficients, out, group)


def _add_sparsity_terms(Ut, V, sparsity_coefficients, out, group=None):
    """out += 2 (sp0 ||U||_1 + sp1 ||V||_1) with the MATRIX 1-norm (max column abs-sum, np.linalg.norm(., ord=1)) -- not the
    entry-wise l1 the reference's docstring states (nmf.py:452)."""
    sp = [0 if s is None else s for s in sparsity_coefficients]
    if sp[0] or sp[1]:
        cs = Ut.abs().sum(dim=1).double()         # columns of U are rows of Ut
        if _dist.is_sharded(group):
            _dist.allreduce_(cs, group)
        nU = cs.max()
        nV = V.abs().sum(dim=0).max().double()
        out.add_(2 * (sp[0] * nU + sp[1] * nV))


def _step_cost(eng, X, Ut, V, update_rule, beta, sparsity_coefficients, block, group=None):
    """The cost line of one_nmf_step (nmf.py:449-455) into word 16 of the float64 status block `block`, on the current stream."""
    _step_cost_local(eng, X, Ut, V, update_rule, beta, block[16:17])
    _step_cost_finish(Ut, V, update_rule, sparsity_coefficients, block, group)


def _one_nmf_step_dev(eng, ws, X, rank, Ut_in, V_in, update_rule, beta, sparsity_coefficients, fixed_modes, normalize,
                      deterministic, group=None, skip_cost=False, before_u_solve=None, before_v_solve=None, mu_cost_out=None):
    """Device-resident step.  Ut_in (r x m) and V_in (r x n) are not modified.  Returns the new factors and the number
    of HALS solves run; the cost and the solves' status words are left in ws.block (read back by the caller).
    With `group` (torch.distributed process group) X / Ut are this rank's row block and V is replicated (dist.py).
    run_steps' hooks: `before_u_solve` / `before_v_solve` are called right before the U-side / V-side HALS solve is
    launched; with `skip_cost` the cost line is left to the caller (who overlaps it with the next V-side solve)."""
    sharded = _dist.is_sharded(group)
    if update_rule not in ["hals", "mu"]:
        raise err.InvalidArgumentValue(f"Invalid update rule: {update_rule}") from None
    if update_rule == "hals" and beta != 2:
        raise err.InvalidArgumentValue(f"The hals is only valid for the frobenius norm, corresponding to the beta divergence with beta = 2. Here, beta was set to {beta}. To compute NMF with this value of beta, please use the mu update_rule.") from None
    if len(sparsity_coefficients) != 2:
        raise ValueError("NMF needs 2 sparsity coefficients to be performed")

    Ut, V = Ut_in, V_in
    nstat = 0
    dev = X.device

    if 0 not in fixed_modes:
        if update_rule == "hals":
            timer = None
            if not deterministic:
                _sync(dev)
                t0 = time.time()
            done = _gram_on_side(ws, eng, V, ws.G)      # VVt (nmf.py:407)
            eng.xht(X, V, out=ws.VMt)                   # VMt  (nmf.py:408)
            if done is not None:
                torch.cuda.current_stream(dev).wait_event(done)
            if not deterministic:
                _sync(dev)
                timer = time.time() - t0
            if before_u_solve is not None:
                before_u_solve()
            ws.last_step_async = False
            budget_u = HALS_INNER["maxiter"]
            if sharded and not deterministic:    # wall-clock rule (nnls.py:190-194): rank 0's budget on every rank
                budget_u = _timed_budget(eng, ws.VMt, ws.G, Ut, sparsity_coefficients[0], False, timer, group)
            if sharded:
                Ut = Ut_in.clone()                  # the chunked sharded protocols work in place (nmf.py:415: from U_in^T)
            if sharded and normalize[0]:
                # the row norm runs over the columns of all ranks, once per row update: rows walked from the host (dist.py)
                eps, cnt, eps0 = _dist.sharded_hals_solve_rownorm(eng, ws.VMt, ws.G, Ut, group, budget=budget_u,
                                                                  delta=HALS_INNER["delta"], sparsity=sparsity_coefficients[0])
                ws.block[8 * nstat:8 * nstat + 4] = torch.tensor([eps, cnt, eps0, 0.0], dtype=torch.float64)
            elif sharded and deterministic and hasattr(eng, "hals_stop_restore") and not ws.sync_next and ws.async_sharded \
                    and ws.async_ready:
                # no host round trip: blind chunk + all-reduce + device-side replay of the stopping rule; a missed guess
                # shows in the status block and run_steps redoes the iteration through the branch below
                _dist.sharded_hals_solve_async(eng, ws.VMt, ws.G, Ut, group, ws.guess_u, ws.block[8 * nstat:8 * nstat + 8],
                                               budget=HALS_INNER["maxiter"], delta=HALS_INNER["delta"],
                                               sparsity=sparsity_coefficients[0])
                ws.last_step_async = True
            elif sharded:
                eps, cnt, eps0 = _dist.sharded_hals_solve(eng, ws.VMt, ws.G, Ut, group, ws.guess_u,
                                                          budget=budget_u, delta=HALS_INNER["delta"],
                                                          sparsity=sparsity_coefficients[0])
                ws.block[8 * nstat:8 * nstat + 4] = torch.tensor([eps, cnt, eps0, 0.0], dtype=torch.float64)
            else:
                Ut = _hals_call(eng, ws.VMt, ws.G, Ut_in, sparsity_coefficients[0], normalize[0], deterministic, timer,
                                ws.block[8 * nstat:8 * nstat + 8],
                                safe=ws.guess_u if getattr(ws, "safe_solve", False) else None, guess=ws.guess_u)
            nstat += 1
        else:
            if mu_cost_out is not None:                 # + beta_divergence(X, U_in V_in, 1): the previous iteration's cost
                Ut = eng.mu_left(X, Ut_in, V, beta, cost_out=mu_cost_out)
            else:
                Ut = eng.mu_left(X, Ut_in, V, beta)     # nmf.py:422

    if 1 not in fixed_modes:
        if update_rule == "hals":
            timer = None
            if not deterministic:
                _sync(dev)
                t0 = time.time()
            ws.g64_ok = getattr(ws, "G64", None) is not None and isinstance(eng, _engine.Engine)
            if ws.g64_ok:
                eng.gram(Ut, out=ws.G2, out64=ws.G64)   # UtU  (nmf.py:432) -- in line: see _gram_on_side; + its fp64 sums
            else:
                eng.gram(Ut, out=ws.G2)
            eng.xty(X, Ut, out=ws.UtM)                  # UtM  (nmf.py:433)
            if sharded:                                 # sum over the row blocks: r x n and r x r over xGMI, one collective
                if getattr(ws, "v_terms", None) is not None:
                    _dist.allreduce_(ws.v_terms, group)
                else:
                    _dist.allreduce_(ws.G2, group)
                    _dist.allreduce_(ws.UtM, group)
                if ws.g64_ok:                           # the fp64 sums of the Gram for the identity cost: r x r doubles
                    _dist.allreduce_(ws.G64, group)
            if not deterministic:
                _sync(dev)
                timer = time.time() - t0
            if before_v_solve is not None:
                before_v_solve()
            V = _hals_call(eng, ws.UtM, ws.G2, V_in, sparsity_coefficients[1], normalize[1], deterministic, timer,
                           ws.block[8 * nstat:8 * nstat + 8], safe=ws.guess_v if getattr(ws, "safe_solve", False) else None,
                           group=group if sharded else None, guess=ws.guess_v)
            nstat += 1
        else:
            if sharded:
                # the sums over the rows of X are additive over the row blocks: numerator and denominator
                # (beta = 1: the r column sums of U) are all-reduced, then every rank applies the same update
                num, den, dvec = eng.mu_right_accum(X, Ut, V_in, beta)
                _dist.allreduce_(num, group)
                _dist.allreduce_(den if den is not None else dvec, group)
                V = eng.mu_apply(V_in, num, den, dvec, beta)
            else:
                V = eng.mu_right(X, Ut, V_in, beta)     # nmf.py:447

    if not skip_cost:
        _step_cost(eng, X, Ut, V, update_rule, beta, sparsity_coefficients, ws.block, group)
    return Ut, V, nstat
