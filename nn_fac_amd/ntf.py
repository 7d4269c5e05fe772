"""NTF (nonnegative PARAFAC / CP) driver on the MI355X engine -- drop-in for nn_fac/ntf.py
(ntf :19-199, compute_ntf :201-344, one_ntf_step :347-477), tensors of any order >= 3.

Per updated mode the reference statements map to the C ABI as follows (factors kept transposed, R x dim):

    cross = hadamard of the other factors' Grams   (ntf.py:442-445) -> nnf_gram_f32 + nnf_hadamard_f32
    krao ; rhs = unfolded[mode] @ krao             (ntf.py:448-449) -> nnf_mttkrp3_f32 (one pass over the tensor IN PLACE:
                                                     neither the unfoldings nor the Khatri-Rao matrix are materialised)
    hals_nnls_acc(rhs^T, cross, F[mode]^T)         (ntf.py:454-456) -> nnf_hals_solve_f32
    mu_betadivmin(F[mode], krao^T, unfolded[mode]) (ntf.py:459-460) -> nnf_mu_right_f32 on the transposed unfolding (MU path only)
    cost                                           (ntf.py:462-475) -> HALS loops: the reference's own form
        ||T||^2 - 2<F,rhs> + sum_k f_k^T cross f_k on the last updated mode's operands, inner products in fp64, with the
        kernel's error estimate as a guard (nnf_nmf_gram_cost_f32: no pass over T); wherever that form cannot carry the
        cost -- an almost exact fit (the difference cancels), an iterate within the estimate of the stopping threshold,
        a single step, MU -- ||T - model||^2 / the beta-divergence directly: nnf_cp3_betadiv_f32, one pass over the tensor.

Differences kept on purpose: ``one_ntf_step`` exposes ``alpha`` like the reference (default 0.5 = wall-clock dependent,
ntf.py:349); ``compute_ntf`` adds ``alpha`` / ``delta`` keywords (default: the reference's) so that deterministic runs
(alpha = inf) are reachable through the driver.  Tensors of order > 3 run through the same 3-way kernels on VIEWS of the
tensor that group adjacent modes, against the Khatri-Rao product of each group's factors (_NtfState.view3); the dimension
tree and the fused cost pass are 3-way only.
"""
import math
import os
import time

import numpy as np
import torch

from .utils import errors as err
from .utils import initialize_factors as init_factors
from . import engine as _engine
from ._convert import device_of, to_dev, to_dev_t, like_input
from .update_rules.nnls import sweep_budget
from . import dist as _dist


def ntf(tensor, rank, init="random", factors_0=[], n_iter_max=100, tol=1e-8,
        update_rule="hals", beta=2,
        sparsity_coefficients=[], fixed_modes=[], normalize=[],
        verbose=False, return_costs=False):
    """Nonnegative PARAFAC of `tensor` (reference docstring: ntf.py:23-179)."""
    factors = []
    nb_modes = len(tensor.shape)
    if init.lower() == "custom":
        factors = factors_0
        if len(factors) != nb_modes:
            raise err.CustomNotEngouhFactors("Custom initialization, but not enough factors")
        else:
            for array in factors:
                if array is None:
                    raise err.CustomNotValidFactors("Custom initialization, but (at least) one factor is set to 'None'")
    else:
        factors = init_factors.ntf_initialization(tensor, rank, init, deterministic=False, seed=0)

    return compute_ntf(tensor, rank, factors, n_iter_max=n_iter_max, tol=tol,
                       update_rule=update_rule, beta=beta,
                       sparsity_coefficients=sparsity_coefficients, fixed_modes=fixed_modes, normalize=normalize,
                       verbose=verbose, return_costs=return_costs)


class _NtfState:
    """Device-resident tensor, its squared norm and (MU only) the materialised unfoldings.

    With `group` (torch.distributed process group) T is this rank's block of the LEADING mode and the mode-0
    factor is sharded the same way; the other factors, every Gram and every scalar are replicated: the mode-0 update is
    local up to the global stopping scalar of its sweeps (dist.sharded_hals_solve), the other modes all-reduce their
    MTTKRP output (R x I_k) together with the mode-0 Gram (R x R) -- one collective --, the cost one f64."""

    def __init__(self, eng, T, group=None):
        if T.dim() < 3:
            raise NotImplementedError("NTF needs a tensor of order >= 3 (a matrix is nmf's business)")
        if T.dim() > 3 and _dist.world(group) > 1:
            raise NotImplementedError("leading-mode-sharded NTF is built for 3-way tensors")
        self.eng = eng
        self.group = group
        self.T = T.contiguous()
        self.nway = self.T.dim()
        t2 = self.T.view(self.T.shape[0], -1)
        self.norm2 = eng.dot(t2, t2)          # float64 device scalar, ||T||^2
        if _dist.is_sharded(group):
            _dist.allreduce_(self.norm2, group)
        self.guess0 = _dist.SweepGuess()
        # leading-mode-sharded runs: the mode-0 solve with the device-side stopping decision (dist.sharded_hals_solve_async),
        # as in the NMF step -- engaged once two consecutive solves differ by <= 4 sweeps, a missed guess redoes the
        # iteration with the host-synchronous protocol (run_ntf_steps)
        self.async_sharded = _dist.is_sharded(group) and _dist.opt_in("NNF_SHARDED_ASYNC", group)
        self.async_ready = self.sync_next = self.last_step_async = False
        self.last_cnt0 = None
        self.async_hits = self.async_misses = 0
        self._unf = {}
        self.direct_cost = False           # set for the rest of a run once the Gram-identity cost was found unreliable
        self._Y, self._Y_of, self._grams = None, None, {}
        # per-iteration status: one HALS status block per mode, then {cost, 1 = identity cost not reliable, its error
        # estimate} at [8 * nway ...]; a ring with pinned host mirrors (run_ntf_steps)
        self.cost_at = 8 * self.nway
        self.blocks = torch.zeros((3, self.cost_at + 8), dtype=torch.float64, device=T.device)
        self.host = torch.zeros((3, self.cost_at + 8), dtype=torch.float64)
        if T.is_cuda:
            self.host = self.host.pin_memory()
        self.select(0)

    def select(self, slot):
        self.slot = slot
        self.block = self.blocks[slot]

    def view3(self, mode, Ft):
        """Order-N tensors through the 3-way kernels: mode `mode` of T (I_0 x ... x I_{N-1}, C order) is mode m3 of a 3-way
        VIEW of the same memory whose other two axes are groups of adjacent modes, against the Khatri-Rao product of each
        group's factors (first factor of a group slowest = tl.unfold / tl.tenalg.khatri_rao order):
            first mode : (I_0, I_1, rest)         [F_0, F_1, KR(F_2..)]           m3 = 0
            last mode  : (rest, I_{N-2}, I_{N-1}) [KR(F_0..F_{N-3}), F_{N-2}, F_{N-1}]   m3 = 2
            otherwise  : (left, I_n, right)       [KR(F_0..F_{n-1}), F_n, KR(F_{n+1}..)] m3 = 1
        Returns (T3, [three transposed factors], m3).  The grouped factors are R x prod(group dims): small next to T."""
        N, sh = self.nway, self.T.shape
        if N == 3:
            return self.T, list(Ft), mode
        if mode == 0:
            return self.T.view(sh[0], sh[1], -1), [Ft[0], Ft[1], _kr_group_t(Ft[2:])], 0
        if mode == N - 1:
            return self.T.view(-1, sh[N - 2], sh[N - 1]), [_kr_group_t(Ft[:N - 2]), Ft[N - 2], Ft[N - 1]], 2
        left = 1
        for s in sh[:mode]:
            left *= int(s)
        return self.T.view(left, sh[mode], -1), [_kr_group_t(Ft[:mode]), Ft[mode], _kr_group_t(Ft[mode + 1:])], 1

    def _ybuf(self, R):
        I, J, K = self.T.shape
        if self._Y is None or tuple(self._Y.shape) != (R, I, J):
            self._Y = torch.empty((R, I, J), dtype=self.T.dtype, device=self.T.device)
        return self._Y

    def partial(self, Ft2):
        """Y[r][i][j] = sum_k T[i][j][k] F2[k][r] (tl.tenalg.mode_dot(T, F2^T, 2) with the new axis first): one pass over T,
        shared by the mode-0 and mode-1 right-hand sides of an iteration (F2 does not change between them).  When the
        previous iteration's cost pass already produced it for this very factor (cost_and_partial), that one is returned."""
        if self._Y_of is Ft2:
            return self._Y
        self._Y_of = None
        return self.eng.ttm3(self.T, Ft2, 2, out=self._ybuf(Ft2.shape[0]))

    def cost_and_partial(self, Ft, cost):
        """||T - [[F0,F1,F2]]||^2 into `cost` AND the partial product of the next iteration, one pass over T."""
        self.eng.cp3_partial_cost(self.T, Ft, self._ybuf(Ft[2].shape[0]), cost)
        self._Y_of = Ft[2]               # (a strong reference: the identity test above cannot meet a recycled tensor)

    def gram_of(self, i, F):
        """F F^T (R x R) of factor i, computed once per value of the factor (each is used by two mode updates)."""
        hit = self._grams.get(i)
        if hit is not None and hit[0] is F:
            return hit[1]
        G = self.eng.gram(F)
        self._grams[i] = (F, G)
        return G

    def unfolded_t(self, mode):
        """tl.unfold(T, mode)^T = moveaxis(mode -> last).reshape(-1, dim), contiguous (MU path; the last mode is a view)."""
        if mode not in self._unf:
            self._unf[mode] = torch.movedim(self.T, mode, -1).reshape(-1, self.T.shape[mode]).contiguous()
        return self._unf[mode]


def _kr_group_t(Fts):
    """khatri_rao(group)^T for a list of transposed factors (R x dim each): R x prod(dims), first factor slowest."""
    res = Fts[0]
    for f in Fts[1:]:
        res = (res[:, :, None] * f[:, None, :]).reshape(res.shape[0], -1)
    return res.contiguous()


def _krao_t(Ft, skip):
    """khatri_rao(factors, skip_matrix=skip)^T as an R x prod(other dims) tensor, first remaining mode slowest."""
    others = [f for i, f in enumerate(Ft) if i != skip]
    res = others[0]
    for f in others[1:]:
        res = (res[:, :, None] * f[:, None, :]).reshape(res.shape[0], -1)
    return res.contiguous()


def _ntf_cost(eng, st, Ft, update_rule, beta, sparsity_coefficients, cost, fuse_next=False, host_norm=False, ident=None):
    """The cost lines of one_ntf_step (ntf.py:462-475) into the float64 device words `cost` ([0] = the cost), current stream.
    `fuse_next` (HALS, another iteration follows): the same pass over T leaves the next iteration's partial product.
    `ident` = (mode, rhs, Ga, Gb) of the last updated mode: the reference's own expression (ntf.py:462-470)
        ||T||^2 - 2 <F_mode, rhs> + sum_k f_k^T (Ga .* Gb) f_k        (f_k: the rows of the I_mode x R factor)
    with the three inner products in fp64 (Engine.gram_cost) instead of a pass over T; cost[1], cost[2] = the kernel's
    verdict on its own accuracy and its error estimate.  Leading-mode-sharded: every operand is replicated (rhs and the
    mode-0 Gram are all-reduced for the solve, ||T||^2 once per run) -- no collective."""
    sharded = _dist.is_sharded(st.group)
    if ident is not None:
        mode, rhs_t, Ga, Gb = ident
        # rounding of an MTTKRP entry, measured at 500^3 rank 30 and 300 x 200 x 1000 rank 64 (tools/probes/mttkrp_rounding_probe.py,
        # profiles/r04_mttkrp_rounding.txt): 3.5e-8 ... 5.1e-8 relative rms, |mean| <= 7.5e-10 -- its split-K chains are short
        # (a workgroup sums a few hundred slices); the estimate assumes 6e-8 and a mean of 1e-9
        eng.gram_cost(Ft[mode], rhs_t, Ga, st.norm2, cost[0:3], UtU_b=Gb, rounding=(6e-8, 1e-9))
    elif update_rule == "hals" and fuse_next and hasattr(eng, "cp3_partial_cost") \
            and Ft[0].shape[0] <= getattr(eng, "CP3_FUSED_MAX_RANK", 0):
        st.cost_and_partial(Ft, cost[0:1])           # ||T - model||^2
    elif update_rule == "hals":
        T3, F3, _ = st.view3(0, Ft)
        eng.cp3_betadiv(T3, F3, 2, out=cost[0:1])
        cost[0:1].mul_(2.0)                          # ||T - model||^2
    else:
        T3, F3, _ = st.view3(0, Ft)
        eng.cp3_betadiv(T3, F3, beta, out=cost[0:1])
    cost = cost[0:1]
    if sharded and ident is None:
        _dist.allreduce_(cost, st.group)             # additive over the blocks of the leading mode
    sparsity_error = None
    for index, sparse in enumerate(sparsity_coefficients):
        if sparse:
            # np.linalg.norm(factor, ord=1): max column abs-sum of the dim x R factor = max row abs-sum of Ft
            cs = Ft[index].abs().sum(dim=1).double()
            if sharded and index == 0:
                _dist.allreduce_(cs, st.group)
            term = 2 * sparse * cs.max()
            sparsity_error = term if sparsity_error is None else sparsity_error + term
    if sparsity_error is not None:
        cost.add_(sparsity_error)
    if not host_norm:     # (run_ntf_steps divides on the host when it reads the block: one 5 us launch less per iteration)
        cost.div_(st.norm2)


def _one_ntf_step_dev(st, rank, Ft_in, update_rule, beta, sparsity_coefficients, fixed_modes, normalize, alpha, delta,
                      skip_cost=False, fuse_next=False, host_norm=False, ident=False):
    eng = st.eng
    if update_rule not in ["hals", "mu"]:
        raise err.InvalidArgumentValue(f"Invalid update rule: {update_rule}") from None
    if update_rule == "hals" and beta != 2:
        raise err.InvalidArgumentValue(f"The hals is only valid for the frobenius norm, corresponding to the beta divergence with beta = 2. Here, beta was set to {beta}. To compute NMF with this value of beta, please use the mu update_rule.") from None
    for fixed_value in fixed_modes:
        sparsity_coefficients[fixed_value] = None
    Ft = list(Ft_in)
    dev = st.T.device
    nstat = 0
    sharded = _dist.is_sharded(st.group)
    if sharded and (update_rule != "hals" or not math.isinf(alpha) or normalize[0]):
        raise NotImplementedError("leading-mode-sharded NTF: HALS with alpha = inf and no normalisation of mode 0")
    # Dimension tree: with modes 0 and 1 both updated, their right-hand sides (ntf.py:448-449) are two contractions of the
    # same partial product Y = T x_2 F2^T -- one pass over T instead of two.  Only where the result does not depend on the
    # wall clock (alpha = inf): the timed rule prices every mode's own Gram + MTTKRP (ntf.py:440-451).
    Y = None
    N = st.nway
    if N == 3 and update_rule == "hals" and math.isinf(alpha) and 0 not in fixed_modes and 1 not in fixed_modes \
            and hasattr(eng, "mttkrp3_from_partial"):
        Y = st.partial(Ft[2])

    def rhs_of(mode, out=None):
        if Y is not None and mode == 0:
            return eng.mttkrp3_from_partial(Y, Ft[1], 2, out=out)
        if Y is not None and mode == 1:
            return eng.mttkrp3_from_partial(Y, Ft[0], 1, out=out)      # Ft[0]: already this iteration's update
        T3, F3, m3 = st.view3(mode, Ft)
        return eng.mttkrp3(T3, F3, m3, out=out)

    last = None            # (mode, rhs, Ga, Gb) of the last updated mode: what the reference's cost line is made of (ntf.py:462-470)
    for mode in [m for m in range(N) if m not in fixed_modes]:
        if update_rule == "hals":
            deterministic = math.isinf(alpha)
            if not deterministic:
                torch.cuda.synchronize(dev)
                t0 = time.time()
            Ga = Gb = None              # the two Grams whose Hadamard product is `cross` (ntf.py:442-445)
            if sharded and mode != 0:
                # MTTKRP output (R x I_mode) and the Gram of the sharded mode-0 factor (R x R) are sums over the blocks of
                # the leading mode (ntf.py:442-449): they share an allocation and ONE all-reduce
                R_, dim = Ft[mode].shape
                buf = torch.empty(R_ * dim + R_ * R_, dtype=Ft[mode].dtype, device=dev)
                rhs_t, g0 = buf[:R_ * dim].view(R_, dim), buf[R_ * dim:].view(R_, R_)
                eng.gram(Ft[0], out=g0)
                rhs_of(mode, out=rhs_t)
                _dist.allreduce_(buf, st.group)
                Ga, Gb = g0, st.gram_of(3 - mode, Ft[3 - mode])
            else:
                grams = [st.gram_of(i, f) for i, f in enumerate(Ft) if i != mode]
                Ga = grams[0]
                for g in grams[1:-1]:                   # order > 3: all but the last factor of the product folded here
                    Ga = eng.hadamard(Ga, g)
                Gb = grams[-1] if len(grams) > 1 else None
                rhs_t = rhs_of(mode)
            budget = 100
            fused = deterministic and hasattr(eng, "hals_solve_cross") and not (sharded and mode == 0)
            if fused:
                # Hadamard product formed while the solve stages its Gram, start values read from the current factor, the
                # result written to a NEW tensor: no Hadamard launch, no copy in front of the solve
                new = torch.empty_like(Ft[mode])
                eng.hals_solve_cross(rhs_t, Ga, Gb, Ft[mode], new, budget, delta=delta, sparsity=sparsity_coefficients[mode],
                                     normalize=normalize[mode], status=st.block[8 * nstat:8 * nstat + 8])
                nstat += 1
                Ft[mode] = new
                last = (mode, rhs_t, Ga, Gb)
                continue
            cross = Ga if Gb is None else eng.hadamard(Ga, Gb)
            last = (mode, rhs_t, cross, None)
            new = Ft[mode].clone()
            if sharded and mode == 0:
                if st.async_sharded and st.async_ready and not st.sync_next and hasattr(eng, "hals_stop_restore"):
                    _dist.sharded_hals_solve_async(eng, rhs_t, cross, new, st.group, st.guess0,
                                                   st.block[8 * nstat:8 * nstat + 8], budget=budget, delta=delta,
                                                   sparsity=sparsity_coefficients[mode])
                    st.last_step_async = True
                else:
                    eps, cnt, eps0 = _dist.sharded_hals_solve(eng, rhs_t, cross, new, st.group, st.guess0, budget=budget,
                                                              delta=delta, sparsity=sparsity_coefficients[mode])
                    st.block[8 * nstat:8 * nstat + 4] = torch.tensor([eps, cnt, eps0, 0.0], dtype=torch.float64)
                nstat += 1
                Ft[mode] = new
                continue
            if not deterministic:
                torch.cuda.synchronize(dev)
                timer = time.time() - t0
                probe = new.clone()
                t0 = time.time()
                eng.hals_sweeps(rhs_t, cross, probe, 1, sparsity=sparsity_coefficients[mode], normalize=normalize[mode])
                torch.cuda.synchronize(dev)
                rho = timer / max(time.time() - t0, 10e-7) if timer else 100000
                budget = max(1, sweep_budget(100, alpha, rho))
            eng.hals_solve(rhs_t, cross, new, budget, delta=delta, sparsity=sparsity_coefficients[mode],
                           normalize=normalize[mode], status=st.block[8 * nstat:8 * nstat + 8])
            nstat += 1
            Ft[mode] = new
        else:
            # mu_betadivmin(F, krao^T, unfold) (ntf.py:459-460) on the transposed problem unfold^T ~ krao F^T: the unfolding
            # has only I_mode rows, its transpose gives the streaming kernel prod(other dims) rows to split over
            Ft[mode] = eng.mu_right(st.unfolded_t(mode), _krao_t(Ft, mode), Ft[mode], beta)

    if not skip_cost:
        _ntf_cost(eng, st, Ft, update_rule, beta, sparsity_coefficients, st.block[st.cost_at:st.cost_at + 3],
                  fuse_next=fuse_next and Y is not None, host_norm=host_norm, ident=last if ident else None)
    return Ft, nstat


def _identity_cost_applies(st, update_rule, fixed_modes):
    """HALS loops on the device take their cost from the last updated mode's operands (_ntf_cost) -- unless that mode is the
    sharded one (its factor and right-hand side are blocks, not replicas), an earlier iterate of this run showed the form
    cannot carry the cost (st.direct_cost), or NNF_COST=direct in the environment asks for the pass over T."""
    updated = [m for m in range(st.nway) if m not in fixed_modes]
    return (st.T.is_cuda and update_rule == "hals" and isinstance(st.eng, _engine.Engine) and bool(updated)
            and not (_dist.is_sharded(st.group) and updated[-1] == 0)
            and not st.direct_cost and os.environ.get("NNF_COST") != "direct")


def run_ntf_steps(st, rank, Ft, n_iter, update_rule, beta, sparsity_coefficients, fixed_modes, normalize, alpha, delta,
                  retired, tol=None):
    """The `for iteration` loop of compute_ntf (ntf.py:313-340) with the device one iteration ahead of the host, like
    nmf.run_steps: the status block of iteration i reaches the host through an asynchronous copy + event and is looked at
    after iteration i+1 has been enqueued (0.91 -> 0.76 ms per iteration at 500^3, rank 30).  `retired(iteration, cost,
    sweeps)` is called in order and returns True to stop; the factors of the stopping iteration are returned, the
    speculative one behind it is dropped.  (Running the cost on a second stream next to the following iteration's MTTKRP
    kernels was tried: 0.78 ms, both want HBM.)

    HALS costs come from the Gram identity (_ntf_cost) while the kernel's own error estimate stays below 5e-4 of the cost
    and -- `tol` given: the caller stops on |cost[i-1] - cost[i]| < tol (ntf.py:337) -- while that difference is further
    from `tol` than the two estimates together.  The first iterate that fails either test is evaluated again by the pass
    over T, and so is every later one; the previous iterate's cost is re-evaluated as well and handed to
    `retired.revise_last`, so that the stopping test only ever compares two costs of the same kind."""
    cuda = st.T.is_cuda
    main = torch.cuda.current_stream(st.T.device) if cuda else None
    pending, result, stop = [], Ft, False
    ident = _identity_cost_applies(st, update_rule, fixed_modes)
    last = None           # (cost, estimate) of the last retired iterate, normalised, while both came from the identity

    def retire():
        nonlocal result, stop, last
        step = pending[0]
        step["ev"].synchronize()
        host = st.host[step["slot"]]
        for i in range(step["nstat"]):
            code = int(host[8 * i + _engine.ST_ERR])
            if code in (_dist.ERR_BEFORE_WINDOW, _dist.ERR_NOT_STOPPED) and step["async0"]:
                raise _GuessMissed()
            if code != 0:
                raise err.EngineError("hals grid barrier timed out; result invalid")
        if step["ident"]:
            if float(host[st.cost_at + 1]) != 0.0:
                raise _IdentityUnreliable()
            c, e = float(host[st.cost_at]) / norm2_host, float(host[st.cost_at + 2]) / norm2_host
            if tol is not None and tol > 0 and last is not None and abs(last[0] - c) < tol + e + last[1]:
                raise _IdentityNearStop()
            last = (c, e)
        pending.pop(0)
        result = step["Ft"]
        if _dist.is_sharded(st.group) and update_rule == "hals" and 0 not in fixed_modes and step["nstat"] >= 1:
            cnt0 = int(host[_engine.ST_CNT]) - 1                  # the sharded mode-0 solve is the first status block
            st.async_ready = st.last_cnt0 is not None and abs(cnt0 - st.last_cnt0) <= 4
            st.last_cnt0 = cnt0
            if step["async0"]:                                    # centre the next blind chunk on this count
                st.async_hits += 1
                st.guess0.value = max(8, min(cnt0 + 4, st.guess0.max_chunk))
        stop = bool(retired(step["it"], float(host[st.cost_at]) / norm2_host,
                            [int(host[8 * i + _engine.ST_CNT]) - 1 for i in range(step["nstat"])]))

    norm2_host = float(st.norm2)       # (one read, before the loop: the blocks carry the un-normalised cost)
    iteration = 0
    while iteration < n_iter:
        st.select(iteration % st.blocks.shape[0])
        st.last_step_async = False
        Ft, nstat = _one_ntf_step_dev(st, rank, Ft, update_rule, beta, sparsity_coefficients, fixed_modes, normalize,
                                      alpha, delta, fuse_next=True, host_norm=True, ident=ident)   # (fuse_next also in the last
        #                    iteration: every cost of a run comes from the same kernel, whatever n_iter_max -- bitwise repeatable)
        st.sync_next = False
        st.host[st.slot].copy_(st.block, non_blocking=cuda)
        pending.append(dict(it=iteration, slot=st.slot, Ft=Ft, nstat=nstat, async0=st.last_step_async, ident=ident,
                            ev=main.record_event() if cuda else _NoEvent()))
        iteration += 1
        try:
            if len(pending) > 1:
                retire()
                if stop:
                    break
            if iteration == n_iter:
                while pending and not stop:
                    retire()
        except _GuessMissed:
            # the blind chunk of the device-side protocol missed the stopping sweep: drop what is in flight and redo this
            # iteration with the exact, host-synchronous protocol (every rank takes this branch: the status is a function
            # of all-reduced sums)
            failed = pending[0]["it"]
            if cuda:
                main.synchronize()
            pending.clear()
            st.sync_next = True
            st.async_misses += 1
            Ft = result
            iteration = failed
        except _IdentityUnreliable:
            # this iteration again, and every later one, with the pass over T (every rank: the words are replicated)
            failed = pending[0]["it"]
            main.synchronize()
            pending.clear()
            st.direct_cost, ident = True, False
            Ft = result
            iteration = failed
            if last is not None and hasattr(retired, "revise_last"):
                # whichever test failed, the iterate before it was costed by the identity: re-evaluate it too -- the stopping test
                # then compares two costs of the same kind.  Its pass leaves the partial product the repeated iteration starts
                # from (st.partial finds it)
                tree = (st.nway == 3 and math.isinf(alpha) and 0 not in fixed_modes and 1 not in fixed_modes
                        and hasattr(st.eng, "mttkrp3_from_partial"))
                words = torch.zeros(3, dtype=torch.float64, device=st.T.device)
                _ntf_cost(st.eng, st, Ft, update_rule, beta, sparsity_coefficients, words, fuse_next=tree, host_norm=True)
                retired.revise_last(float(words[0]) / norm2_host)
            last = None
    if cuda and pending:
        main.synchronize()
    return result


class _GuessMissed(Exception):
    """Leading-mode-sharded run: the device-side stopping decision of the mode-0 solve missed (status 3 / 4)."""


class _IdentityUnreliable(Exception):
    """The Gram-identity cost's own error estimate is above 5e-4 of the cost (an almost exact fit)."""


class _IdentityNearStop(_IdentityUnreliable):
    """Two consecutive identity costs differ by `tol` give or take their error estimates: the stopping test needs better."""


class _NoEvent:
    def synchronize(self):
        pass


def compute_ntf(tensor_in, rank, factors_in, n_iter_max=100, tol=1e-8,
                update_rule="hals", beta=2,
                sparsity_coefficients=[], fixed_modes=[], normalize=[],
                verbose=False, return_costs=False, alpha=0.5, delta=0.01, sweep_log=None):
    """Outer loop of ntf.py:288-344.  Returns the list of factors (dim x R each) [, costs, toc]."""
    dev = device_of(tensor_in, *factors_in)
    eng = _engine.get_engine(dev)
    T = to_dev(tensor_in, dev)
    st = _NtfState(eng, T)
    Ft = [to_dev_t(f, dev).clone() for f in factors_in]
    nb_modes = T.dim()
    if sparsity_coefficients is None or len(sparsity_coefficients) != nb_modes:
        print("Irrelevant number of sparsity coefficient (different from the number of modes), they have been set to None.")
        sparsity_coefficients = [None for i in range(nb_modes)]
    if fixed_modes is None:
        fixed_modes = []
    if normalize is None or len(normalize) != nb_modes:
        print("Irrelevant number of normalization booleans (different from the number of modes), they have been set to False.")
        normalize = [False for i in range(nb_modes)]
    cost_fct_vals, toc = [], []
    tic = time.time()

    def retired(iteration, cost, sweeps):
        """Host side of one finished iteration (ntf.py:325-340); True = the stopping test fired."""
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
        cost_fct_vals[-1] = cost
    retired.revise_last = revise_last

    Ft = run_ntf_steps(st, rank, Ft, n_iter_max, update_rule, beta, sparsity_coefficients, fixed_modes, normalize,
                       alpha, delta, retired, tol=tol)
    # the reference returns np.array(factors), which needs equal mode sizes on NumPy >= 1.24; a list always works
    factors = [like_input(f.t(), factors_in[i]) for i, f in enumerate(Ft)]
    if return_costs:
        return factors, cost_fct_vals, toc
    return factors


def one_ntf_step(unfolded_tensors, rank, in_factors, norm_tensor, update_rule, beta,
                 sparsity_coefficients, fixed_modes, normalize,
                 alpha=0.5, delta=0.01):
    """One pass over the modes (ntf.py:422-477).  `unfolded_tensors` is the reference's list of unfoldings; the tensor is
    rebuilt from the mode-0 unfolding (a plain reshape).  Returns (factors, cost)."""
    dev = device_of(*unfolded_tensors, *in_factors)
    eng = _engine.get_engine(dev)
    dims = [int(u.shape[0]) for u in unfolded_tensors]
    T = to_dev(unfolded_tensors[0], dev).reshape(dims)
    st = _NtfState(eng, T)
    Ft = [to_dev_t(f, dev) for f in in_factors]
    Ft, nstat = _one_ntf_step_dev(st, rank, Ft, update_rule, beta, list(sparsity_coefficients), fixed_modes, normalize,
                                  alpha, delta)
    host = st.block.cpu()
    return [like_input(f.t(), in_factors[i]) for i, f in enumerate(Ft)], float(host[st.cost_at])
