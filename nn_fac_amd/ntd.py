"""NTD (nonnegative Tucker decomposition) driver on the MI355X engine -- drop-in for nn_fac/ntd.py
(ntd :27-246, compute_ntd :248-433, one_ntd_step :436-645, one_ntd_step_mu :658-698), tensors of any order >= 3 (every mode
product runs through the 3-way kernel on a VIEW (left, I_n, right) of the tensor; the core update merges the trailing core
modes into one -- their Gram is the Kronecker product -- so that nnf_ntd_core_pg_f32's three-mode loop serves it).

HALS step, per updated mode n (factors kept transposed, r_n x I_n; statement -> C ABI):

    elemprod[i] = F_i^T F_i, i != n                      (ntd.py:534-537)  nnf_gram_f32
    UtU = <core x_{i!=n} elemprod, core> over modes != n (ntd.py:539-544)  tiny (core-sized) -- torch on the device
    temp = T x_{i!=n} F_i^T                              (ntd.py:550)      nnf_ttm3_f32: ONE streaming pass over T with the
        W^T X / X H^T kernels (first or last axis contracted), then the same product on the small intermediate;
        T x_0 F_0^T is shared by the mode-1 and mode-2 updates (F_0 does not change between them)
    UtM = (<temp, core> over modes != n)^T               (ntd.py:555-556)  tiny GEMM on the device
    F_n = hals_nnls_acc(UtM, UtU, F_n^T)^T               (ntd.py:571-573)  nnf_hals_solve_f32
    core: step, <= 300 projected-gradient steps, error   (ntd.py:581-619,639) nnf_ntd_core_pg_f32 (one launch, fp64 in LDS)

MU step: the factor updates are mu_betadivmin on the transposed unfoldings (nnf_mu_right_f32 on (prod other dims) x I_n:
the last mode is a view of T, modes 0 and 1 are materialised once per run), the core update mu_tensorial (mu.py:99-159) is the right-update accumulation of the mode-0 problem
(nnf_mu_right_accum_f32 -- U V is never materialised) followed by two core-sized contractions, and the cost is
nnf_betadiv_f32 on the mode-0 problem.

Tensor-sized work goes through the C ABI; what stays in torch is core-sized (prod(ranks) entries) plumbing.
"""
import math
import time
import warnings

import numpy as np
import torch

from .utils import errors as err
from .utils import initialize_factors as init_factors
from . import engine as _engine
from ._convert import device_of, to_dev, to_dev_t, like_input
from .update_rules.nnls import sweep_budget


def ntd(tensor, ranks, init="random", core_0=None, factors_0=[], n_iter_max=100, tol=1e-6,
        update_rule="hals", beta=2,
        sparsity_coefficients=[], fixed_modes=[], normalize=[], mode_core_norm=None,
        verbose=False, return_costs=False, deterministic=False, seed=0):
    """Nonnegative Tucker decomposition of `tensor` (reference docstring: ntd.py:32-203)."""
    nb_modes = len(tensor.shape)
    if deterministic:
        np.random.seed(seed)
    if type(ranks) is int:   # if only one rank is provided, use it for all modes
        ranks = [ranks for i in range(nb_modes)]
    elif len(ranks) != nb_modes:
        raise err.InvalidRanksException("The number of ranks is different than the dim of the tensor, which is incorrect.") from None
    for i in range(nb_modes):
        if ranks[i] > tensor.shape[i]:
            ranks[i] = tensor.shape[i]
            warnings.warn(f"The {i}-th mode rank was larger than the shape of the tensor, which is incorrect (rank: {ranks[i]}, tensor shape: {tensor.shape[i]}). The rank was then set to the shape of the tensor.")
    _engine.check_rank(max(ranks), "ntd")
    if update_rule == "hals":
        assert beta == 2, f"Beta parameter is only used for MU update rule. Please set update_rule to 'mu' to use another beta value than 2. (Current setting: beta = {beta} and update_rule = {update_rule})."
    if init.lower() == "custom":
        factors = factors_0
        core = core_0
        if len(factors) != nb_modes:
            raise err.CustomNotEngouhFactors("Custom initialization, but not enough factors")
        else:
            for array in factors:
                if array is None:
                    raise err.CustomNotValidFactors("Custom initialization, but (at least) one factor is set to 'None'")
            if core is None:
                raise err.CustomNotValidCore("Custom initialization, but the core is set to 'None'")
    else:
        core, factors = init_factors.ntd_initialization(tensor, ranks, init, deterministic=deterministic, seed=seed)
    if (init.lower() == "chromas") and (0 not in fixed_modes):
        fixed_modes.append(0)
    return compute_ntd(tensor, ranks, core, factors, n_iter_max=n_iter_max, tol=tol,
                       update_rule=update_rule, beta=beta,
                       sparsity_coefficients=sparsity_coefficients, fixed_modes=fixed_modes,
                       normalize=normalize, mode_core_norm=mode_core_norm,
                       verbose=verbose, return_costs=return_costs, deterministic=deterministic, seed=seed)


class _NtdState:
    """Device-resident tensor, its squared norm, (MU only) the materialised unfoldings, and the per-step status block."""

    def __init__(self, eng, T):
        if T.dim() < 3:
            raise NotImplementedError("NTD needs a tensor of order >= 3")
        self.eng = eng
        self.T = T.contiguous()
        self.nway = self.T.dim()
        self.t0 = self.T.view(self.T.shape[0], -1)
        self.norm2 = eng.dot(self.t0, self.t0)          # float64 device scalar, ||T||^2 (read once by the driver)
        self.norm2_host = None
        self._unf_t = {}
        # one HALS status block (8 doubles) per mode, then 6 doubles of the core update, then the cost; two of them with
        # pinned host mirrors: compute_ntd enqueues iteration i+1 before it looks at the block of iteration i
        self.pg_at = 8 * self.nway
        self.cost_at = self.pg_at + 6
        self.blocks = torch.zeros((2, self.cost_at + 2), dtype=torch.float64, device=T.device)
        self.host = torch.zeros((2, self.cost_at + 2), dtype=torch.float64)
        if T.is_cuda:
            self.host = self.host.pin_memory()
        self.select(0)

    def select(self, slot):
        self.slot = slot
        self.block = self.blocks[slot]

    def unfolded_t(self, mode):
        """tl.unfold(T, mode)^T as a contiguous (prod(other dims)) x I_mode matrix (MU path).  The last mode is a view of T;
        modes 0 and 1 are materialised once per run."""
        if mode not in self._unf_t:
            self._unf_t[mode] = torch.movedim(self.T, mode, -1).reshape(-1, self.T.shape[mode]).contiguous()
        return self._unf_t[mode]

    def norm_sq(self):
        if self.norm2_host is None:
            self.norm2_host = float(self.norm2)
        return self.norm2_host


def _normalize_core(core, mode_core_norm):
    """ntd.py:621-626: every mode-`mode_core_norm` slice of the core divided by its l2 norm (zero slices are left alone)."""
    unf = torch.movedim(core, mode_core_norm, 0)
    nrm = unf.reshape(unf.shape[0], -1).norm(dim=1)
    scale = torch.where(nrm != 0, 1.0 / nrm, torch.ones_like(nrm))
    shape = [-1] + [1] * (core.dim() - 1)
    return torch.movedim(unf * scale.view(shape), 0, mode_core_norm).contiguous()


def _mode_mul(eng, X, A, axis):
    """X x_axis A for a rank-sized operand A (p x q, q = X.shape[axis]) -- tl.tenalg.mode_dot(X, A, axis): the contraction is
    nnf_small_gemm_f32 on the matrix view with `axis` first (out = A @ unfold(X, axis)); moving the axis there and back is a
    copy, not arithmetic."""
    Xm = torch.movedim(X, axis, 0).contiguous()
    out = eng.small_gemm(A.contiguous(), Xm.view(Xm.shape[0], -1))
    return torch.movedim(out.view([A.shape[0]] + list(Xm.shape[1:])), 0, axis).contiguous()


def _core_mode_dots(eng, core, mats, skip=None):
    """tl.tenalg.multi_mode_dot(core, mats, skip) (ntd.py:539, :672) -- a chain of rank-sized mode products (mats[i]: new_dim x old_dim),
    every link a nnf_small_gemm_f32 launch."""
    out = core
    for i, M in enumerate(mats):
        if i == skip:
            continue
        out = _mode_mul(eng, out, M, i)
    return out


def _ttm(eng, X, Ft_i, pos):
    """Mode product of a contiguous tensor X (any order) along axis `pos` with a transposed factor Ft_i (r x d): the 3-way
    kernel nnf_ttm3_f32 on the view (left, d, right).  First axis -> the W^T X kernel, result in place; last axis -> the
    X H^T kernel, the new axis comes out FIRST; a middle axis -> the VALU slab kernel, in place.  Returns (tensor, new_pos)."""
    sh = list(X.shape)
    left, right = 1, 1
    for d in sh[:pos]:
        left *= int(d)
    for d in sh[pos + 1:]:
        right *= int(d)
    r = Ft_i.shape[0]
    if left == 1:                       # first axis (or only extents of 1 before it): the result stays where it is
        out = eng.ttm3(X.reshape(sh[pos], right, 1), Ft_i, 0)                   # (r, right, 1)
        return out.view(sh[:pos] + [r] + sh[pos + 1:]), pos
    if right == 1:                      # last axis (or only extents of 1 behind it): the new axis comes out first
        out = eng.ttm3(X.reshape(1, left, sh[pos]), Ft_i, 2)                    # (r, 1, left)
        return out.view([r] + sh[:pos] + sh[pos + 1:]), 0
    out = eng.ttm3(X.reshape(left, sh[pos], right), Ft_i, 1)                    # (left, r, right)
    return out.view(sh[:pos] + [r] + sh[pos + 1:]), pos


def _contract_others(eng, temp, axes, core, n):
    """(<temp, core> over the modes != n)^T (ntd.py:555-556): temp carries the rank extents of the other modes in the order
    `axes[:-1]` and I_n last; r_n x I_n = (core with mode n first, the others in that order) @ (temp as P x I_n)."""
    others = axes[:-1]
    A = core.permute([n] + others).reshape(core.shape[n], -1).contiguous()
    B = temp.reshape(-1, temp.shape[-1])
    return eng.xty(B, A)          # A (r_n x P) @ B (P x I_n): the W^T X kernel with P = prod(other ranks) "rows"


def _one_ntd_step_dev(st, core_in, Ft_in, sparsity_coefficients, fixed_modes, normalize, mode_core_norm, alpha, delta):
    """ntd.py:514-645 on the device.  Ft: transposed factors (r_n x I_n).  Returns (core, Ft, number of HALS solves);
    the cost is left in st.block[st.cost_at], the status words of the solves in st.block[0 : 8 N], of the core update behind."""
    eng = st.eng
    N = st.nway
    for fixed_value in fixed_modes:
        sparsity_coefficients[fixed_value] = None
    core = core_in.clone()
    Ft = list(Ft_in)
    dev = st.T.device
    modes_list = [m for m in range(N) if m not in fixed_modes]
    if not modes_list:
        raise UnboundLocalError("one_ntd_step needs at least one non-fixed factor mode (ntd.py:581 reuses its 'temp')")
    nstat = 0
    W0 = None           # T x_0 F_0^T, shared by the updates of every mode but the first
    grams = [None] * N
    deterministic = math.isinf(alpha)
    for mode in modes_list:
        if not deterministic:
            torch.cuda.synchronize(dev)
            t0 = time.time()
        for i in range(N):
            if i != mode:
                grams[i] = eng.gram(Ft[i])                                    # elemprod (ntd.py:534-537)
        others = [i for i in range(N) if i != mode]
        tmp = _core_mode_dots(eng, core, [grams[i] if i != mode else None for i in range(N)], skip=mode)
        # UtU = <tmp, core> over the other modes (ntd.py:544) = unfold(tmp, mode) unfold(core, mode)^T: the X H^T kernel on the two
        # r_n x prod(other ranks) unfoldings
        unf = lambda t: torch.movedim(t, mode, 0).reshape(t.shape[mode], -1).contiguous()   # noqa: E731
        UtU = eng.xht(unf(core), unf(tmp))                                                  # r_n x r_n
        # temp = T x_{i != mode} F_i^T (ntd.py:550).  The pass over T contracts its first axis (W^T X kernel; shared by all
        # modes but the first) or, for mode 0, its last one (X H^T kernel); what follows works on a tensor I/r times smaller:
        # the modes above `mode` are last when their turn comes (taken in decreasing order, each result moves to the front),
        # the ones below sit in the middle (slab kernel).  `mode` itself always ends up last.
        if mode == 0:
            temp, _ = _ttm(eng, st.T, Ft[N - 1], N - 1)
            axes = [N - 1] + list(range(N - 1))
            rest = list(range(N - 2, 0, -1))
        else:
            if W0 is None:
                W0, _ = _ttm(eng, st.T, Ft[0], 0)
            temp, axes = W0, list(range(N))
            rest = [i for i in range(N - 1, 0, -1) if i != mode]
        for i in rest:
            pos = axes.index(i)
            temp, newpos = _ttm(eng, temp, Ft[i], pos)
            if newpos != pos:                                                 # a last axis came out first
                axes = [i] + axes[:pos] + axes[pos + 1:]
        UtM = _contract_others(eng, temp, axes, core, mode)                   # (ntd.py:555-556)
        new = Ft[mode].clone()
        budget = 100
        if not deterministic:
            torch.cuda.synchronize(dev)
            timer = time.time() - t0
            probe = new.clone()
            t0 = time.time()
            eng.hals_sweeps(UtM, UtU, probe, 1, sparsity=sparsity_coefficients[mode], normalize=normalize[mode])
            torch.cuda.synchronize(dev)
            rho = timer / max(time.time() - t0, 10e-7) if timer else 100000
            budget = max(1, sweep_budget(100, alpha, rho))
        eng.hals_solve(UtM, UtU, new, budget, delta=delta, sparsity=sparsity_coefficients[mode],
                       normalize=normalize[mode], status=st.block[8 * nstat:8 * nstat + 8])
        nstat += 1
        Ft[mode] = new
        if mode == 0:
            W0 = None
    last = modes_list[-1]
    # all_MtX = temp x_last F_last^T with the NEW factor; all_MtM = elemprod with the last Gram refreshed (ntd.py:581-583):
    # `last` is temp's last axis, its product comes out first -> back to the core's mode order (core-sized)
    mtx, _ = _ttm(eng, temp.contiguous(), Ft[last], len(axes) - 1)
    order = [last] + axes[:-1]
    all_MtX = mtx.permute([order.index(i) for i in range(N)]).contiguous()
    grams[last] = eng.gram(Ft[last])
    for i in range(N):
        if grams[i] is None:            # a fixed mode that was never "other": cannot happen with >= 1 free mode
            grams[i] = eng.gram(Ft[i])
    sparse = 0 if sparsity_coefficients[-1] is None else sparsity_coefficients[-1]
    pg = st.block[st.pg_at:st.pg_at + 6]
    if N == 3:
        eng.ntd_core_pg(core, all_MtX, grams, sparse, delta, 300, st.norm_sq(), status=pg)
    else:
        # the projected-gradient kernel loops over three modes: the trailing core modes are merged into one, whose Gram is
        # the Kronecker product of theirs (core x_2 M_2 x_3 M_3 ... = merged core x_2 (M_2 (x) M_3 ...), row-major merge;
        # sigma_max of a Kronecker product is the product of the sigma_max: same step, ntd.py:588-596)
        tail = 1
        for d in core.shape[2:]:
            tail *= int(d)
        if tail > 128:
            raise NotImplementedError("NTD of order > 3: the product of the core's trailing extents must be <= 128")
        Mk = grams[2]
        for g in grams[3:]:
            Mk = torch.kron(Mk.contiguous(), g.contiguous())
        c3 = core.view(core.shape[0], core.shape[1], tail)
        eng.ntd_core_pg(c3, all_MtX.view(c3.shape), [grams[0], grams[1], Mk], sparse, delta, 300, st.norm_sq(), status=pg)
    cost = st.block[st.cost_at:st.cost_at + 1]
    if normalize[-1]:
        core = _normalize_core(core, mode_core_norm)
    # ||T - core x_0 F_0 x_1 F_1 ...||^2.  The reference evaluates it in the Gram form ||T||^2 - 2<MtX, core> + <MtM core,
    # core> (ntd.py:635-639) in fp64; with fp32 contractions that form is good to ~1e-7 ||T||^2, which is the size of the cost
    # itself for a near-exact fit (tools/stress_tensor.py: 1e-2 ... 2e-1 relative error on normalised costs of 1e-6).  One
    # streaming pass over T against the mode-0 matrix form (the cost kernel, product never materialised) is exact to ~1e-6
    # relative and costs ~1 % of the iteration.
    eng.frob_resid(st.t0, Ft[0], _core_expand_mode0(eng, core, Ft), out=cost)
    sparsity_error = None
    for index, sp in enumerate(sparsity_coefficients):
        if sp:
            if index < N:      # np.linalg.norm(factor, ord=1): max column abs-sum = max row abs-sum of the transposed factor
                term = 2 * sp * Ft[index].abs().sum(dim=1).max().double()
            elif index == N:
                term = 2 * sp * core.abs().sum().double()
            else:
                raise NotImplementedError("TODEBUG: Too many sparsity coefficients, should have been raised before.")
            sparsity_error = term if sparsity_error is None else sparsity_error + term
    if sparsity_error is not None:
        cost.add_(sparsity_error)
    cost.div_(st.norm2)
    return core, Ft, nstat


def _core_expand_mode0(eng, core, Ft, st=None):
    """unfold(core x_1 F_1 x_2 F_2 ..., 0): the r_0 x prod(I_1..) right operand of the mode-0 matrix problem (ntd.py:635-639,
    :672): rank-sized mode products, nnf_small_gemm_f32 each; the result is r_0 / I_0 of the tensor's size.  With `st` the result
    is remembered together with the operands it was made from: the cost at the end of an MU iteration and the mode-0 update
    that opens the next one expand the SAME core with the SAME factors."""
    if st is not None:
        hit = getattr(st, "_expand0", None)
        if hit is not None and hit[0] is core and len(hit[1]) == len(Ft) - 1 and all(a is b for a, b in zip(hit[1], Ft[1:])):
            return hit[2]
    w = core
    for i in range(core.dim() - 1, 0, -1):
        w = _mode_mul(eng, w, Ft[i].t(), i)
    w = w.reshape(w.shape[0], -1).contiguous()
    if st is not None:
        st._expand0 = (core, tuple(Ft[1:]), w)
    return w


def _mu_tensorial_dev(st, core, Ft, beta):
    """mu.py:138-159.  num/den = (L2 | L1) x_i F_i^T: the mode-0 product is the right-update accumulation of the matrix
    problem T_(0) ~ F_0 V0 (one fused pass over T), the others are mode products of an r_0 x I_1 x ... intermediate."""
    eng = st.eng
    N = st.nway
    if beta < 0:
        raise err.InvalidArgumentValue("Invalid value for beta: negative one.") from None
    V0 = _core_expand_mode0(eng, core, Ft)
    num, den, dvec = eng.mu_right_accum(st.t0, Ft[0], V0, beta)
    dims = [int(d) for d in st.T.shape[1:]]

    def down(x):   # (r_0, I_1 * ... ) -> x_i F_i^T for every i >= 1 -> the core's shape
        w, axes = x.view([x.shape[0]] + dims), list(range(N))
        for i in range(N - 1, 0, -1):
            pos = axes.index(i)
            w, newpos = _ttm(eng, w.contiguous(), Ft[i], pos)
            if newpos != pos:
                axes = [i] + axes[:pos] + axes[pos + 1:]
        return w.permute([axes.index(i) for i in range(N)]).contiguous()
    num3 = down(num)
    if dvec is not None:   # beta = 1: L1 = ones -> outer product of the factors' column sums
        den3 = dvec.view([-1] + [1] * (N - 1))
        for i in range(1, N):
            shape = [1] * N
            shape[i] = -1
            den3 = den3 * Ft[i].sum(dim=1).double().view(shape)
        den3 = den3.float()
    else:
        den3 = down(den)
    from .utils.beta_divergence import gamma_beta
    ratio = num3 / den3
    g = gamma_beta(beta)
    if g != 1:
        ratio = ratio ** g
    return torch.clamp(core * ratio, min=1e-12).contiguous()


def _one_ntd_step_mu_dev(st, core_in, Ft_in, beta, fixed_modes, normalize, mode_core_norm):
    """ntd.py:664-698 on the device; the (un-normalised, ntd.py:696) cost is left in st.block[st.cost_at]."""
    eng = st.eng
    N = st.nway
    core = core_in.clone()
    Ft = list(Ft_in)
    for mode in [m for m in range(N) if m not in fixed_modes]:
        # V = unfold(core x_{i != mode} F_i, mode): r_mode x prod(other dims), core-width GEMMs
        if mode == 0:
            V = _core_expand_mode0(eng, core, Ft, st)      # (what the previous iteration's cost expanded, if nothing changed)
        else:
            mats = [Ft[i].t() if i != mode else None for i in range(N)]
            V = torch.movedim(_core_mode_dots(eng, core, mats, skip=mode), mode, 0)
            V = V.reshape(V.shape[0], -1).contiguous()
        # mu_betadivmin(F, V, unfold(T, mode)) (ntd.py:672) on the TRANSPOSED problem unfold^T ~ V^T F^T: the unfolding is
        # short and fat (I_mode rows), its transpose gives the streaming kernel prod(other dims) rows to split over
        Ft[mode] = eng.mu_right(st.unfolded_t(mode), V, Ft[mode], beta)
    core = _mu_tensorial_dev(st, core, Ft, beta)
    if normalize[-1]:
        core = _normalize_core(core, mode_core_norm)
    eng.betadiv(st.t0, Ft[0], _core_expand_mode0(eng, core, Ft, st), beta, out=st.block[st.cost_at:st.cost_at + 1])
    return core, Ft


def compute_ntd(tensor_in, ranks, core_in, factors_in, n_iter_max=100, tol=1e-6,
                update_rule="hals", beta=2,
                sparsity_coefficients=[], fixed_modes=[], normalize=[], mode_core_norm=None,
                verbose=False, return_costs=False, deterministic=False, seed=0, sweep_log=None, pg_log=None):
    """Outer loop of ntd.py:355-433.  Returns (core, factors) [, costs, toc]; sweep_log / pg_log (not in the reference)
    collect the inner sweep counts and projected-gradient iteration counts."""
    _engine.check_rank(max(ranks), "compute_ntd")
    dev = device_of(tensor_in, core_in, *factors_in)
    eng = _engine.get_engine(dev)
    st = _NtdState(eng, to_dev(tensor_in, dev))
    core = to_dev(core_in, dev).clone().contiguous()
    Ft = [to_dev_t(f, dev).clone() for f in factors_in]
    nb_modes = st.T.dim()
    if sparsity_coefficients is None or len(sparsity_coefficients) != nb_modes + 1:
        print("Irrelevant number of sparsity coefficient (different from the number of modes + 1 for the core), they have been set to None.")
        sparsity_coefficients = [None for i in range(nb_modes + 1)]
    if fixed_modes is None:
        fixed_modes = []
    if normalize is None or len(normalize) != nb_modes + 1:
        print("Irrelevant number of normalization booleans (different from the number of modes + 1 for the core), they have been set to False.")
        normalize = [False for i in range(nb_modes + 1)]
    if normalize[-1] and (mode_core_norm is None or mode_core_norm < 0 or mode_core_norm >= nb_modes):
        print("The core was asked to be normalized, but an invalid mode was specified. Normalization has been set to False.")
        normalize[-1] = False
    if not normalize[-1] and (mode_core_norm is not None and mode_core_norm >= 0 and mode_core_norm < nb_modes):
        print("The core was asked NOT to be normalized, but mode_core_norm was set to a valid norm. Is this a mistake?")
    cost_fct_vals, toc = [], []
    tic = time.time()
    if update_rule not in ("hals", "mu"):
        raise err.InvalidArgumentValue(f"The update rule provided is not valid. Please choose between 'hals' and 'mu' (Got {update_rule}).")
    cuda = st.T.is_cuda
    main = torch.cuda.current_stream(st.T.device) if cuda else None
    pending, stop = [], False
    result = (core, Ft)

    def retire():
        """Host side of one finished iteration (ntd.py:410-428), one iteration behind the device."""
        nonlocal result, stop
        iteration, slot, core_i, Ft_i, nstat, ev = pending.pop(0)
        if ev is not None:
            ev.synchronize()
        host = st.host[slot]
        cost = float(host[st.cost_at])
        for i in range(nstat):
            if int(host[8 * i + _engine.ST_ERR]) != 0:
                raise err.EngineError("hals grid barrier timed out; result invalid")
        if sweep_log is not None:
            sweep_log.extend(int(host[8 * i + _engine.ST_CNT]) - 1 for i in range(nstat))
        if update_rule == "hals" and int(host[st.pg_at + 5]) != 0:
            raise err.EngineError("NTD core update: grid barrier timed out; result invalid")
        if pg_log is not None and update_rule == "hals":
            pg_log.append(int(host[st.pg_at]))
        result = (core_i, Ft_i)
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
            stop = True

    for iteration in range(n_iter_max):
        st.select(iteration % 2)
        nstat = 0
        if update_rule == "hals":
            core, Ft, nstat = _one_ntd_step_dev(st, core, Ft, sparsity_coefficients, fixed_modes, normalize, mode_core_norm,
                                                math.inf if deterministic else 0.5, 0.01)
        else:
            core, Ft = _one_ntd_step_mu_dev(st, core, Ft, beta, fixed_modes, normalize, mode_core_norm)
        st.host[st.slot].copy_(st.block, non_blocking=cuda)
        pending.append((iteration, st.slot, core, Ft, nstat, main.record_event() if cuda else None))
        if len(pending) > 1:
            retire()
            if stop:
                break
    while pending and not stop:
        retire()
    if cuda and pending:
        main.synchronize()
    core, Ft = result
    core_out = like_input(core, core_in)
    factors = [like_input(f.t(), factors_in[i]) for i, f in enumerate(Ft)]
    if return_costs:
        return core_out, factors, cost_fct_vals, toc
    return core_out, factors


def one_ntd_step(tensor, ranks, in_core, in_factors, norm_tensor,
                 sparsity_coefficients, fixed_modes, normalize, mode_core_norm,
                 alpha=0.5, delta=0.01):
    """One HALS pass over the modes + projected-gradient core update (ntd.py:436-645).
    Returns (core, factors, normalised cost).  `norm_tensor` is accepted for signature parity; ||T||^2 is recomputed on
    the device."""
    _engine.check_rank(max(ranks), "one_ntd_step")
    dev = device_of(tensor, in_core, *in_factors)
    eng = _engine.get_engine(dev)
    st = _NtdState(eng, to_dev(tensor, dev))
    core = to_dev(in_core, dev).contiguous()
    Ft = [to_dev_t(f, dev) for f in in_factors]
    core, Ft, nstat = _one_ntd_step_dev(st, core, Ft, sparsity_coefficients, fixed_modes, normalize, mode_core_norm,
                                        alpha, delta)
    host = st.block.cpu()
    for i in range(nstat):
        if int(host[8 * i + _engine.ST_ERR]) != 0:
            raise err.EngineError("hals grid barrier timed out; result invalid")
    return like_input(core, in_core), [like_input(f.t(), in_factors[i]) for i, f in enumerate(Ft)], float(host[st.cost_at])


def one_ntd_step_mu(tensor, ranks, in_core, in_factors, beta, norm_tensor,
                    fixed_modes, normalize, mode_core_norm):
    """One MU pass over the modes and the core (ntd.py:658-698).  Returns (core, factors, beta-divergence)."""
    _engine.check_rank(max(ranks), "one_ntd_step_mu")
    dev = device_of(tensor, in_core, *in_factors)
    eng = _engine.get_engine(dev)
    st = _NtdState(eng, to_dev(tensor, dev))
    core = to_dev(in_core, dev).contiguous()
    Ft = [to_dev_t(f, dev) for f in in_factors]
    core, Ft = _one_ntd_step_mu_dev(st, core, Ft, beta, fixed_modes, normalize, mode_core_norm)
    return (like_input(core, in_core), [like_input(f.t(), in_factors[i]) for i, f in enumerate(Ft)],
            float(st.block[st.cost_at]))
