"""CPU oracle for the nn_fac HALS-NNLS / beta-MU hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference algorithms (ax-le/nn-fac 0.3.4).
It is the checker for the HIP engine, never the thing shipped: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``nn_fac_amd/`` imports it.

Parity status: PINNED.  ``oracle/gen_golden.py`` runs the real reference (imported
from /root/reference in the build container) on seeded inputs, first re-asserting
the reference's own known answers (tests/NMF_tests.py:35-41,68-135), and stores
inputs + outputs in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those fixtures to ~1e-12 (fp64).

Each function cites the reference lines it follows.  The statement sequence (same
``np.dot`` shapes, same Python row loop, same temporaries) is kept so that timing the
oracle is a fair stand-in for timing the reference (``cpu_baseline.kind = "port"``).

The tensor helpers restate the documented semantics of tensorly 0.6.0 (pinned in the
reference's setup.py:30, not vendored in /root/reference and not installable offline).
"""
import math
import time

import numpy as np

EPS_MU = 1e-12  # reference: nn_fac/update_rules/mu.py:18


# --------------------------------------------------------------------------------------
# exceptions (nn_fac/utils/errors.py:8-18).  They derive from BaseException there too.
# --------------------------------------------------------------------------------------
class ArgumentException(BaseException):
    pass


class InvalidArgumentValue(ArgumentException):
    pass


class CustomNotValidFactors(ArgumentException):
    pass


class CustomNotEngouhFactors(ArgumentException):
    pass


class InvalidInitializationType(ArgumentException):
    pass


class OptimException(BaseException):
    pass


class ZeroColumnWhenUnautorized(OptimException):
    pass


# --------------------------------------------------------------------------------------
# a1: accelerated HALS NNLS  (nn_fac/update_rules/nnls.py:24-198)
# --------------------------------------------------------------------------------------
def hals_nnls_acc(UtM, UtU, in_V, maxiter=500, atime=None, alpha=0.5, delta=0.01,
                  sparsity_coefficient=None, normalize=False, nonzero=False,
                  sweep_log=None):
    """Gauss-Seidel sweep over the rows of V with projection on the nonnegative orthant.

    nnls.py:130-135 argument checks; :147 works on a copy; :149-152 initial state;
    :156 loop condition; :158-185 row update; :187-196 bookkeeping; :198 return.
    ``sweep_log`` (oracle-only) receives nodelta of every sweep.
    """
    for name, a in (("UtM", UtM), ("UtU", UtU), ("in_V", in_V)):
        if len(np.shape(a)) != 2:
            raise ArgumentException(f"Argument {name} should be a matrix, got shape {np.shape(a)}.")
    r, n = np.shape(UtM)
    if not in_V.size:
        # nnls.py:138-145 -- effectively dead in the reference (np.linalg.linalg); kept for shape.
        V = np.linalg.solve(UtU, UtM)
        V[V < 0] = 0
        V = (np.sum(UtM * V) / np.sum(UtU * np.dot(V, V.T))) * V
    else:
        V = in_V.copy()

    rho, eps0, cnt, eps = 100000, 0, 1, 1
    t0 = time.time()
    while eps >= delta * eps0 and cnt <= 1 + alpha * rho and cnt <= maxiter:
        nodelta = 0
        for k in range(r):
            if UtU[k, k] != 0:
                if sparsity_coefficient != None:  # noqa: E711  (reference compares with !=)
                    step = np.maximum((UtM[k, :] - UtU[k, :] @ V - sparsity_coefficient * np.ones(n)) / UtU[k, k],
                                      -V[k, :])
                else:
                    step = np.maximum((UtM[k, :] - UtU[k, :] @ V) / UtU[k, k], -V[k, :])
                V[k, :] = V[k, :] + step
                nodelta = nodelta + np.dot(step, np.transpose(step))
                if nonzero and (V[k, :] == 0).all():
                    V[k, :] = 1e-16 * np.max(V)
            elif nonzero:
                raise ZeroColumnWhenUnautorized("Column " + str(k) + " of U is zero with nonzero condition")
            if normalize:
                nrm = np.linalg.norm(V[k, :])
                if nrm != 0:
                    V[k, :] /= nrm
                else:
                    V[k, :] = 1 / n ** (1 / 2)
        if cnt == 1:
            eps0 = nodelta
            btime = max(time.time() - t0, 10e-7)
            if atime:
                rho = atime / btime
        eps = nodelta
        if sweep_log is not None:
            sweep_log.append(float(nodelta))
        cnt += 1
    return V, eps, cnt, rho


# --------------------------------------------------------------------------------------
# f4: HALS NNLS coupled to a target matrix (PARAFAC2's caller of the path)
#     nn_fac/update_rules/nnls.py:204-352;  min_{V>=0} ||M - UV||_F^2 + mu ||V - Vtarget||_F^2
# --------------------------------------------------------------------------------------
def hals_coupling_nnls_acc(UtM, UtU, in_V, Vtarget, mu, maxiter=500, atime=None, alpha=0.5, delta=0.01,
                           normalize=False, nonzero=False, sweep_log=None):
    """nnls.py:296-305 start value; :307-310 state; :314 loop condition (maxiter first, same predicate);
    :316-329 coupled row update -- the zero-diagonal test is on UtU[k,k], NOT on UtU[k,k] + mu;
    :331-332 plain ValueError (not ZeroColumnWhenUnautorized); :334-340 normalisation; :342-350 bookkeeping."""
    r, n = np.shape(UtM)
    if not in_V.size:
        V = np.linalg.solve(UtU, UtM)      # (np.linalg.linalg.solve in the reference: removed from NumPy 2)
        V[V < 0] = 0
        V = (np.sum(UtM * V) / np.sum(UtU * np.dot(V, V.T))) * V
    else:
        V = in_V.copy()
    rho, eps0, cnt, eps = 100000, 0, 1, 1
    t0 = time.time()
    while cnt <= maxiter and eps >= delta * eps0 and cnt <= 1 + alpha * rho:
        nodelta = 0
        for k in range(r):
            if UtU[k, k] != 0:
                step = np.maximum((UtM[k, :] - UtU[k, :] @ V + mu * (Vtarget[k, :] - V[k, :])) / (UtU[k, k] + mu),
                                  -V[k, :])
                V[k, :] = V[k, :] + step
                nodelta = nodelta + np.dot(step, np.transpose(step))
                if nonzero and (V[k, :] == 0).all():
                    V[k, :] = 1e-16 * np.max(V)
            elif nonzero:
                raise ValueError("Column " + str(k) + " is zero with nonzero condition")
            if normalize:
                nrm = np.linalg.norm(V[k, :])
                if nrm != 0:
                    V[k, :] /= nrm
                else:
                    V[k, :] = 1 / n ** (1 / 2)
        if cnt == 1:
            eps0 = nodelta
            btime = max(time.time() - t0, 10e-7)
            if atime:
                rho = atime / btime
        eps = nodelta
        if sweep_log is not None:
            sweep_log.append(float(nodelta))
        cnt += 1
    return V, eps, cnt, rho


# --------------------------------------------------------------------------------------
# a6: beta-divergence and the MU exponent  (nn_fac/utils/beta_divergence.py:17-80)
# --------------------------------------------------------------------------------------
def gamma_beta(beta):
    """beta_divergence.py:75-80."""
    if beta < 1:
        return 1 / (2 - beta)
    if beta > 2:
        return 1 / (beta - 1)
    return 1


def beta_divergence(a, b, beta):
    """beta_divergence.py:42-52 (inputs must be strictly positive for beta in {0,1})."""
    if beta < 0:
        raise InvalidArgumentValue("Invalid value for beta: negative one.")
    if beta == 1:
        q = np.divide(a, b, where=(b != 0))
        return np.sum(a * np.log(q, where=(q != 0)) - a + b)
    if beta == 0:
        return np.sum(a / b - np.log(a / b, where=(a != 0)) - 1)
    return np.sum(1 / (beta * (beta - 1)) * (a ** beta + (beta - 1) * b ** beta - beta * a * (b ** (beta - 1))))


# --------------------------------------------------------------------------------------
# a5: multiplicative update  (nn_fac/update_rules/mu.py:20-97)
# --------------------------------------------------------------------------------------
def mu_betadivmin(U, V, M, beta):
    """mu.py:79-97.  No epsilon in any denominator; result clipped at 1e-12."""
    if beta < 0:
        raise InvalidArgumentValue("Invalid value for beta: negative one.")
    K = np.dot(U, V)
    if beta == 1:
        Kinv = K ** (-1)
        line = np.sum(V.T, axis=0)
        denom = np.array([line for _ in range(np.shape(K)[0])])
        return np.maximum(U * (np.dot((Kinv * M), V.T) / denom), EPS_MU)
    if beta == 2:
        denom = np.dot(K, V.T)
        return np.maximum(U * (np.dot(M, V.T) / denom), EPS_MU)
    if beta == 3:
        denom = np.dot(K ** 2, V.T)
        return np.maximum(U * (np.dot((K * M), V.T) / denom) ** gamma_beta(beta), EPS_MU)
    denom = np.dot(K ** (beta - 1), V.T)
    return np.maximum(U * (np.dot((K ** (beta - 2) * M), V.T) / denom) ** gamma_beta(beta), EPS_MU)


def switch_alternate_mu(data, U, V, beta, matrix):
    """mu.py:24-29."""
    if matrix in ("U", "W"):
        return mu_betadivmin(U, V, data, beta)
    if matrix in ("V", "H"):
        return np.transpose(mu_betadivmin(V.T, U.T, data.T, beta))
    raise InvalidArgumentValue(f"Invalid value for matrix: got {matrix}.")


# --------------------------------------------------------------------------------------
# a2/a3/a4: NMF driver  (nn_fac/nmf.py:19-458)
# --------------------------------------------------------------------------------------
def one_nmf_step(data, rank, U_in, V_in, norm_data, update_rule, beta,
                 sparsity_coefficients, fixed_modes, normalize, deterministic, sweeps=None):
    """nmf.py:387-458.  ``sweeps`` (oracle-only) collects cnt-1 of each hals call."""
    if update_rule not in ("hals", "mu"):
        raise InvalidArgumentValue(f"Invalid update rule: {update_rule}")
    if update_rule == "hals" and beta != 2:
        raise InvalidArgumentValue("hals is only valid for beta = 2.")
    if len(sparsity_coefficients) != 2:
        raise ValueError("NMF needs 2 sparsity coefficients to be performed")
    U, V = U_in.copy(), V_in.copy()
    a = math.inf if deterministic else 0.5

    if 0 not in fixed_modes:
        if update_rule == "hals":
            t = time.time()
            VVt = np.dot(V, np.transpose(V))
            VMt = np.dot(V, np.transpose(data))
            t = time.time() - t
            out = hals_nnls_acc(VMt, VVt, np.transpose(U_in), maxiter=100, atime=t, alpha=a, delta=0.01,
                                sparsity_coefficient=sparsity_coefficients[0], normalize=normalize[0],
                                nonzero=False)
            U = np.transpose(out[0])
            if sweeps is not None:
                sweeps.append(out[2] - 1)
        else:
            U = switch_alternate_mu(data, U, V, beta, "U")

    if 1 not in fixed_modes:
        if update_rule == "hals":
            t = time.time()
            UtU = np.dot(np.transpose(U), U)
            UtM = np.dot(np.transpose(U), data)
            t = time.time() - t
            out = hals_nnls_acc(UtM, UtU, V_in, maxiter=100, atime=t, alpha=a, delta=0.01,
                                sparsity_coefficient=sparsity_coefficients[1], normalize=normalize[1],
                                nonzero=False)
            V = out[0]
            if sweeps is not None:
                sweeps.append(out[2] - 1)
        else:
            V = switch_alternate_mu(data, U, V, beta, "V")

    sp = np.where(np.array(sparsity_coefficients) == None, 0, sparsity_coefficients)  # noqa: E711
    if update_rule == "hals":
        # matrix 1-norm (max column abs-sum), NOT entry-wise l1: nmf.py:452
        cost = np.linalg.norm(data - np.dot(U, V), ord='fro') ** 2 \
            + 2 * (sp[0] * np.linalg.norm(U, ord=1) + sp[1] * np.linalg.norm(V, ord=1))
    else:
        cost = beta_divergence(data, np.dot(U, V), beta)
    return U, V, cost


def compute_nmf(data, rank, U_in, V_in, n_iter_max=100, tol=1e-8, update_rule="hals", beta=2,
                sparsity_coefficients=[None, None], fixed_modes=[], normalize=[False, False],
                verbose=False, return_costs=False, deterministic=False, sweeps=None):
    """nmf.py:284-329."""
    U, V = U_in.copy(), V_in.copy()
    costs, toc = [], []
    norm_data = np.linalg.norm(data)
    tic = time.time()
    if sparsity_coefficients is None:
        sparsity_coefficients = [None, None]
    if fixed_modes is None:
        fixed_modes = []
    if normalize is None or normalize is False:
        normalize = [False, False]
    for it in range(n_iter_max):
        U, V, c = one_nmf_step(data, rank, U, V, norm_data, update_rule, beta, sparsity_coefficients,
                               fixed_modes, normalize, deterministic, sweeps=sweeps)
        toc.append(time.time() - tic)
        costs.append(c)
        if it > 0 and abs(costs[-2] - costs[-1]) < tol:
            break
    if return_costs:
        return np.array(U), np.array(V), costs, toc
    return np.array(U), np.array(V)


def nmf_random_init(shape, rank, seed):
    """initialize_factors.py:40-46 with deterministic=True: legacy global RandomState stream."""
    import random
    np.random.seed(seed)
    random.seed(seed)
    m, n = shape
    return np.random.rand(m, rank), np.random.rand(rank, n)


def nndsvd(V, rank):
    """f3: NNDSVD start values, initialize_factors.py:160-206 (Boutsidis & Gallopoulos).  The negativity test of the
    reference (`V.any() < 0`, :162) can never fire and is not restated.  np.linalg.svd(V) is the reference's call
    (full_matrices=True, :173); only the first `rank` triplets are read, so the thin SVD gives the same result, and the
    result does not depend on the sign convention of the singular vectors (flipping a pair swaps the two candidates)."""
    U, S, Et = np.linalg.svd(V, full_matrices=False)
    E = Et.T
    W = np.zeros((V.shape[0], rank))
    H = np.zeros((rank, V.shape[1]))
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
    H[0, :] = np.sqrt(S[0]) * np.abs(E[:, 0].T)
    for i in range(1, rank):
        uu, vv = U[:, i], E[:, i]
        uup, uun = np.multiply(uu >= 0, uu), np.multiply(uu < 0, -uu)
        vvp, vvn = np.multiply(vv >= 0, vv), np.multiply(vv < 0, -vv)
        n_uup, n_vvp = np.linalg.norm(uup, 2), np.linalg.norm(vvp, 2)
        n_uun, n_vvn = np.linalg.norm(uun, 2), np.linalg.norm(vvn, 2)
        termp, termn = n_uup * n_vvp, n_uun * n_vvn
        if termp >= termn:
            W[:, i] = np.sqrt(S[i] * termp) / n_uup * uup
            H[i, :] = np.sqrt(S[i] * termp) / n_vvp * vvp.T
        else:
            W[:, i] = np.sqrt(S[i] * termn) / n_uun * uun
            H[i, :] = np.sqrt(S[i] * termn) / n_vvn * vvn.T
    return np.maximum(W, 1e-12), np.maximum(H, 1e-12)


def ntf_nndsvd_init(tensor, rank):
    """initialize_factors.py:98-105: NNDSVD of every unfolding (modes shorter than the rank fall back to rand)."""
    out = []
    for mode in range(tensor.ndim):
        if tensor.shape[mode] < rank:
            out.append(np.random.rand(tensor.shape[mode], rank))
        else:
            out.append(nndsvd(unfold(tensor, mode), rank)[0])
    return out


def nmf(data, rank, init="random", U_0=None, V_0=None, n_iter_max=100, tol=1e-8, update_rule="hals", beta=2,
        sparsity_coefficients=[None, None], fixed_modes=[], normalize=[False, False], verbose=False,
        return_costs=False, deterministic=False, seed=0, sweeps=None):
    """nmf.py:175-193."""
    if min(data.shape) < rank:
        rank = min(data.shape)
    if deterministic:
        np.random.seed(seed)
    if init.lower() == "custom":
        if U_0 is None or V_0 is None:
            raise CustomNotValidFactors("Custom initialization, but (at least) one factor is set to 'None'")
    elif init.lower() == "random":
        if deterministic:
            U_0, V_0 = nmf_random_init(data.shape, rank, seed)
        else:
            U_0, V_0 = np.random.rand(data.shape[0], rank), np.random.rand(rank, data.shape[1])
    elif init.lower() == "nndsvd":
        U_0, V_0 = nndsvd(data, rank)
    else:
        raise InvalidInitializationType("Initialization type not understood.")
    return compute_nmf(data, rank, U_0, V_0, n_iter_max=n_iter_max, tol=tol, update_rule=update_rule, beta=beta,
                       sparsity_coefficients=sparsity_coefficients, fixed_modes=fixed_modes, normalize=normalize,
                       verbose=verbose, return_costs=return_costs, deterministic=deterministic, sweeps=sweeps)


# --------------------------------------------------------------------------------------
# f4: multilayer beta-NMF (nn_fac/multilayer_nmf.py:7-51) and its scaling (nn_fac/utils/normalize_wh.py:6-22)
# --------------------------------------------------------------------------------------
def normalize_WH(W, H, matrix):
    if matrix == "H":
        scalH = np.sum(H, axis=1)
        return W @ np.diag(scalH), np.diag(1 / scalH) @ H
    if matrix == "W":
        scalW = np.sum(W, axis=0)
        return W @ np.diag(1 / scalW), np.diag(scalW) @ H
    raise ValueError(f"Matrix must be either 'W' or 'H', but it is {matrix}")


def multilayer_beta_NMF(data, all_ranks, beta=1, n_iter_max_each_nmf=100, init_each_nmf="nndsvd", deterministic=False,
                        seed=0):
    """multilayer_nmf.py:7-44: layer 0 on the data, layer i on W[i-1]; each layer = nmf(mu, beta) + normalize_WH(.,"H")
    (:46-51).  Returns W, H, errors (L x n_iter_max_each_nmf)."""
    if deterministic:
        np.random.seed(seed)
    L = len(all_ranks)
    if sorted(all_ranks, reverse=True) != all_ranks:
        raise ValueError("The ranks of deep NMF should be decreasing.")
    W, H = [None] * L, [None] * L
    errors = np.empty((L, n_iter_max_each_nmf))
    cur = data
    for i in range(L):
        Wi, Hi, costs, _ = nmf(cur, all_ranks[i], init=init_each_nmf, n_iter_max=n_iter_max_each_nmf, tol=1e-8,
                               update_rule="mu", beta=beta, normalize=[False, True], return_costs=True,
                               deterministic=deterministic, seed=seed)
        W[i], H[i] = normalize_WH(Wi, Hi, "H")
        errors[i] = np.array(costs)
        cur = W[i]
    return W, H, errors


# --------------------------------------------------------------------------------------
# f4: deep KL-NMF  (nn_fac/deep_nmf.py:13-113, nn_fac/update_rules/deep_mu.py:8-14)
# --------------------------------------------------------------------------------------
def deep_KL_mu(W_Lm1, W_L, H_L, WH_Lp1, lambda_):
    """deep_mu.py:8-14 (scipy.special.lambertw, principal branch, real part; eps = 1e-12)."""
    from scipy.special import lambertw
    ONES = np.ones_like(W_Lm1)
    a = ONES @ H_L.T - lambda_ * np.log(WH_Lp1)
    b = W_L * ((W_Lm1 / (W_L @ H_L)) @ H_L.T)
    lambert = lambertw(b * np.exp(a / lambda_) / lambda_, k=0).real
    return np.maximum(1e-12, (1 / lambda_ * b) / (lambert + 1e-12))


def one_step_deep_KL_nmf(data, W, H, all_ranks, lambda_, delta):
    """deep_nmf.py:84-113."""
    L = len(all_ranks)
    errors = []
    for layer in range(L):
        D = data if layer == 0 else W[layer - 1]
        H[layer] = switch_alternate_mu(D, W[layer], H[layer], 1, "H")
        W[layer], H[layer] = normalize_WH(W[layer], H[layer], "H")
        if layer == L - 1:
            W[layer] = switch_alternate_mu(D, W[layer], H[layer], 1, "W")
        else:
            lam = lambda_[layer + 1] / lambda_[layer]
            W[layer] = deep_KL_mu(D, W[layer], H[layer], W[layer + 1] @ H[layer + 1], lam)
        errors.append(beta_divergence(D, W[layer] @ H[layer], 1))
    return W, H, errors


def deep_KL_NMF(data, all_ranks, n_iter_max_each_nmf=100, n_iter_max_deep_loop=100, init="multilayer_nmf",
                init_multi_layer="nndsvd", W_0=None, H_0=None, delta=1e-6, tol=1e-6, deterministic=False, seed=0):
    """deep_nmf.py:13-82.  Returns W, H, reconstruction_errors (L x (n_iter_max_deep_loop + 1), NaN where not reached)."""
    L = len(all_ranks)
    assert L > 1
    rec = np.full((L, n_iter_max_deep_loop + 1), np.nan)
    if sorted(all_ranks, reverse=True) != all_ranks:
        raise ValueError("The ranks of deep NMF should be decreasing.")
    if init == "multilayer_nmf":
        W, H, e = multilayer_beta_NMF(data, all_ranks, beta=1, n_iter_max_each_nmf=n_iter_max_each_nmf,
                                      init_each_nmf=init_multi_layer, deterministic=deterministic, seed=seed)
        rec[:, 0] = e[:, -1]
    elif init == "custom":
        W, H = [w.copy() for w in W_0], [h.copy() for h in H_0]
        rec[0, 0] = beta_divergence(data, W[0] @ H[0], 1)
        for i in range(1, L):
            rec[i, 0] = beta_divergence(W[i - 1], W[i] @ H[i], 1)
    else:
        raise ValueError("The init method is not supported.")
    lambda_ = 1 / np.array(rec[:, 0])
    glob = [lambda_.T @ rec[:, 0]]
    for it in range(n_iter_max_deep_loop):
        W, H, errors = one_step_deep_KL_nmf(data, W, H, all_ranks, lambda_, delta)
        rec[:, it + 1] = lambda_ * errors
        glob.append(lambda_.T @ errors)
        if it > 1 and abs(glob[-2] - glob[-1]) < tol:
            break
    return W, H, rec


# --------------------------------------------------------------------------------------
# tensorly 0.6.0 semantics used by ntf.py (published behaviour)
# --------------------------------------------------------------------------------------
def unfold(t, mode):
    return np.moveaxis(t, mode, 0).reshape(t.shape[mode], -1)


def fold(u, mode, shape):
    full = [shape[mode]] + [s for i, s in enumerate(shape) if i != mode]
    return np.moveaxis(u.reshape(full), 0, mode)


def khatri_rao(mats, skip_matrix=None):
    ms = [m for i, m in enumerate(mats) if i != skip_matrix]
    res = ms[0]
    R = res.shape[1]
    for M in ms[1:]:
        res = (res[:, None, :] * M[None, :, :]).reshape(-1, R)
    return res


def mode_dot(t, M, mode, transpose=False):
    if transpose:
        M = np.conj(M.T)
    shape = list(t.shape)
    shape[mode] = M.shape[0]
    return fold(M @ unfold(t, mode), mode, shape)


def multi_mode_dot(t, mats, skip=None, transpose=False):
    out = t
    for i, M in enumerate(mats):
        if i == skip:
            continue
        out = mode_dot(out, M, i, transpose=transpose)
    return out


# --------------------------------------------------------------------------------------
# a7: NTF driver  (nn_fac/ntf.py:201-477)
# --------------------------------------------------------------------------------------
def one_ntf_step(unfolded_tensors, rank, in_factors, norm_tensor, update_rule, beta,
                 sparsity_coefficients, fixed_modes, normalize, alpha=0.5, delta=0.01, sweeps=None):
    """ntf.py:422-477.  Cost reuses the loop-leaked mode/rhs/krao of the last updated mode."""
    if update_rule not in ("hals", "mu"):
        raise InvalidArgumentValue(f"Invalid update rule: {update_rule}")
    if update_rule == "hals" and beta != 2:
        raise InvalidArgumentValue("hals is only valid for beta = 2.")
    for f in fixed_modes:
        sparsity_coefficients[f] = None
    factors = in_factors.copy()
    modes = [m for m in range(len(unfolded_tensors)) if m not in fixed_modes]
    for mode in modes:
        if update_rule == "hals":
            t = time.time()
            cross = np.ones((rank, rank))
            for i, f in enumerate(factors):
                if i != mode:
                    cross *= np.dot(np.transpose(f), f)
            krao = khatri_rao(factors, skip_matrix=mode)
            rhs = np.dot(unfolded_tensors[mode], krao)
            t = time.time() - t
            out = hals_nnls_acc(np.transpose(rhs), cross, np.transpose(factors[mode]), maxiter=100, atime=t,
                                alpha=alpha, delta=delta, sparsity_coefficient=sparsity_coefficients[mode],
                                normalize=normalize[mode])
            factors[mode] = np.transpose(out[0])
            if sweeps is not None:
                sweeps.append(out[2] - 1)
        else:
            krao = khatri_rao(factors, skip_matrix=mode)
            factors[mode] = mu_betadivmin(factors[mode], krao.T, unfolded_tensors[mode], beta)
    sparsity_error = 0
    for idx, s in enumerate(sparsity_coefficients):
        if s:
            sparsity_error += 2 * (s * np.linalg.norm(factors[idx], ord=1))
    if update_rule == "hals":
        rec = norm_tensor ** 2 - 2 * np.dot(factors[mode].reshape(-1), rhs.reshape(-1)) \
            + np.sqrt(np.sum(np.abs(np.dot(factors[mode], np.transpose(krao))) ** 2)) ** 2
    else:
        rec = beta_divergence(unfolded_tensors[mode], factors[mode] @ krao.T, beta)
    return factors, (rec + sparsity_error) / (norm_tensor ** 2)


def compute_ntf(tensor_in, rank, factors_in, n_iter_max=100, tol=1e-8, update_rule="hals", beta=2,
                sparsity_coefficients=[], fixed_modes=[], normalize=[], verbose=False, return_costs=False,
                alpha=0.5, delta=0.01, sweeps=None):
    """ntf.py:288-344.  ``alpha``/``delta`` are oracle-only pass-throughs to one_ntf_step
    (the reference never overrides them: ntf.py:316-317, so its HALS path is wall-clock dependent)."""
    factors = list(factors_in).copy()
    tensor = tensor_in.copy()
    norm_tensor = np.sqrt(np.sum(np.abs(tensor) ** 2))
    nb = tensor.ndim
    if sparsity_coefficients is None or len(sparsity_coefficients) != nb:
        sparsity_coefficients = [None] * nb
    if fixed_modes is None:
        fixed_modes = []
    if normalize is None or len(normalize) != nb:
        normalize = [False] * nb
    costs, toc = [], []
    tic = time.time()
    unf = [unfold(tensor, m) for m in range(nb)]
    for it in range(n_iter_max):
        factors, c = one_ntf_step(unf, rank, factors, norm_tensor, update_rule, beta, sparsity_coefficients,
                                  fixed_modes, normalize, alpha=alpha, delta=delta, sweeps=sweeps)
        toc.append(time.time() - tic)
        costs.append(c)
        if it > 0 and abs(costs[-2] - costs[-1]) < tol:
            break
    if return_costs:
        return factors, costs, toc
    return factors


# --------------------------------------------------------------------------------------
# a8: NTD  (nn_fac/ntd.py:248-698, nn_fac/update_rules/mu.py:99-159)
# --------------------------------------------------------------------------------------
class InvalidRanksException(ArgumentException):
    """errors.py: raised when len(ranks) != tensor order (ntd.py:213)."""


class CustomNotValidCore(ArgumentException):
    """errors.py: custom init with core_0 None (ntd.py:234)."""


def contract(a, modes_a, b, modes_b):
    """tensorly 0.6.0 tl.tenalg.contract == tensordot over the listed modes."""
    return np.tensordot(a, b, axes=(modes_a, modes_b))


def random_tucker_full(shape, rank, seed):
    """tl.random.random_tucker(shape, rank, full=True, random_state=seed): factors drawn first,
    in mode order, then the core, then the full tensor."""
    rs = np.random.RandomState(seed)
    factors = [rs.random_sample((s, r)) for s, r in zip(shape, rank)]
    core = rs.random_sample(tuple(rank))
    return multi_mode_dot(core, factors)


def tucker_hooi(tensor, ranks, n_iter_max=100, tol=10e-5):
    """tensorly.decomposition.tucker(tensor, ranks) of tensorly 0.6.0 (setup.py:30; THIRD-PARTY, not under /root/reference --
    its published algorithm restated): higher-order SVD start (the leading left singular vectors of every unfolding), then
    HOOI sweeps -- factor n := leading left singular vectors of unfold(tensor x_{i != n} F_i^T, n) -- until the relative
    reconstruction error sqrt(|  ||T||^2 - ||core||^2  |) / ||T|| changes by less than tol = 10e-5 (checked from the third
    sweep on) or n_iter_max = 100.  Singular vectors are defined up to sign; the only caller (ntd_initialization, 'tucker')
    takes absolute values of core and factors, which removes the ambiguity.
    PINNED by the reference's own known answers for init="tucker" (tests/NTD_tests.py:157-175 HALS, :197-215 MU beta=2) through
    ntd_tucker_init + compute_ntd: tests/test_oracle_golden.py::test_tucker_init_known_answers (agreement ~1e-13)."""
    N = tensor.ndim

    def leading(M, k):
        U, _, _ = np.linalg.svd(M, full_matrices=False)
        return U[:, :k]
    factors = [leading(unfold(tensor, m), ranks[m]) for m in range(N)]
    norm_t = np.sqrt(np.sum(tensor ** 2))
    errs = []
    core = None
    for it in range(n_iter_max):
        for m in range(N):
            approx = multi_mode_dot(tensor, factors, skip=m, transpose=True)
            factors[m] = leading(unfold(approx, m), ranks[m])
        core = multi_mode_dot(tensor, factors, transpose=True)
        errs.append(np.sqrt(abs(norm_t ** 2 - np.sum(core ** 2))) / norm_t)
        if it > 1 and abs(errs[-2] - errs[-1]) < tol:
            break
    return core, factors


def ntd_tucker_init(tensor, ranks):
    """initialize_factors.py:68-75 ('tucker'): |core| + 1e-12, |factors| + 1e-12 of tl_tucker(tensor, ranks)."""
    core, factors = tucker_hooi(tensor, list(ranks))
    return np.abs(core) + 1e-12, [np.abs(f) + 1e-12 for f in factors]


def ntd_random_init(shape, ranks, seed):
    """initialize_factors.py:53-66 with deterministic=True (np.random.seed(seed); random.seed has no effect here)."""
    np.random.seed(seed)
    factors = []
    for mode in range(len(shape)):
        f = np.random.rand(shape[mode], ranks[mode])
        f[f < 1e-12] = 1e-12
        factors.append(f)
    core = np.random.rand(int(np.prod(ranks))).reshape(tuple(ranks))
    core[core < 1e-12] = 1e-12
    return core, factors


def one_ntd_step(tensor, ranks, in_core, in_factors, norm_tensor, sparsity_coefficients, fixed_modes, normalize,
                 mode_core_norm, alpha=0.5, delta=0.01, sweeps=None, pg_iters=None):
    """ntd.py:514-645.  The core update reuses `temp` and `elemprod` left behind by the LAST updated mode (:581-583).
    sweeps / pg_iters (not in the reference) collect the inner sweep counts and the projected-gradient iteration count."""
    for f in fixed_modes:
        sparsity_coefficients[f] = None
    core = in_core.copy()
    factors = in_factors.copy()
    nd = tensor.ndim
    modes_list = [m for m in range(nd) if m not in fixed_modes]
    for mode in modes_list:
        elemprod = factors.copy()                                                   # :534-537
        for i, factor in enumerate(factors):
            if i != mode:
                elemprod[i] = np.dot(np.conj(np.transpose(factor)), factor)
        temp = multi_mode_dot(core, elemprod, skip=mode)                            # :539
        con_modes = [i for i in range(nd) if i != mode]
        UtU = contract(temp, con_modes, core, con_modes)                            # :544
        temp = multi_mode_dot(tensor, factors, skip=mode, transpose=True)           # :550
        MtU = contract(temp, con_modes, core, con_modes)                            # :555
        UtM = np.transpose(MtU)
        V, eps, cnt, rho = hals_nnls_acc(UtM, UtU, np.transpose(factors[mode]), maxiter=100, atime=None, alpha=alpha,
                                         delta=delta, sparsity_coefficient=sparsity_coefficients[mode],
                                         normalize=normalize[mode])                 # :571-573 (atime only feeds rho)
        if sweeps is not None:
            sweeps.append(cnt - 1)
        factors[mode] = np.transpose(V)
    last = modes_list[-1]
    all_MtX = mode_dot(temp, np.transpose(factors[last]), last)                     # :581
    all_MtM = elemprod.copy()
    all_MtM[last] = factors[last].T @ factors[last]
    gradient_step = 1                                                               # :588-596
    for MtM in all_MtM:
        gradient_step *= 1 / np.linalg.svd(MtM, compute_uv=False)[0]                # svds(MtM, k=1)[1][0]
    gradient_step = round(gradient_step, 6)
    cnt, upd_0, upd = 1, 0, 1
    sparse = 0 if sparsity_coefficients[-1] is None else sparsity_coefficients[-1]
    while cnt <= 300 and upd >= delta * upd_0:                                      # :609-619
        gradient = -all_MtX + multi_mode_dot(core, all_MtM, transpose=False) + sparse * np.ones(core.shape)
        delta_core = np.minimum(gradient_step * gradient, core)
        core = core - delta_core
        upd = np.sqrt(np.sum(delta_core ** 2))
        if cnt == 1:
            upd_0 = upd
        cnt += 1
    if pg_iters is not None:
        pg_iters.append(cnt - 1)
    if normalize[-1]:                                                               # :621-626
        unfolded_core = unfold(core, mode_core_norm).copy()
        for idx in range(unfolded_core.shape[0]):
            nrm = np.sqrt(np.sum(unfolded_core[idx] ** 2))
            if nrm != 0:
                unfolded_core[idx] = unfolded_core[idx] / nrm
        core = fold(unfolded_core, mode_core_norm, core.shape)
    sparsity_error = 0                                                              # :629-637
    for index, sp in enumerate(sparsity_coefficients):
        if sp:
            if index < len(factors):
                sparsity_error += 2 * (sp * np.linalg.norm(factors[index], ord=1))
            elif index == len(factors):
                sparsity_error += 2 * (sp * np.sum(np.abs(core)))
            else:
                raise NotImplementedError("Too many sparsity coefficients")
    rec_error = norm_tensor ** 2 - 2 * np.sum(all_MtX * core) + np.sum(multi_mode_dot(core, all_MtM, transpose=False) * core)
    return core, factors, (rec_error + sparsity_error) / (norm_tensor ** 2)        # :639-640


def mu_tensorial(G, factors, tensor, beta):
    """mu.py:138-159 (core update of NTD-MU)."""
    if beta < 0:
        raise InvalidArgumentValue("Invalid value for beta: negative one.")
    K = multi_mode_dot(G, factors)
    if beta == 1:
        L1, L2 = np.ones(np.shape(K)), K ** (-1) * tensor
    elif beta == 2:
        L1, L2 = K, np.ones(np.shape(K)) * tensor
    elif beta == 3:
        L1, L2 = K ** 2, K * tensor
    else:
        L1, L2 = K ** (beta - 1), K ** (beta - 2) * tensor
    ft = [f.T for f in factors]
    return np.maximum(G * (multi_mode_dot(L2, ft) / multi_mode_dot(L1, ft)) ** gamma_beta(beta), 1e-12)


def one_ntd_step_mu(tensor, ranks, in_core, in_factors, beta, norm_tensor, fixed_modes, normalize, mode_core_norm):
    """ntd.py:664-698.  The cost is NOT normalised (:696)."""
    core = in_core.copy()
    factors = in_factors.copy()
    modes_list = [m for m in range(tensor.ndim) if m not in fixed_modes]
    for mode in modes_list:
        factors[mode] = mu_betadivmin(factors[mode], unfold(multi_mode_dot(core, factors, skip=mode), mode),
                                      unfold(tensor, mode), beta)
    core = mu_tensorial(core, factors, tensor, beta)
    if normalize[-1]:
        unfolded_core = unfold(core, mode_core_norm).copy()
        for idx in range(unfolded_core.shape[0]):
            nrm = np.sqrt(np.sum(unfolded_core[idx] ** 2))
            if nrm != 0:
                unfolded_core[idx] = unfolded_core[idx] / nrm
        core = fold(unfolded_core, mode_core_norm, core.shape)
    return core, factors, beta_divergence(tensor, multi_mode_dot(core, factors), beta)


def compute_ntd(tensor_in, ranks, core_in, factors_in, n_iter_max=100, tol=1e-6, update_rule="hals", beta=2,
                sparsity_coefficients=[], fixed_modes=[], normalize=[], mode_core_norm=None, return_costs=False,
                deterministic=False, sweeps=None, pg_iters=None):
    """ntd.py:355-433 (verbose printing dropped)."""
    core = core_in.copy()
    factors = factors_in.copy()
    tensor = tensor_in
    norm_tensor = np.sqrt(np.sum(np.abs(tensor) ** 2))
    nb_modes = tensor.ndim
    if sparsity_coefficients is None or len(sparsity_coefficients) != nb_modes + 1:
        sparsity_coefficients = [None for _ in range(nb_modes + 1)]
    if fixed_modes is None:
        fixed_modes = []
    if normalize is None or len(normalize) != nb_modes + 1:
        normalize = [False for _ in range(nb_modes + 1)]
    if normalize[-1] and (mode_core_norm is None or mode_core_norm < 0 or mode_core_norm >= nb_modes):
        normalize[-1] = False
    costs, toc = [], []
    for iteration in range(n_iter_max):
        if update_rule == "hals":
            core, factors, cost = one_ntd_step(tensor, ranks, core, factors, norm_tensor, sparsity_coefficients,
                                               fixed_modes, normalize, mode_core_norm,
                                               alpha=math.inf if deterministic else 0.5, sweeps=sweeps,
                                               pg_iters=pg_iters)
        elif update_rule == "mu":
            core, factors, cost = one_ntd_step_mu(tensor, ranks, core, factors, beta, norm_tensor, fixed_modes,
                                                  normalize, mode_core_norm)
        else:
            raise InvalidArgumentValue(f"The update rule provided is not valid (Got {update_rule}).")
        toc.append(0.0)
        costs.append(cost)
        if iteration > 0 and abs(costs[-2] - costs[-1]) < tol:
            break
    if return_costs:
        return core, factors, costs, toc
    return core, factors


# --------------------------------------------------------------------------------------
# synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------------------
def synth_nmf(m, n, r, seed=0, dtype=np.float32):
    """X = W*H* + 1e-2*rand, plus inits U0, V0, all from one legacy RandomState(seed) stream."""
    rng = np.random.RandomState(seed)
    W, H = rng.rand(m, r), rng.rand(r, n)
    X = (W @ H + 1e-2 * rng.rand(m, n)).astype(dtype)
    U0 = rng.rand(m, r).astype(dtype)
    V0 = rng.rand(r, n).astype(dtype)
    return X, U0, V0


def synth_ntf(shape, R, seed=0, dtype=np.float32):
    rng = np.random.RandomState(seed)
    gen = [rng.rand(s, R) for s in shape]
    T = np.einsum('ir,jr,kr->ijk', *gen) + 1e-2 * rng.rand(*shape)
    F0 = [rng.rand(s, R).astype(dtype) for s in shape]
    return T.astype(dtype), F0
