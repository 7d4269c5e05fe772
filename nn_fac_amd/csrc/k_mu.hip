// beta-divergence multiplicative updates (nn_fac/update_rules/mu.py:79-97) fused with the product K = U V.
//
//   left  (mu_betadivmin on U):   Ut_new[k,i] = max(Ut[k,i] * (num[k,i]/den[k,i])^gamma, 1e-12)
//            num = ((K^(b-2) .* X) V^T)^T,  den = (K^(b-1) V^T)^T          (b = 1: den[k] = rowsum(V)[k])
//   right (switch_alternate_mu "V"): V_new[k,j] = max(V[k,j] * (num/den)^gamma, 1e-12)
//            num = U^T (K^(b-2) .* X),      den = U^T K^(b-1)              (b = 1: den[k] = colsum(U)[k])
//   b = 2 needs no K at all: num = X V^T (xht) / U^T X (xty), den = (V V^T) U^T / (U^T U) V  (Gram form).
//
// One pass over X per update.  Per 16 x 64 (right) or 64 x 16 (left) block of X a wave runs
//   MFMA #1 : P = U V tile, k over the rank            (operands in LDS: "F_K" image + resident fragments)
//   VALU    : R = K^(b-2) .* X  [and K^(b-1)] in the accumulator layout (masked outside the matrix)
//   MFMA #2 : num (+den) += factor-fragment * R, k over the block's rows (right) / columns (left): the accumulator
//             tile of MFMA #1 is used AS the B operand of MFMA #2 with no lane movement, because P is computed in the
//             orientation whose row index is the contracted one (cdna_hip_programming.md s.3, "accumulator as operand").
// X goes HBM -> VGPR once, 16 bytes per lane; K is never written anywhere.
// Built for r <= 64 (MT <= 4); larger ranks return NNF_ERR_UNSUPPORTED for beta != 2.
#include "k_mu_kernels.h"

// One source, three translation units (Makefile: -DMU_PART=0|1|2) so that the ~60 instantiations of the two fused kernels
// compile side by side: 0 = right update + the small helpers, 1 = left update (beta = 1 and general beta), 2 = the left
// kernel's cost-carrying forms (KL cost, NTF cost + partial product).  The small kernels and launch templates above the
// entry points are `static`: every unit sees them, only the units that use them emit them.
#ifndef MU_PART
#error "k_mu.hip is compiled per part: -DMU_PART=0|1|2"
#endif
NNF_BUILD_FLAGS(NNF_CAT(k_mu, MU_PART), "MU_WG_PER_CU=" NNF_STR(MU_WG_PER_CU) " MU_STEP_FENCE()=" NNF_STR(MU_STEP_FENCE()))

// F_new = max(F * (num/den)^gamma, 1e-12); num/den summed over slabs in fp64 (fixed order); den_vec: per-row denominator (KL)
static __global__ __launch_bounds__(256) void nnf_mu_finish_kernel(const float* __restrict__ F, int64_t ldf, int r, int64_t cols,
                                                            const float* __restrict__ snum, const float* __restrict__ sden,
                                                            int nslab, int64_t slab_stride, int64_t lds,
                                                            const double* __restrict__ den_vec, float gamma,
                                                            float* __restrict__ out, int64_t ldo) {
    const int64_t total = (int64_t)r * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / cols, j = e - k * cols;
        double nu = 0.0, de = 0.0;
        // slabs eight at a time (loads of a batch in flight together, sums in slab order): one load per trip is a memory
        // round trip per slab -- ~100 of them in a row behind the right update of config C
        auto slab_sum = [&](const float* __restrict__ base) {
            const float* p = base + k * lds + j;
            double acc = 0.0;
            for (int s = 0; s < nslab; s += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(s + u < nslab ? s + u : nslab - 1) * slab_stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (s + u < nslab) ? (double)v[u] : 0.0;
            }
            return acc;
        };
        nu = slab_sum(snum);
        if (den_vec) de = den_vec[k];
        else de = slab_sum(sden);
        float ratio = (float)(nu / de);
        if (gamma != 1.f) ratio = powf(ratio, gamma);
        out[k * ldo + j] = fmaxf(F[k * ldf + j] * ratio, 1e-12f);
    }
}

// out[k] = sum_j A[k][j]  (fp64).  Long rows (the r x m factor: one workgroup per row took 112 us at m = 100000) are cut
// into gridDim.y pieces whose partial sums (part[k][piece]) are added in piece order by a second, tiny launch.
static __global__ __launch_bounds__(256) void nnf_rowsum_kernel(const float* __restrict__ A, int64_t lda, int64_t K,
                                                         double* __restrict__ out, int64_t per) {
    __shared__ double red[4];
    const float* p = A + (int64_t)blockIdx.x * lda;
    const int64_t j0 = (int64_t)blockIdx.y * per, j1 = (j0 + per < K) ? (j0 + per) : K;
    double s = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 8 * 256) {   // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[j + 256 * u < j1 ? j + 256 * u : j1 - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (j + 256 * u < j1) ? (double)v[u] : 0.0;
    }
    const double t = nnf_block_sum_f64(s, red);
    if (threadIdx.x == 0) out[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = t;
}
static __global__ __launch_bounds__(64) void nnf_rowsum_fin_kernel(const double* __restrict__ part, int np, int r,
                                                            double* __restrict__ out) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= r) return;
    double s = 0.0;
    for (int i = 0; i < np; ++i) s += part[(int64_t)k * np + i];
    out[k] = s;
}
// rowsum of an r x K matrix into out[r] on stream st; scratch from the cursor only when the rows are long
static int nnf_launch_rowsum(nnf_ws_cursor& cur, const float* A, int64_t lda, int r, int64_t K, double* out, hipStream_t st) {
    int np = (int)(K / 8192);
    if (np > 64) np = 64;
    if (np <= 1) {
        hipLaunchKernelGGL(nnf_rowsum_kernel, dim3(r, 1), dim3(256), 0, st, A, lda, K, out, K);
        NNF_CHECK_LAUNCH();
        return NNF_OK;
    }
    double* part = (double*)cur.take((size_t)r * np * 8);
    if (!part) return NNF_ERR_WORKSPACE;
    hipLaunchKernelGGL(nnf_rowsum_kernel, dim3(r, np), dim3(256), 0, st, A, lda, K, part, nnf_cdiv(K, (int64_t)np));
    NNF_CHECK_LAUNCH();
    hipLaunchKernelGGL(nnf_rowsum_fin_kernel, dim3((r + 63) / 64), dim3(64), 0, st, part, np, r, out);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}


// beta = 2 (Gram form): out[k][j] = max(F[k][j] * num[k][j] / (sum_l G[k][l] F[l][j]), 1e-12)
static __global__ __launch_bounds__(256) void nnf_mu2_finish_kernel(const float* __restrict__ F, int64_t ldf, int r, int64_t cols,
                                                             const float* __restrict__ G, const float* __restrict__ num,
                                                             int64_t ldn, float* __restrict__ out, int64_t ldo) {
    __shared__ float Gs[NNF_MAX_RANK * NNF_MAX_RANK];
    for (int e = threadIdx.x; e < r * r; e += 256) Gs[e] = G[e];
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < cols; j += (int64_t)gridDim.x * 256) {
        for (int k = 0; k < r; ++k) {
            float d = 0.f;
            for (int l = 0; l < r; l += 8) {   // eight factor entries in flight per trip, same order of the sum
                float fv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) fv[u] = F[(int64_t)(l + u < r ? l + u : r - 1) * ldf + j];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (l + u < r) d = fmaf(Gs[k * r + l + u], fv[u], d);
            }
            out[(int64_t)k * ldo + j] = fmaxf(F[(int64_t)k * ldf + j] * (num[(int64_t)k * ldn + j] / d), 1e-12f);
        }
    }
}

static float gamma_of(double beta) { return beta < 1.0 ? (float)(1.0 / (2.0 - beta)) : (beta > 2.0 ? (float)(1.0 / (beta - 1.0)) : 1.f); }

static size_t mu_shm(int MT, int REM, int r, bool regf) {   // two double-buffered chunk images (+ the resident fragments unless in registers)
    return ((regf ? 0 : (size_t)4 * ((r + 3) / 4) * 64) + (size_t)2 * (2 * MT + (REM > 0 ? 1 : 0)) * 256) * 16;
}

// leftover ranks handled on the VALU pipe: up to 4 next to one MFMA tile (ranks 17..20), up to 2 next to two or three (33, 34,
// 49, 50) -- four leftover ranks at two tiles left scratch reloads inside the right kernel's chunk loop, at three tiles hipcc
// spilled hundreds of registers (256 per wave at two workgroups per CU)
#define MU_REM_OF(q, rem) ((rem) <= 2 ? 2 : ((q) == 1 && (rem) <= 4 ? 4 : 0))
// Rank split of the fused kernels: MT full 16-rank tiles on MFMA, plus -- for 16q+1 .. 16q+4 ranks, aligned X, not the
// general-beta form -- the leftover ranks on the VALU pipe (REM = 2 or 4) instead of a padded tile.
static inline void mu_split_rank(int r, bool rem_ok, int& MT, int& REM) {
    const int q = r / 16, rem = r % 16;
    if (rem_ok && q >= 1 && q <= 3 && rem >= 1 && MU_REM_OF(q, rem) > 0) { MT = q; REM = MU_REM_OF(q, rem); }
    else { MT = (r + 15) / 16; REM = 0; }
}
// FN<MT, REM, BM, VEC>(args) over the instantiated (MT, REM, VEC) combinations; BMV without leftover-rank forms: REMOK = false
#define MU_CALL(FN, BMV, REMOK, R, VECF, ...)                                                                \
    do {                                                                                                     \
        int MT_, REM_;                                                                                       \
        mu_split_rank((R), (REMOK) && (VECF), MT_, REM_);                                                    \
        if (REM_ == 2) switch (MT_) {                                                                        \
            case 1: return FN<1, (REMOK) ? 2 : 0, BMV, true>(__VA_ARGS__);                                   \
            case 2: return FN<2, (REMOK) ? 2 : 0, BMV, true>(__VA_ARGS__);                                   \
            default: return FN<3, (REMOK) ? 2 : 0, BMV, true>(__VA_ARGS__);                                  \
        }                                                                                                    \
        if (REM_ == 4) return FN<1, (REMOK) ? 4 : 0, BMV, true>(__VA_ARGS__);                                \
        switch (MT_) {                                                                                       \
            case 1: return (VECF) ? FN<1, 0, BMV, true>(__VA_ARGS__) : FN<1, 0, BMV, false>(__VA_ARGS__);    \
            case 2: return (VECF) ? FN<2, 0, BMV, true>(__VA_ARGS__) : FN<2, 0, BMV, false>(__VA_ARGS__);    \
            case 3: return (VECF) ? FN<3, 0, BMV, true>(__VA_ARGS__) : FN<3, 0, BMV, false>(__VA_ARGS__);    \
            case 4: return (VECF) ? FN<4, 0, BMV, true>(__VA_ARGS__) : FN<4, 0, BMV, false>(__VA_ARGS__);    \
            default: return NNF_ERR_UNSUPPORTED;                                                             \
        }                                                                                                    \
    } while (0)

template <int MT, int REM, int BM, bool VEC>
static int launch_mu_right(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx,
                           const float* Ut, int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* V_out,
                           int64_t ldvo, hipStream_t st, float* num_out = nullptr, int64_t ldnum = 0,
                           float* den_out = nullptr, int64_t ldden = 0, double* den_vec_out = nullptr) {
    if ((int64_t)(16 * (MT + 1)) * ldu * 4 + 4 * (m + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;   // 32-bit image offsets
    const int ncb = (int)nnf_cdiv(n, 256);
    const int64_t ldp = nnf_rup(n, 4);
    const int nacc = (BM == BM_GEN) ? 2 : 1;
    int64_t nsplit = (BM == BM_KL ? 2 : 1) * (int64_t)ctx->num_cus / ncb;   // resident 4-wave workgroups per CU
    if (nsplit < 1) nsplit = 1;
    const int64_t max_split = nnf_cdiv(m, 64);
    if (nsplit > max_split) nsplit = max_split;
    const int64_t slab_elems = (int64_t)r * ldp;
    double* dvec = (double*)cur.take((size_t)r * 8);
    if (!dvec) return NNF_ERR_WORKSPACE;
    if (BM == BM_KL) {  // den[k] = colsum(U)[k] = rowsum(Ut)[k]   (mu.py:86-87 on the transposed problem)
        const int rc = nnf_launch_rowsum(cur, Ut, ldu, r, m, num_out ? den_vec_out : dvec, st);
        if (rc != NNF_OK) return rc;
    }
    const int64_t ws_max = (int64_t)(cur.remaining() / 4) / (slab_elems * nacc);
    if (ws_max < 1) return NNF_ERR_WORKSPACE;
    if (nsplit > ws_max) nsplit = ws_max;
    int64_t rps = nnf_rup(nnf_cdiv(m, nsplit), 64);
    while ((rps + 128) * ldx * 4 >= (int64_t)0x7fff0000) {
        if (rps <= 64) return NNF_ERR_UNSUPPORTED;
        rps = nnf_rup(rps / 2, 64);
    }
    nsplit = nnf_cdiv(m, rps);
    if (nsplit > ws_max) return NNF_ERR_WORKSPACE;
    float* snum = (float*)cur.take((size_t)nsplit * slab_elems * 4);
    float* sden = nacc == 2 ? (float*)cur.take((size_t)nsplit * slab_elems * 4) : nullptr;
    if (!snum || (nacc == 2 && !sden)) return NNF_ERR_WORKSPACE;
    const int a_vec_ok = ((((uintptr_t)Ut) & 15) == 0 && (ldu & 3) == 0) ? 1 : 0;
    const size_t shm = mu_shm(MT, REM, r, BM == BM_KL);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_mu_right_kernel<MT, REM, BM, VEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const int grid = 8 * (int)nnf_cdiv(nsplit, 8) * ncb;
    nnf_probe(ctx, NNF_PROBE_MU_RIGHT, 0, st);
    hipLaunchKernelGGL((nnf_mu_right_kernel<MT, REM, BM, VEC>), dim3(grid), dim3(256), shm, st, X, m, n, ldx, Ut, ldu, V, ldv, r,
                       (float)beta, snum, sden, ldp, ncb, (int)nsplit, rps, a_vec_ok);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_MU_RIGHT, 1, st);
    if (num_out) {   // accumulate only (row-sharded runs): this block's numerator / denominator, slab-reduced in fixed order
        int rc = nnf_launch_reduce_slabs(snum, (int)nsplit, slab_elems, r, n, ldp, num_out, ldnum, st);
        if (rc != NNF_OK) return rc;
        if (nacc == 2) return nnf_launch_reduce_slabs(sden, (int)nsplit, slab_elems, r, n, ldp, den_out, ldden, st);
        return NNF_OK;
    }
    int64_t fg = nnf_cdiv((int64_t)r * n, 256);
    if (fg > 2048) fg = 2048;
    hipLaunchKernelGGL(nnf_mu_finish_kernel, dim3((int)fg), dim3(256), 0, st, V, ldv, r, n, snum, sden, (int)nsplit,
                       slab_elems, ldp, BM == BM_KL ? dvec : (const double*)nullptr, gamma_of(beta), V_out, ldvo);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

template <int MT, int REM, int BM, bool VEC>
static int launch_mu_left(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx,
                          const float* Ut, int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* Ut_out,
                          int64_t lduo, hipStream_t st, int raw_num = 0, mu_left_extra ex = mu_left_extra{nullptr, 0, 1, nullptr},
                          double* cost_out = nullptr) {
    if (64 * ldx * 4 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    if ((int64_t)(16 * (MT + 1)) * ldv * 4 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;   // 32-bit image offsets
    double* dvec = (double*)cur.take((size_t)r * 8);
    if (!dvec) return NNF_ERR_WORKSPACE;
    const int a_vec_ok = ((((uintptr_t)V) & 15) == 0 && (ldv & 3) == 0) ? 1 : 0;
    const size_t shm = mu_shm(MT, REM, r, BM != BM_GEN);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_mu_left_kernel<MT, REM, BM, VEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (BM == BM_KL || BM == BM_KLC) {  // den[k] = rowsum(V)[k]   (mu.py:86-87)
        const int rc = nnf_launch_rowsum(cur, V, ldv, r, n, dvec, st);
        if (rc != NNF_OK) return rc;
    }
    // Rows per workgroup (4 waves x 4, 3 or 2 tiles of 16 rows): whole ROUNDS of resident workgroups, all of about the same
    // length -- R = ceil(T / (16 slots)) rounds of `slots` workgroups, each 8 to 16 tiles, as a mix of two adjacent sizes.
    // 256-row workgroups everywhere put 977 workgroups on the 768 slots of the 250000-row pass of config D: a second round
    // that is 27 % full and as long as the first.  Less than one round of 128-row workgroups: 128 rows each (most CUs busy).
    const int64_t slots = (int64_t)MU_LEFT_WGPC(MT, REM, BM) * ctx->num_cus, T = nnf_cdiv(m, 16);
    const int64_t W = nnf_cdiv(T, 16 * slots) * slots;
    int64_t n_hi = 0, n_mid = 0, grid = W;
    if (T <= 8 * slots) {
        grid = nnf_cdiv(m, 128);
    } else if (T > 12 * W) {
        n_hi = nnf_cdiv(T - 12 * W, 4);
        n_mid = W - n_hi;
    } else {
        n_mid = nnf_cdiv(T - 8 * W, 4);
    }
    if (n_hi * 256 + n_mid * 192 + (grid - n_hi - n_mid) * 128 < m) return NNF_ERR_UNSUPPORTED;   // (cannot happen)
    if (BM == BM_FROB || BM == BM_KLC) {
        ex.partial = (double*)cur.take((size_t)grid * 8);
        if (!ex.partial || !cost_out) return NNF_ERR_WORKSPACE;
    }
    nnf_probe(ctx, NNF_PROBE_MU_LEFT, 0, st);
    hipLaunchKernelGGL((nnf_mu_left_kernel<MT, REM, BM, VEC>), dim3((int)grid), dim3(256), shm, st, X, m, n, ldx, Ut, ldu, V, ldv, r,
                       (float)beta, dvec, raw_num ? -1.f : gamma_of(beta), Ut_out, lduo, a_vec_ok, (int)n_hi, (int)n_mid, ex);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_MU_LEFT, 1, st);
    if (BM == BM_FROB || BM == BM_KLC) return nnf_launch_sum_f64(ex.partial, grid, 1.0, cost_out, st);
    return NNF_OK;
}

#define MU_DISPATCH(FN, ...)                                                                              \
    do {                                                                                                  \
        const bool vec = x_vec_ok(X, ldx);                                                                \
        if (beta == 1.0) MU_CALL(FN, BM_KL, true, r, vec, __VA_ARGS__);                                   \
        MU_CALL(FN, BM_GEN, false, r, vec, __VA_ARGS__);                                                  \
    } while (0)

static int mu_args_ok(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                      const float* V, int64_t ldv, int r, double beta, const float* out) {
    if (!ctx || !X || !Ut || !V || !out || m < 1 || n < 1 || r < 1 || ldx < n || ldu < m || ldv < n) return NNF_ERR_ARG;
    if (!(beta >= 0.0)) return NNF_ERR_ARG;
    if (r > NNF_MAX_RANK) return NNF_ERR_UNSUPPORTED;
    return NNF_OK;
}

#if MU_PART == 1
extern "C" int nnf_mu_left_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                               int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* Ut_out, int64_t lduo,
                               void* stream) {
    int rc = mu_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, Ut_out);
    if (rc != NNF_OK) return rc;
    if (lduo < m) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    if (beta == 2.0) {  // U * (X V^T) / (U (V V^T))   (mu.py:89-91 reordered through the r x r Gram)
        float* G = (float*)cur.take((size_t)r * r * 4);
        float* num = (float*)cur.take((size_t)r * nnf_rup(m, 4) * 4);
        if (!G || !num) return NNF_ERR_WORKSPACE;
        if ((rc = nnf_gram_impl(ctx, cur, V, r, n, ldv, G, r, st)) != NNF_OK) return rc;
        if ((rc = nnf_xht_impl(ctx, cur, X, m, n, ldx, V, r, ldv, num, nnf_rup(m, 4), st)) != NNF_OK) return rc;
        int64_t fg = nnf_cdiv(m, 256);
        if (fg > 2048) fg = 2048;
        hipLaunchKernelGGL(nnf_mu2_finish_kernel, dim3((int)fg), dim3(256), 0, st, Ut, ldu, r, m, G, num, nnf_rup(m, 4),
                           Ut_out, lduo);
        NNF_CHECK_LAUNCH();
        return NNF_OK;
    }
    MU_DISPATCH(launch_mu_left, ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, Ut_out, lduo, st);
}

#endif   // MU_PART == 1
#if MU_PART == 0
extern "C" int nnf_mu_right_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                                int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* V_out, int64_t ldvo,
                                void* stream) {
    int rc = mu_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, V_out);
    if (rc != NNF_OK) return rc;
    if (ldvo < n) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    if (beta == 2.0) {  // V * (U^T X) / ((U^T U) V)
        float* G = (float*)cur.take((size_t)r * r * 4);
        float* num = (float*)cur.take((size_t)r * nnf_rup(n, 4) * 4);
        if (!G || !num) return NNF_ERR_WORKSPACE;
        if ((rc = nnf_gram_impl(ctx, cur, Ut, r, m, ldu, G, r, st)) != NNF_OK) return rc;
        if ((rc = nnf_xty_impl(ctx, cur, X, m, n, ldx, Ut, r, ldu, num, nnf_rup(n, 4), st)) != NNF_OK) return rc;
        int64_t fg = nnf_cdiv(n, 256);
        if (fg > 2048) fg = 2048;
        hipLaunchKernelGGL(nnf_mu2_finish_kernel, dim3((int)fg), dim3(256), 0, st, V, ldv, r, n, G, num, nnf_rup(n, 4),
                           V_out, ldvo);
        NNF_CHECK_LAUNCH();
        return NNF_OK;
    }
    MU_DISPATCH(launch_mu_right, ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, V_out, ldvo, st);
}


// ---- two-phase right update for row-sharded runs: every rank accumulates the numerator / denominator
// of its row block, the host all-reduces them, nnf_mu_apply_f32 finishes.  beta = 2 goes through the Gram form
// (num = Ut X, den = (Ut U) V, both linear in the row blocks); beta = 1 has den[k] = colsum(U)[k] (r doubles).
__global__ __launch_bounds__(256) void nnf_small_gemm_kernel(const float* __restrict__ G, int r, const float* __restrict__ V,
                                                             int64_t ldv, int64_t n, float* __restrict__ out, int64_t ldo) {
    const int64_t total = (int64_t)r * n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / n, j = e - k * n;
        float s = 0.f;
        for (int l = 0; l < r; l += 8) {   // same order as nnf_mu2_finish_kernel; eight entries in flight per trip
            float vv[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lc = l + u < r ? l + u : r - 1;
                vv[u] = V[(int64_t)lc * ldv + j];
                gv[u] = G[k * r + lc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (l + u < r) s = fmaf(gv[u], vv[u], s);
        }
        out[k * ldo + j] = s;
    }
}

extern "C" int nnf_mu_right_accum_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                                      int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* num, int64_t ldnum,
                                      float* den, int64_t ldden, double* den_vec_f64, void* stream) {
    int rc = mu_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, num);
    if (rc != NNF_OK) return rc;
    if (ldnum < n) return NNF_ERR_ARG;
    if (beta == 1.0 ? (den_vec_f64 == nullptr) : (den == nullptr || ldden < n)) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    if (beta == 2.0) {
        float* G = (float*)cur.take((size_t)r * r * 4);
        if (!G) return NNF_ERR_WORKSPACE;
        if ((rc = nnf_gram_impl(ctx, cur, Ut, r, m, ldu, G, r, st)) != NNF_OK) return rc;
        if ((rc = nnf_xty_impl(ctx, cur, X, m, n, ldx, Ut, r, ldu, num, ldnum, st)) != NNF_OK) return rc;
        int64_t fg = nnf_cdiv((int64_t)r * n, 256);
        if (fg > 2048) fg = 2048;
        hipLaunchKernelGGL(nnf_small_gemm_kernel, dim3((int)fg), dim3(256), 0, st, G, r, V, ldv, n, den, ldden);
        NNF_CHECK_LAUNCH();
        return NNF_OK;
    }
    MU_DISPATCH(launch_mu_right, ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, nullptr, 0, st, num, ldnum, den, ldden,
                den_vec_f64);
}

extern "C" int nnf_mu_apply_f32(nnf_ctx* ctx, const float* F, int64_t ldf, int r, int64_t cols, const float* num,
                                int64_t ldnum, const float* den, int64_t ldden, const double* den_vec_f64, double beta,
                                float* out, int64_t ldo, void* stream) {
    if (!ctx || !F || !num || !out || r < 1 || cols < 1 || ldf < cols || ldnum < cols || ldo < cols || !(beta >= 0.0))
        return NNF_ERR_ARG;
    if (den_vec_f64 == nullptr && (den == nullptr || ldden != ldnum)) return NNF_ERR_ARG;
    int64_t fg = nnf_cdiv((int64_t)r * cols, 256);
    if (fg > 2048) fg = 2048;
    hipLaunchKernelGGL(nnf_mu_finish_kernel, dim3((int)fg), dim3(256), 0, (hipStream_t)stream, F, ldf, r, cols, num, den, 1,
                       (int64_t)0, ldnum, den_vec_f64, gamma_of(beta), out, ldo);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

#endif   // MU_PART == 0
#if MU_PART == 1
// Raw KL numerator of the left update, num[k,i] = sum_j (X[i,j] / (UV)[i,j]) V[k,j]  (mu.py:85, before the division by the
// row sums of V): the `b` term of deep_KL_mu (deep_mu.py:10) is U .* num.  Same fused kernel as nnf_mu_left_f32.
extern "C" int nnf_mu_left_num_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                                   int64_t ldu, const float* V, int64_t ldv, int r, float* num, int64_t ldnum, void* stream) {
    const double beta = 1.0;
    int rc = mu_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, num);
    if (rc != NNF_OK) return rc;
    if (ldnum < m) return NNF_ERR_ARG;
    if (r > 64) return NNF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    MU_DISPATCH(launch_mu_left, ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, num, ldnum, st, 1);
}

#endif   // MU_PART == 1
#if MU_PART == 0
// out[p x cols] = A[p x q] * B[q x cols]: a rank-sized left operand against a wide matrix (deep NMF: (W_{l+1} H_{l+1})^T =
// H_{l+1}^T W_{l+1}^T, deep_nmf.py:93; the rank-sized links of the NTD chains).  One thread per output, k in order.
__global__ __launch_bounds__(256) void nnf_small_gemm_rect_kernel(const float* __restrict__ A, int64_t lda, int p, int q,
                                                                  const float* __restrict__ B, int64_t ldb, int64_t cols,
                                                                  float* __restrict__ out, int64_t ldo, int64_t bstride,
                                                                  int64_t ostride) {
    // grid = (column blocks of 256, chunks of 8 output rows, batch); thread = one column, 8 outputs, k in order
    extern __shared__ float sA[];
    const int k0 = blockIdx.y * 8;
    const int nr = (p - k0 < 8) ? (p - k0) : 8;
    for (int e = threadIdx.x; e < nr * q; e += 256) sA[e] = A[(int64_t)(k0 + e / q) * lda + (e % q)];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    const float* Bb = B + (int64_t)blockIdx.z * bstride;
    float* ob = out + (int64_t)blockIdx.z * ostride;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l0 = 0; l0 < q; l0 += 8) {      // eight loads of B in flight (one at a time the loop is a chain of memory latencies)
        float b[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = (l0 + t < q) ? Bb[(int64_t)(l0 + t) * ldb + j] : 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (l0 + t < q) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < nr) acc[u] = fmaf(sA[u * q + l0 + t], b[t], acc[u]);
            }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nr) ob[(int64_t)(k0 + u) * ldo + j] = acc[u];
}
// batched form: out[z] = A B[z] for z < batch (B[z] = B + z * bstride, out[z] = out + z * ostride); batch = 1: plain
int nnf_small_gemm_launch(const float* A, int64_t lda, int p, int q, const float* B, int64_t ldb, int64_t cols, float* out,
                          int64_t ldo, int64_t batch, int64_t bstride, int64_t ostride, hipStream_t st) {
    if ((int64_t)8 * q * 4 > 64 * 1024 || batch > 65535 || nnf_cdiv(p, 8) > 65535) return NNF_ERR_UNSUPPORTED;
    const size_t shm = (size_t)8 * q * 4;
    if (shm > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_small_gemm_rect_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(nnf_small_gemm_rect_kernel, dim3((unsigned)nnf_cdiv(cols, 256), (unsigned)nnf_cdiv(p, 8), (unsigned)batch),
                       dim3(256), shm, st, A, lda, p, q, B, ldb, cols, out, ldo, bstride, ostride);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
extern "C" int nnf_small_gemm_f32(nnf_ctx* ctx, const float* A, int64_t lda, int p, int q, const float* B, int64_t ldb,
                                  int64_t cols, float* out, int64_t ldo, void* stream) {
    if (!ctx || !A || !B || !out || p < 1 || q < 1 || cols < 1 || lda < q || ldb < cols || ldo < cols) return NNF_ERR_ARG;
    return nnf_small_gemm_launch(A, lda, p, q, B, ldb, cols, out, ldo, 1, 0, 0, (hipStream_t)stream);   // (q <= 2048: 8 rows of A in LDS)
}

// deep_KL_mu (deep_mu.py:8-14), element-wise tail:  a = hsum[k] - lambda*log(WHnext[k,i]),  b = F[k,i]*num[k,i],
//   out = max(1e-12, (b/lambda) / (W0(b*exp(a/lambda)/lambda) + 1e-12))      with W0 the principal Lambert W branch.
// exp(a/lambda) overflows long before its product with b does, so the argument is carried as its logarithm
// L = log b + a/lambda - log lambda and w + log w = L is solved by Newton steps in fp64 (w > 0; quadratic from the
// asymptotic start L - log L for L > 1, from z/(1+z) below); for L < -36 W0(z) = z to double precision.
__device__ __forceinline__ double nnf_lambertw_logarg(double L) {
    if (L < -36.0) return exp(L);
    double w;
    if (L > 1.0) {
        w = L - log(L);
    } else {
        const double z = exp(L);
        w = z / (1.0 + z);
        if (w < 1e-300) return z;
    }
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const double f = w + log(w) - L;
        const double wn = w - f * w / (1.0 + w);
        const double d = fabs(wn - w);
        w = wn > 0.0 ? wn : 0.5 * w;
        if (d <= 1e-15 * fabs(w)) break;
    }
    return w;
}
__global__ __launch_bounds__(256) void nnf_deep_kl_apply_kernel(const float* __restrict__ F, int64_t ldf, int r, int64_t cols,
                                                                const float* __restrict__ num, int64_t ldn,
                                                                const double* __restrict__ hsum,
                                                                const float* __restrict__ WHn, int64_t ldw, double lambda,
                                                                float* __restrict__ out, int64_t ldo) {
    const int64_t total = (int64_t)r * cols;
    const double loglam = log(lambda);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / cols, i = e - k * cols;
        const double b = (double)F[k * ldf + i] * (double)num[k * ldn + i];
        const double a = hsum[k] - lambda * log((double)WHn[k * ldw + i]);
        double res;
        const double q = a / lambda;
        const double Lz = log(b) + q - loglam;
        if (!(b > 0.0)) {
            res = 0.0;   // b = 0: numerator 0 (the reference gives 0 / (0 + eps) = 0, then the 1e-12 floor)
        } else if (q > 709.782712893384 || Lz > 709.782712893384) {
            // the reference forms exp(a/lambda) and b*exp(.)/lambda in float64 (deep_mu.py:11): beyond log(DBL_MAX) that is
            // +inf, lambertw(inf) = inf and the quotient is 0 -> the floor.  Kept: results identical to the reference's.
            res = 0.0;
        } else {
            const double w = nnf_lambertw_logarg(Lz);
            res = (b / lambda) / (w + 1e-12);
        }
        out[k * ldo + i] = (float)fmax(1e-12, res);
    }
}
extern "C" int nnf_deep_kl_apply_f32(nnf_ctx* ctx, const float* F, int64_t ldf, int r, int64_t cols, const float* num,
                                     int64_t ldnum, const double* hsum_f64, const float* WHnext, int64_t ldw, double lambda,
                                     float* out, int64_t ldo, void* stream) {
    if (!ctx || !F || !num || !hsum_f64 || !WHnext || !out || r < 1 || cols < 1 || ldf < cols || ldnum < cols ||
        ldw < cols || ldo < cols || !(lambda > 0.0))
        return NNF_ERR_ARG;
    int64_t fg = nnf_cdiv((int64_t)r * cols, 256);
    if (fg > 4096) fg = 4096;
    hipLaunchKernelGGL(nnf_deep_kl_apply_kernel, dim3((int)fg), dim3(256), 0, (hipStream_t)stream, F, ldf, r, cols, num, ldnum,
                       hsum_f64, WHnext, ldw, lambda, out, ldo);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

#endif   // MU_PART == 0
#if MU_PART == 2
// One pass over a dense 3-way tensor T (I x J x K) for BOTH the squared residual of the current CP model and the partial
// product the next iteration's mode-0 / mode-1 right-hand sides are contracted from (nnf_mttkrp3_from_partial_f32):
//   *cost_f64 = sum_ijk (T[i,j,k] - sum_r F0[i,r] F1[j,r] F2[k,r])^2        (ntf.py:470, evaluated directly)
//   Y[r][i][j] = sum_k T[i,j,k] F2[k,r]                                     (tl.tenalg.mode_dot(T, F2^T, 2), new axis first)
// Both need the final factors of an iteration and the whole tensor: fused, an iteration reads T twice (this pass + the
// mode-2 MTTKRP) instead of four times.  T is seen as an (I*J) x K matrix; the left factor rows F0[i,:].*F1[j,:] are
// generated once per wave; the kernel is the left MU kernel with R = X (MFMA #1: model tile, MFMA #2: Y += F2-fragment . X).
extern "C" int nnf_cp3_partial_cost_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft0,
                                        int64_t ld0, const float* Ft1, int64_t ld1, const float* Ft2, int64_t ld2, int R,
                                        float* Y, double* cost_f64, void* stream) {
    if (!ctx || !T || !Ft0 || !Ft1 || !Ft2 || !Y || !cost_f64 || I < 1 || J < 1 || K < 1 || R < 1 || ld0 < I || ld1 < J ||
        ld2 < K)
        return NNF_ERR_ARG;
    if (R > 64) return NNF_ERR_UNSUPPORTED;   // (the fused two-MFMA kernels are built for rank <= 64; callers fall back)
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    const int64_t m = I * J;
    const mu_left_extra ex{Ft1, ld1, J, nullptr};
    const bool vec = x_vec_ok(T, K);
    MU_CALL(launch_mu_left, BM_FROB, true, R, vec, ctx, cur, T, m, K, K, Ft0, ld0, Ft2, ld2, R, 2.0, Y, m, st, 1, ex, cost_f64);
}

// KL multiplicative update of the left factor (nnf_mu_left_f32 with beta = 1) that ALSO returns beta_divergence(X, U V, 1)
// of the factors it starts from (mu.py:84-88 + nmf.py:455).  The update kernel forms every entry of P = U V anyway and holds
// the matching X value: the divergence term rides along (VALU work next to an MFMA-bound kernel), so the cost of outer
// iteration i is a by-product of the left update of iteration i+1 and the separate pass over X (nnf_betadiv_f32: a quarter of
// a KL iteration at 100000 x 2000, rank 50) is only needed after the last iteration.  Same update as nnf_mu_left_f32, bit
// for bit.  r <= 64.
extern "C" int nnf_mu_left_kl_cost_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                                       int64_t ldu, const float* V, int64_t ldv, int r, float* Ut_out, int64_t lduo,
                                       double* cost_f64, void* stream) {
    int rc = mu_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, 1.0, Ut_out);
    if (rc != NNF_OK) return rc;
    if (lduo < m || !cost_f64) return NNF_ERR_ARG;
    if (r > 64) return NNF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    const mu_left_extra ex{nullptr, 0, 1, nullptr};
    const bool vec = x_vec_ok(X, ldx);
    MU_CALL(launch_mu_left, BM_KLC, true, r, vec, ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, 1.0, Ut_out, lduo, st, 0, ex, cost_f64);
}
#endif   // MU_PART == 2
