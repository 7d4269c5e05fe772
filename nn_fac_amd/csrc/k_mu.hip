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
#include "k_stream_common.h"
#include <math.h>

enum { BM_KL = 1, BM_FROB = 2, BM_KLC = 3, BM_GEN = 9 };   // BM_KLC: the KL update + the KL divergence of its INPUT factors   // BM_FROB: R = X (plain X V^T) + the squared residual, see nnf_cp3_partial_cost_f32

// extra operands of the left kernel's BM_FROB form: Khatri-Rao left factor generated from two short factors, cost partials
struct mu_left_extra {
    const float* Fb;      // != nullptr: U[k][i] = Ut[k][i / nb] * Fb[k][i % nb]  (row (a, b) of a 3-way tensor seen as (A*B) x K)
    int64_t ldb, nb;
    double* partial;      // BM_FROB: one fp64 partial of sum (X - UV)^2 per workgroup
};

// F_K image of a 64-wide chunk of a row-major r x K matrix A (the rank index is the MFMA k index):
//   img[(t*MT + s4)*64 + lane].c = A[16*s4 + 4*c + (lane>>4)][k0 + 16*t + (lane&15)]       (zero outside r x K)
template <int MT>
__device__ __forceinline__ void stageK_load(const float* __restrict__ A, int64_t lda, int r, int64_t K, int64_t k0,
                                            f32x4 (&regs)[MT]) {
    const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
    const int64_t col = k0 + 16 * t + (L & 15);
#pragma unroll
    for (int s4 = 0; s4 < MT; ++s4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col < K) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = 16 * s4 + 4 * c + (L >> 4);
                if (row < r) v[c] = A[(int64_t)row * lda + col];
            }
        }
        regs[s4] = v;
    }
}
template <int MT>
__device__ __forceinline__ void stageK_store(f32x4* __restrict__ img, const f32x4 (&regs)[MT]) {
    const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
#pragma unroll
    for (int s4 = 0; s4 < MT; ++s4) img[(t * MT + s4) * 64 + L] = regs[s4];
}
template <int MT>
__device__ __forceinline__ void stageK(const float* __restrict__ A, int64_t lda, int r, int64_t K, int64_t k0,
                                       f32x4* __restrict__ img) {
    f32x4 regs[MT];
    stageK_load<MT>(A, lda, r, K, k0, regs);
    stageK_store<MT>(img, regs);
}

template <int MT>
__device__ __forceinline__ void stageA_direct(const float* __restrict__ A, int64_t lda, int r, int64_t K, int64_t k0,
                                              bool vec_ok, f32x4* __restrict__ img) {
    f32x4 regs[MT];
    stageA_load<MT>(A, lda, r, K, k0, vec_ok, regs);
    stageA_store<MT>(img, regs);
}

template <int BM>
__device__ __forceinline__ void mu_elem(float x, float p, float beta, float& r1, float& r2) {
    if constexpr (BM == BM_KL || BM == BM_KLC) {
        r1 = x * __builtin_amdgcn_rcpf(p);
        r2 = 0.f;
    } else if constexpr (BM == BM_FROB) {
        r1 = x;
        r2 = 0.f;
    } else {
        // r2 = p^(beta-1), r1 = p^(beta-2) x
        const float lp = __builtin_amdgcn_logf(p);               // log2
        r2 = __builtin_amdgcn_exp2f((beta - 1.f) * lp);
        r1 = r2 * __builtin_amdgcn_rcpf(p) * x;
    }
}

// =========================================================================================================
// right update: slabs of num (and den) [ks][r][ldp], split over the rows of X like xty.
// =========================================================================================================
// The V fragments of a wave's 64 columns are loop-invariant and live in registers (4*MT float4, straight from global),
// which leaves 64 KB of LDS (the two double-buffered images of the Ut chunk) and lets two workgroups share a CU.
// (KL only: the general-beta form carries a second accumulator set and keeps the fragments in LDS, one workgroup per CU.)
template <int MT, int BM, bool VEC>
__global__ __launch_bounds__(256, (BM == BM_KL ? 2 : 1)) void nnf_mu_right_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                              const float* __restrict__ Ut, int64_t ldu,
                                                              const float* __restrict__ V, int64_t ldv, int r, float beta,
                                                              float* __restrict__ snum, float* __restrict__ sden,
                                                              int64_t ldp, int ncb, int nsplit, int64_t rows_per_split,
                                                              int a_vec_ok) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = (r + 3) >> 2;
    constexpr bool REGF = (BM == BM_KL);                             // resident fragments in registers / in LDS
    f32x4* ldsVf = reinterpret_cast<f32x4*>(smem);                 // !REGF: [4][KS][64]: V[4s+g][jw+4jj..+3]
    f32x4* ldsA = ldsVf + (REGF ? 0 : (size_t)4 * KS * 64);          // [2][MT*256]  F_A image of the Ut chunk
    f32x4* ldsK = ldsA + (size_t)2 * MT * 256;                       // [2][MT*256]  F_K image of the Ut chunk
    int ks, cb;
    nnf_xcd_map(blockIdx.x, ncb, ks, cb);
    if (ks >= nsplit) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, g = lane >> 4;
    const int64_t i_begin = (int64_t)ks * rows_per_split;
    const int64_t i_end = (i_begin + rows_per_split < m) ? (i_begin + rows_per_split) : m;
    const int nchunk = (int)((i_end - i_begin + 63) >> 6);
    const int64_t jw = (int64_t)cb * 256 + w * 64, jl = jw + 4 * jj;
    const rsrc_t rs = nnf_make_rsrc(X + i_begin * ldx, (uint32_t)(((i_end - i_begin - 1) * ldx + n) * 4));
    const int voff = (jl < n) ? (int)(((int64_t)4 * g * ldx + jl) * 4) : (int)0x7ffffff0;
    const int ldx4 = (int)(ldx * 4);

    // resident V fragments of this wave's 64 columns: vfr[s] = V[4s+g][jl .. jl+3], s < KS (zero beyond r x n)
    f32x4 vfr[REGF ? 4 * MT : 1];
    if constexpr (REGF) {
#pragma unroll
        for (int s_ = 0; s_ < 4 * MT; ++s_) {
            const int k = 4 * s_ + g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < r && jl < n) {
                const float* p = V + (int64_t)k * ldv + jl;
                v[0] = p[0];
                if (jl + 1 < n) v[1] = p[1];
                if (jl + 2 < n) v[2] = p[2];
                if (jl + 3 < n) v[3] = p[3];
            }
            vfr[s_] = v;
        }
    } else {
        for (int e = threadIdx.x; e < 4 * KS * 64; e += 256) {
            const int ww = e / (KS * 64), rem = e - ww * KS * 64, s_ = rem >> 6, L = rem & 63;
            const int k = 4 * s_ + (L >> 4);
            const int64_t j = (int64_t)cb * 256 + ww * 64 + 4 * (L & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < r && j < n) {
                const float* p = V + (int64_t)k * ldv + j;
                v[0] = p[0];
                if (j + 1 < n) v[1] = p[1];
                if (j + 2 < n) v[2] = p[2];
                if (j + 3 < n) v[3] = p[3];
            }
            ldsVf[e] = v;
        }
    }
    f32x4 num[MT][4], den[BM == BM_GEN ? MT : 1][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            num[mt][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BM == BM_GEN) den[mt][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 xb[2][4];   // ring of two 16-row groups: group gi lives in xb[gi & 1] and is refilled with group gi + 2
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[t][c] = nnf_bload4<VEC>(rs, voff, (16 * t + c) * ldx4);
    stageA_direct<MT>(Ut, ldu, r, i_end, i_begin, a_vec_ok, ldsA);
    stageK<MT>(Ut, ldu, r, i_end, i_begin, ldsK);
    __syncthreads();

    for (int q = 0; q < nchunk; ++q) {
        const f32x4* imgA = ldsA + (size_t)(q & 1) * MT * 256;
        const f32x4* imgK = ldsK + (size_t)(q & 1) * MT * 256;
        // next chunk's operand images: global loads now, LDS writes after this chunk's MFMAs (past the end: zeros)
        // the two images are staged through registers one after the other (A during groups 0-1, K during groups 2-3):
        // half the staging registers of loading both up front
        f32x4 sa[MT], sk[MT];
        stageA_load<MT>(Ut, ldu, r, i_end, i_begin + 64 * (int64_t)(q + 1), a_vec_ok, sa);
        const int soff_q = q * 64 * ldx4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // MFMA #1: P[i0+16t+4g+reg][jw+4jj+cc]
            f32x4 accP[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) accP[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < MT; ++s4) {
                const f32x4 ak = imgK[(t * MT + s4) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (4 * s4 + c < KS) {
                        f32x4 bv;
                        if constexpr (REGF) bv = vfr[4 * s4 + c];
                        else bv = ldsVf[(size_t)w * KS * 64 + (4 * s4 + c) * 64 + lane];
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) accP[cc] = MFMA16(ak[c], bv[cc], accP[cc]);
                    }
                }
            }
            // element-wise, masked past the split's last row (0/0 otherwise)
            const int64_t rowrem = (i_end - i_begin) - (64 * (int64_t)q + 16 * t + 4 * g);
            f32x4 R1[4], R2[BM == BM_GEN ? 4 : 1];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float r1, r2;
                    mu_elem<BM>(xb[t & 1][reg][cc], accP[cc][reg], beta, r1, r2);
                    const bool ok = reg < rowrem;
                    R1[cc][reg] = ok ? r1 : 0.f;
                    if constexpr (BM == BM_GEN) R2[cc][reg] = ok ? r2 : 0.f;
                }
            // MFMA #2: num[rk][j] += Ut[rk][i] * R[i][j], k = the block's 16 rows
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = imgA[(mt * 4 + t) * 64 + lane];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        num[mt][cc] = MFMA16(af[mt][reg], R1[cc][reg], num[mt][cc]);
                        if constexpr (BM == BM_GEN) den[mt][cc] = MFMA16(af[mt][reg], R2[cc][reg], den[mt][cc]);
                    }
#pragma unroll
            for (int c = 0; c < 4; ++c) xb[t & 1][c] = nnf_bload4<VEC>(rs, voff, soff_q + (16 * (t + 2) + c) * ldx4);
            if (t == 1) {
                stageA_store<MT>(ldsA + (size_t)((q + 1) & 1) * MT * 256, sa);
                stageK_load<MT>(Ut, ldu, r, i_end, i_begin + 64 * (int64_t)(q + 1), sk);
            }
        }
        stageK_store<MT>(ldsK + (size_t)((q + 1) & 1) * MT * 256, sk);
        __syncthreads();
    }
    if (jl < ldp) {
        float* sn = snum + (int64_t)ks * r * ldp;
        float* sd = (BM == BM_GEN) ? sden + (int64_t)ks * r * ldp : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rk = 16 * mt + 4 * g + reg;
                if (rk < r) {
                    *reinterpret_cast<f32x4*>(sn + (int64_t)rk * ldp + jl) =
                        f32x4{num[mt][0][reg], num[mt][1][reg], num[mt][2][reg], num[mt][3][reg]};
                    if constexpr (BM == BM_GEN)
                        *reinterpret_cast<f32x4*>(sd + (int64_t)rk * ldp + jl) =
                            f32x4{den[mt][0][reg], den[mt][1][reg], den[mt][2][reg], den[mt][3][reg]};
                }
            }
    }
}

// F_new = max(F * (num/den)^gamma, 1e-12); num/den summed over slabs in fp64 (fixed order); den_vec: per-row denominator (KL)
__global__ __launch_bounds__(256) void nnf_mu_finish_kernel(const float* __restrict__ F, int64_t ldf, int r, int64_t cols,
                                                            const float* __restrict__ snum, const float* __restrict__ sden,
                                                            int nslab, int64_t slab_stride, int64_t lds,
                                                            const double* __restrict__ den_vec, float gamma,
                                                            float* __restrict__ out, int64_t ldo) {
    const int64_t total = (int64_t)r * cols;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / cols, j = e - k * cols;
        double nu = 0.0, de = 0.0;
        for (int s = 0; s < nslab; ++s) nu += (double)snum[(int64_t)s * slab_stride + k * lds + j];
        if (den_vec) de = den_vec[k];
        else
            for (int s = 0; s < nslab; ++s) de += (double)sden[(int64_t)s * slab_stride + k * lds + j];
        float ratio = (float)(nu / de);
        if (gamma != 1.f) ratio = powf(ratio, gamma);
        out[k * ldo + j] = fmaxf(F[k * ldf + j] * ratio, 1e-12f);
    }
}

// out[k] = sum_j A[k][j]  (fp64).  Long rows (the r x m factor: one workgroup per row took 112 us at m = 100000) are cut
// into gridDim.y pieces whose partial sums (part[k][piece]) are added in piece order by a second, tiny launch.
__global__ __launch_bounds__(256) void nnf_rowsum_kernel(const float* __restrict__ A, int64_t lda, int64_t K,
                                                         double* __restrict__ out, int64_t per) {
    __shared__ double red[4];
    const float* p = A + (int64_t)blockIdx.x * lda;
    const int64_t j0 = (int64_t)blockIdx.y * per, j1 = (j0 + per < K) ? (j0 + per) : K;
    double s = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) s += (double)p[j];
    const double t = nnf_block_sum_f64(s, red);
    if (threadIdx.x == 0) out[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = t;
}
__global__ __launch_bounds__(64) void nnf_rowsum_fin_kernel(const double* __restrict__ part, int np, int r,
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

// =========================================================================================================
// left update: workgroup = 64*NT rows of X (wave: NT 16-row N tiles), sweeping all columns; no split.
// =========================================================================================================
// NT = 16-row tiles per wave: a workgroup covers 64*NT rows starting at row0 (see nnf_xht_kernel for why the host mixes
// workgroups of NTH and NTH-1 tiles: one balanced round of resident workgroups instead of 391 on 512 slots).
template <int MT, int BM, bool VEC, int NT>
__device__ __forceinline__ void nnf_mu_left_body(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                 const float* __restrict__ Ut, int64_t ldu,
                                                 const float* __restrict__ V, int64_t ldv, int r, float beta,
                                                 const double* __restrict__ den_vec, float gamma,
                                                 float* __restrict__ Ut_out, int64_t lduo, int a_vec_ok, int64_t row0,
                                                 char* smem, const mu_left_extra& ex) {
    const int KS = (r + 3) >> 2;
    constexpr bool REGF = (BM != BM_GEN);                            // resident fragments in registers / in LDS
    float csum = 0.f;                                                // BM_FROB: this lane's share of sum (X - UV)^2
    f32x4* ldsUf = reinterpret_cast<f32x4*>(smem);                 // !REGF: [4][KS][64]: comps nt: Ut[4s+g][i0w+16nt+ii]
    f32x4* ldsA = ldsUf + (REGF ? 0 : (size_t)4 * KS * 64);          // [2][MT*256]  F_A image of the V chunk
    f32x4* ldsK = ldsA + (size_t)2 * MT * 256;                       // [2][MT*256]  F_K image of the V chunk
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ii = lane & 15, g = lane >> 4;
    const int64_t i0w = row0 + 16 * NT * w;
    int64_t rows = m - i0w;
    if (rows > 16 * NT) rows = 16 * NT;
    const uint32_t bytes = rows > 0 ? (uint32_t)(((rows - 1) * ldx + n) * 4) : 0u;
    const rsrc_t rs = nnf_make_rsrc(X + (rows > 0 ? i0w : 0) * ldx, bytes);
    const int voff = (int)(((int64_t)ii * ldx + 4 * g) * 4);
    const int ldx4 = (int)(ldx * 4);
    const int nchunk = (int)((n + 63) >> 6);

    // resident U fragments of this wave's 64 rows, in registers (like the V fragments of the right kernel):
    // ufr[s][nt] = Ut[4s+g][i0w + 16nt + ii], s < KS (zero beyond r x m)
    f32x4 ufr[REGF ? 4 * MT : 1];
    if constexpr (REGF) {
        int64_t kra[NT], krb[NT];      // Khatri-Rao left factor: row i of the (A*B) x K view = (i / nb, i % nb), once per tile
        if (ex.Fb != nullptr) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t i = i0w + 16 * nt + ii;
                kra[nt] = i / ex.nb;
                krb[nt] = i - kra[nt] * ex.nb;
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < 4 * MT; ++s_) {
            const int k = 4 * s_ + g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < r) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t i = i0w + 16 * nt + ii;
                    if (i < m) {
                        if (ex.Fb != nullptr) {   // Khatri-Rao row generated on the fly (loop-invariant: once per wave)
                            v[nt] = Ut[(int64_t)k * ldu + kra[nt]] * ex.Fb[(int64_t)k * ex.ldb + krb[nt]];
                        } else {
                            v[nt] = Ut[(int64_t)k * ldu + i];
                        }
                    }
                }
            }
            ufr[s_] = v;
        }
    } else {
        for (int e = threadIdx.x; e < 4 * KS * 64; e += 256) {
            const int ww = e / (KS * 64), rem = e - ww * KS * 64, s_ = rem >> 6, L = rem & 63;
            const int k = 4 * s_ + (L >> 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < r) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t i = row0 + 16 * NT * ww + 16 * nt + (L & 15);
                    if (i < m) v[nt] = Ut[(int64_t)k * ldu + i];
                }
            }
            ldsUf[e] = v;
        }
    }
    f32x4 num[MT][4], den[BM == BM_GEN ? MT : 1][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            num[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (BM == BM_GEN) den[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 xb[2][4];  // ring of two 16-column groups [group parity][nt]: X[i0w+16nt+ii][16*gi+4g .. +3]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[t][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldx4 + 64 * t);
    stageA_direct<MT>(V, ldv, r, n, 0, a_vec_ok, ldsA);
    stageK<MT>(V, ldv, r, n, 0, ldsK);
    __syncthreads();

    for (int q = 0; q < nchunk; ++q) {
        const f32x4* imgA = ldsA + (size_t)(q & 1) * MT * 256;
        const f32x4* imgK = ldsK + (size_t)(q & 1) * MT * 256;
        f32x4 sa[MT], sk[MT];   // staged one after the other (see the right kernel)
        stageA_load<MT>(V, ldv, r, n, 64 * (int64_t)(q + 1), a_vec_ok, sa);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // MFMA #1 (transposed product): accP[nt][reg] = P[i0w+16nt+ii][64q+16t+4g+reg]
            f32x4 accP[4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) accP[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < MT; ++s4) {
                const f32x4 ak = imgK[(t * MT + s4) * 64 + lane];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (4 * s4 + c < KS) {
                        f32x4 bu;
                        if constexpr (REGF) bu = ufr[4 * s4 + c];
                        else bu = ldsUf[(size_t)w * KS * 64 + (4 * s4 + c) * 64 + lane];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) accP[nt] = MFMA16(ak[c], bu[nt], accP[nt]);
                    }
                }
            }
            const int64_t colrem = n - (64 * (int64_t)q + 16 * t + 4 * g);
            f32x4 R1[4], R2[BM == BM_GEN ? 4 : 1];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool rowok = (16 * nt + ii) < rows;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float r1, r2;
                    mu_elem<BM>(xb[t & 1][nt][reg], accP[nt][reg], beta, r1, r2);
                    const bool ok = rowok && (reg < colrem);
                    if constexpr (BM == BM_FROB) {
                        const float dd = ok ? (xb[t & 1][nt][reg] - accP[nt][reg]) : 0.f;
                        csum = fmaf(dd, dd, csum);
                    }
                    if constexpr (BM == BM_KLC) {   // beta_divergence(X, UV, 1) of the factors this update starts from
                        const float term = nnf_cost_term<NNF_COST_KL>(xb[t & 1][nt][reg], accP[nt][reg], 1.f);
                        csum += ok ? term : 0.f;
                    }
                    R1[nt][reg] = ok ? r1 : 0.f;
                    if constexpr (BM == BM_GEN) R2[nt][reg] = ok ? r2 : 0.f;
                }
            }
            // finish this group's residual sum HERE: left alone, LLVM sinks the whole dependent chain of a chunk (and the 48
            // differences it consumes) to the chunk's last block -- 256 VGPRs + spills instead of ~180
            if constexpr (BM == BM_FROB || BM == BM_KLC) asm volatile("" : "+v"(csum));
            // MFMA #2: num[rk][i] += V[rk][j] * R[j][i], k = the block's 16 columns
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = imgA[(mt * 4 + t) * 64 + lane];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        num[mt][nt] = MFMA16(af[mt][reg], R1[nt][reg], num[mt][nt]);
                        if constexpr (BM == BM_GEN) den[mt][nt] = MFMA16(af[mt][reg], R2[nt][reg], den[mt][nt]);
                    }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                xb[t & 1][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldx4 + 256 * q + 64 * (t + 2));
            if (t == 1) {
                stageA_store<MT>(ldsA + (size_t)((q + 1) & 1) * MT * 256, sa);
                stageK_load<MT>(V, ldv, r, n, 64 * (int64_t)(q + 1), sk);
            }
        }
        stageK_store<MT>(ldsK + (size_t)((q + 1) & 1) * MT * 256, sk);
        __syncthreads();
    }
    // epilogue: tile (mt, nt): rk = 16mt+4g+reg, i = i0w+16nt+ii
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t i = i0w + 16 * nt + ii;
        if (i < m) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int rk = 16 * mt + 4 * g + reg;
                    if (rk < r) {
                        float d;
                        if constexpr (BM == BM_GEN) d = den[mt][nt][reg]; else d = (float)den_vec[rk];
                        if (BM == BM_FROB || gamma < 0.f) {   // raw numerator (nnf_mu_left_num_f32; wave-uniform flag)
                            Ut_out[(int64_t)rk * lduo + i] = num[mt][nt][reg];
                            continue;
                        }
                        float ratio = num[mt][nt][reg] / d;
                        if (gamma != 1.f) ratio = powf(ratio, gamma);
                        Ut_out[(int64_t)rk * lduo + i] = fmaxf(Ut[(int64_t)rk * ldu + i] * ratio, 1e-12f);
                    }
                }
        }
    }
    if constexpr (BM == BM_FROB || BM == BM_KLC) {   // fp32 per lane (a few hundred terms), fp64 from the wave level up, fixed order
        double* red = reinterpret_cast<double*>(smem);    // the chunk images are dead: every wave is past its last read
        __syncthreads();
        const double tot = nnf_block_sum_f64((double)csum, red);
        if (threadIdx.x == 0) ex.partial[blockIdx.x] = tot;
    }
}

template <int MT, int BM, bool VEC>
__global__ __launch_bounds__(256, (BM == BM_GEN ? 1 : ((MT <= 2 && BM == BM_FROB) ? 3 : 2))) void nnf_mu_left_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                             const float* __restrict__ Ut, int64_t ldu,
                                                             const float* __restrict__ V, int64_t ldv, int r, float beta,
                                                             const double* __restrict__ den_vec, float gamma,
                                                             float* __restrict__ Ut_out, int64_t lduo, int a_vec_ok, int n_hi,
                                                             mu_left_extra ex) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = (int)blockIdx.x;
    if (b < n_hi)
        nnf_mu_left_body<MT, BM, VEC, 4>(X, m, n, ldx, Ut, ldu, V, ldv, r, beta, den_vec, gamma, Ut_out, lduo, a_vec_ok,
                                         (int64_t)b * 256, smem, ex);
    else
        nnf_mu_left_body<MT, BM, VEC, 3>(X, m, n, ldx, Ut, ldu, V, ldv, r, beta, den_vec, gamma, Ut_out, lduo, a_vec_ok,
                                         (int64_t)n_hi * 256 + (int64_t)(b - n_hi) * 192, smem, ex);
}

// beta = 2 (Gram form): out[k][j] = max(F[k][j] * num[k][j] / (sum_l G[k][l] F[l][j]), 1e-12)
__global__ __launch_bounds__(256) void nnf_mu2_finish_kernel(const float* __restrict__ F, int64_t ldf, int r, int64_t cols,
                                                             const float* __restrict__ G, const float* __restrict__ num,
                                                             int64_t ldn, float* __restrict__ out, int64_t ldo) {
    __shared__ float Gs[NNF_MAX_RANK * NNF_MAX_RANK];
    for (int e = threadIdx.x; e < r * r; e += 256) Gs[e] = G[e];
    __syncthreads();
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < cols; j += (int64_t)gridDim.x * 256) {
        for (int k = 0; k < r; ++k) {
            float d = 0.f;
            for (int l = 0; l < r; ++l) d = fmaf(Gs[k * r + l], F[(int64_t)l * ldf + j], d);
            out[(int64_t)k * ldo + j] = fmaxf(F[(int64_t)k * ldf + j] * (num[(int64_t)k * ldn + j] / d), 1e-12f);
        }
    }
}

static float gamma_of(double beta) { return beta < 1.0 ? (float)(1.0 / (2.0 - beta)) : (beta > 2.0 ? (float)(1.0 / (beta - 1.0)) : 1.f); }

static size_t mu_shm(int MT, int r, bool regf) {   // two double-buffered chunk images (+ the resident fragments unless in registers)
    return ((regf ? 0 : (size_t)4 * ((r + 3) / 4) * 64) + (size_t)4 * MT * 256) * 16;
}

template <int MT, int BM, bool VEC>
static int launch_mu_right(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx,
                           const float* Ut, int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* V_out,
                           int64_t ldvo, hipStream_t st, float* num_out = nullptr, int64_t ldnum = 0,
                           float* den_out = nullptr, int64_t ldden = 0, double* den_vec_out = nullptr) {
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
    const size_t shm = mu_shm(MT, r, BM == BM_KL);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_mu_right_kernel<MT, BM, VEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const int grid = 8 * (int)nnf_cdiv(nsplit, 8) * ncb;
    nnf_probe(ctx, NNF_PROBE_MU_RIGHT, 0, st);
    hipLaunchKernelGGL((nnf_mu_right_kernel<MT, BM, VEC>), dim3(grid), dim3(256), shm, st, X, m, n, ldx, Ut, ldu, V, ldv, r,
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

template <int MT, int BM, bool VEC>
static int launch_mu_left(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx,
                          const float* Ut, int64_t ldu, const float* V, int64_t ldv, int r, double beta, float* Ut_out,
                          int64_t lduo, hipStream_t st, int raw_num = 0, mu_left_extra ex = mu_left_extra{nullptr, 0, 1, nullptr},
                          double* cost_out = nullptr) {
    if (64 * ldx * 4 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    double* dvec = (double*)cur.take((size_t)r * 8);
    if (!dvec) return NNF_ERR_WORKSPACE;
    const int a_vec_ok = ((((uintptr_t)V) & 15) == 0 && (ldv & 3) == 0) ? 1 : 0;
    const size_t shm = mu_shm(MT, r, BM != BM_GEN);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_mu_left_kernel<MT, BM, VEC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (BM == BM_KL || BM == BM_KLC) {  // den[k] = rowsum(V)[k]   (mu.py:86-87)
        const int rc = nnf_launch_rowsum(cur, V, ldv, r, n, dvec, st);
        if (rc != NNF_OK) return rc;
    }
    // rows per workgroup: 256 everywhere, unless one round of resident workgroups covers the matrix with 3 to 4 row tiles
    // per wave -- then n_hi workgroups of 256 rows and the rest of 192 fill exactly one round
    const int64_t slots = (int64_t)(BM == BM_GEN ? 1 : ((MT <= 2 && BM == BM_FROB) ? 3 : 2)) * ctx->num_cus, T = nnf_cdiv(m, 16);
    int64_t n_hi = nnf_cdiv(m, 256), grid = n_hi;
    if (T > 12 * slots && T <= 16 * slots) {
        n_hi = nnf_cdiv(T - 12 * slots, 4);
        grid = slots;
    }
    if (n_hi * 256 + (grid - n_hi) * 192 < m) return NNF_ERR_UNSUPPORTED;   // (cannot happen)
    if (BM == BM_FROB || BM == BM_KLC) {
        ex.partial = (double*)cur.take((size_t)grid * 8);
        if (!ex.partial || !cost_out) return NNF_ERR_WORKSPACE;
    }
    nnf_probe(ctx, NNF_PROBE_MU_LEFT, 0, st);
    hipLaunchKernelGGL((nnf_mu_left_kernel<MT, BM, VEC>), dim3((int)grid), dim3(256), shm, st, X, m, n, ldx, Ut, ldu, V, ldv, r,
                       (float)beta, dvec, raw_num ? -1.f : gamma_of(beta), Ut_out, lduo, a_vec_ok, (int)n_hi, ex);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_MU_LEFT, 1, st);
    if (BM == BM_FROB || BM == BM_KLC) return nnf_launch_sum_f64(ex.partial, grid, 1.0, cost_out, st);
    return NNF_OK;
}

#define MU_DISPATCH(FN, ...)                                                                              \
    do {                                                                                                  \
        const int MT = (r + 15) / 16;                                                                     \
        const bool vec = x_vec_ok(X, ldx);                                                                \
        const bool kl = (beta == 1.0);                                                                    \
        switch (MT) {                                                                                     \
            case 1: return kl ? (vec ? FN<1, BM_KL, true>(__VA_ARGS__) : FN<1, BM_KL, false>(__VA_ARGS__))  \
                              : (vec ? FN<1, BM_GEN, true>(__VA_ARGS__) : FN<1, BM_GEN, false>(__VA_ARGS__)); \
            case 2: return kl ? (vec ? FN<2, BM_KL, true>(__VA_ARGS__) : FN<2, BM_KL, false>(__VA_ARGS__))  \
                              : (vec ? FN<2, BM_GEN, true>(__VA_ARGS__) : FN<2, BM_GEN, false>(__VA_ARGS__)); \
            case 3: return kl ? (vec ? FN<3, BM_KL, true>(__VA_ARGS__) : FN<3, BM_KL, false>(__VA_ARGS__))  \
                              : (vec ? FN<3, BM_GEN, true>(__VA_ARGS__) : FN<3, BM_GEN, false>(__VA_ARGS__)); \
            case 4: return kl ? (vec ? FN<4, BM_KL, true>(__VA_ARGS__) : FN<4, BM_KL, false>(__VA_ARGS__))  \
                              : (vec ? FN<4, BM_GEN, true>(__VA_ARGS__) : FN<4, BM_GEN, false>(__VA_ARGS__)); \
            default: return NNF_ERR_UNSUPPORTED;                                                          \
        }                                                                                                 \
    } while (0)

static int mu_args_ok(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                      const float* V, int64_t ldv, int r, double beta, const float* out) {
    if (!ctx || !X || !Ut || !V || !out || m < 1 || n < 1 || r < 1 || ldx < n || ldu < m || ldv < n) return NNF_ERR_ARG;
    if (!(beta >= 0.0)) return NNF_ERR_ARG;
    if (r > NNF_MAX_RANK) return NNF_ERR_UNSUPPORTED;
    return NNF_OK;
}

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
        for (int l = 0; l < r; ++l) s = fmaf(G[k * r + l], V[(int64_t)l * ldv + j], s);   // same order as nnf_mu2_finish_kernel
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
    if ((int64_t)p * q > 16384) return NNF_ERR_UNSUPPORTED;
    return nnf_small_gemm_launch(A, lda, p, q, B, ldb, cols, out, ldo, 1, 0, 0, (hipStream_t)stream);
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
    const int MT = (R + 15) / 16;
    const bool vec = x_vec_ok(T, K);
#define CP3PC(MTV)                                                                                                         \
    return vec ? launch_mu_left<MTV, BM_FROB, true>(ctx, cur, T, m, K, K, Ft0, ld0, Ft2, ld2, R, 2.0, Y, m, st, 1, ex, cost_f64) \
               : launch_mu_left<MTV, BM_FROB, false>(ctx, cur, T, m, K, K, Ft0, ld0, Ft2, ld2, R, 2.0, Y, m, st, 1, ex, cost_f64)
    switch (MT) {
        case 1: CP3PC(1);
        case 2: CP3PC(2);
        case 3: CP3PC(3);
        case 4: CP3PC(4);
        default: return NNF_ERR_UNSUPPORTED;
    }
#undef CP3PC
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
    const int MT = (r + 15) / 16;
    const bool vec = x_vec_ok(X, ldx);
#define KLC(MTV)                                                                                                             \
    return vec ? launch_mu_left<MTV, BM_KLC, true>(ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, 1.0, Ut_out, lduo, st, 0, ex, cost_f64) \
               : launch_mu_left<MTV, BM_KLC, false>(ctx, cur, X, m, n, ldx, Ut, ldu, V, ldv, r, 1.0, Ut_out, lduo, st, 0, ex, cost_f64)
    switch (MT) {
        case 1: KLC(1);
        case 2: KLC(2);
        case 3: KLC(3);
        case 4: KLC(4);
        default: return NNF_ERR_UNSUPPORTED;
    }
#undef KLC
}
