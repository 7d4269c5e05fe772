// Device side of the fused beta-divergence updates (k_mu.hip holds the launchers and the C ABI): kept apart so that one
// instantiation can be compiled alone (tools/mu_kernel_regs.sh: registers, spills, ISA of a single kernel in seconds).
#pragma once
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
// (loaded by stageK_bload, k_stream_common.h)
template <int MT>
__device__ __forceinline__ void stageK_store(f32x4* __restrict__ img, const f32x4 (&regs)[MT]) {
    const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
#pragma unroll
    for (int s4 = 0; s4 < MT; ++s4) img[(t * MT + s4) * 64 + L] = regs[s4];
}
// Which rank steps (4 ranks each) of MFMA #1 run.  MT = ceil(r / 16), so every 16-rank group but the last is full; with the
// resident fragments in registers (zero beyond r, like the chunk image) the last group runs 2 or 4 steps behind ONE
// wave-uniform flag: a test per step (4*s4 + c < KS) splits the product into 16 basic blocks and pins every LDS read of
// the image behind a full wait.  The LDS-resident form (general beta) holds exactly KS fragments per wave: exact test.
template <int MT, bool REGF>
__device__ __forceinline__ bool mu_kstep_on(int s4, int c, int KS, bool tail4) {
    if constexpr (REGF) return s4 < MT - 1 || c < 2 || tail4;
    else return 4 * s4 + c < KS;
}

// resident workgroups per CU of the left kernel: three for the small Frobenius forms (<= 168 VGPRs), one for general beta
#define MU_LEFT_WGPC(MT, REM, BM) ((BM) == BM_GEN ? 1 : (((MT) + ((REM) > 0) <= 2 && (REM) <= 2 && (BM) == BM_FROB) ? 3 : 2))
#ifndef MU_WG_PER_CU
#define MU_WG_PER_CU 2
#endif
#ifndef MU_STEP_FENCE
#define MU_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

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
// REM > 0 (KL only): rank = 16*MT + (1..REM), as in nnf_xty_kernel -- the MT full 16-rank tiles run on MFMA and the REM
// leftover ranks on the VALU pipe: their share of P seeds the accumulator of MFMA #1 (rank-1 updates from the resident V rows
// and the F_A image's extra tile), their numerator rows are lane-local dot products with R, reduced over the four row
// groups at the end.  For rank 50 that is 48 + 48 MFMAs + 64 FMAs per 16 x 64 block instead of 56 + 64 MFMAs (fp32 MFMA and
// fp32 FMA share one pipe: 2048 flops in 32 cycles either way), and 40 fewer registers: no spills at two workgroups per CU.
template <int MT, int REM, int BM, bool VEC>
__global__ __launch_bounds__(256, (BM == BM_KL ? MU_WG_PER_CU : 1)) void nnf_mu_right_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                              const float* __restrict__ Ut, int64_t ldu,
                                                              const float* __restrict__ V, int64_t ldv, int r, float beta,
                                                              float* __restrict__ snum, float* __restrict__ sden,
                                                              int64_t ldp, int ncb, int nsplit, int64_t rows_per_split,
                                                              int a_vec_ok) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(REM == 0 || BM == BM_KL, "leftover ranks on the VALU pipe: KL form only");
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);                      // tiles of the F_A image
    constexpr int NR = REM > 0 ? REM : 1;
    const int KS = REM > 0 ? 4 * MT : ((r + 3) >> 2);
    constexpr bool REGF = (BM == BM_KL);                             // resident fragments in registers / in LDS
    const bool tail4 = REM > 0 || KS > 4 * (MT - 1) + 2;
    f32x4* ldsVf = reinterpret_cast<f32x4*>(smem);                 // !REGF: [4][KS][64]: V[4s+g][jw+4jj..+3]
    f32x4* ldsA = ldsVf + (REGF ? 0 : (size_t)4 * KS * 64);          // [2][MTA*256]  F_A image of the Ut chunk
    f32x4* ldsK = ldsA + (size_t)2 * MTA * 256;                      // [2][MT*256]   F_K image of the Ut chunk
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

    auto v_row4 = [&](int k) {   // V[k][jl .. jl+3], zero beyond r x n
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < r && jl < n) {
            const float* p = V + (int64_t)k * ldv + jl;
            v[0] = p[0];
            if (jl + 1 < n) v[1] = p[1];
            if (jl + 2 < n) v[2] = p[2];
            if (jl + 3 < n) v[3] = p[3];
        }
        return v;
    };
    // resident V fragments of this wave's 64 columns: vfr[s] = V[4s+g][jl .. jl+3], s < KS (zero beyond r x n)
    f32x4 vfr[REGF ? 4 * MT : 1];
    f32x4 vrem[NR];                                                  // REM: V[16MT+rr][jl .. jl+3]
    if constexpr (REGF) {
#pragma unroll
        for (int s_ = 0; s_ < 4 * MT; ++s_) vfr[s_] = v_row4(4 * s_ + g);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) vrem[rr] = (REM > 0) ? v_row4(16 * MT + rr) : f32x4{0.f, 0.f, 0.f, 0.f};
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
    f32x4 ev[NR];     // REM: numerator rows 16MT+rr, columns jl..jl+3, partial over this lane's rows
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) ev[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xb[2][4];   // ring of two 16-row groups: group gi lives in xb[gi & 1] and is refilled with group gi + 2
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[t][c] = nnf_bload4<VEC>(rs, voff, (16 * t + c) * ldx4);
    const mu_stage stg = mu_stage_make(Ut, ldu, r, i_end);
    {
        f32x4 sa[MTA], sk[MT];
        stageA_bload<MTA>(stg, i_end, i_begin, a_vec_ok, sa);
        stageK_bload<MT>(stg, i_end, i_begin, sk);
        stageA_store<MTA>(ldsA, sa);
        stageK_store<MT>(ldsK, sk);
    }
    __syncthreads();

    // element-wise phase of one 16-row group: R = op(X, P), masked past the split's last row (0/0 otherwise)
    auto elementwise = [&](const f32x4 (&x)[4], const f32x4 (&accP)[4], int rowrem, f32x4 (&R1)[4],
                           f32x4 (&R2)[BM == BM_GEN ? 4 : 1]) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float r1, r2;
                mu_elem<BM>(x[reg][cc], accP[cc][reg], beta, r1, r2);
                const bool ok = reg < rowrem;
                R1[cc][reg] = ok ? r1 : 0.f;
                if constexpr (BM == BM_GEN) R2[cc][reg] = ok ? r2 : 0.f;
            }
    };

    if constexpr (REGF) {
        // ---- resident fragments in registers: every LDS read of the chunk images is issued by hand, one phase early ----
        // per 16-row group t:   [uv, ak of t in flight]  seed P from the leftover ranks | MFMA #1 (ak) | issue af |
        //                       element-wise, leftover numerator rows | issue uv, ak of t+1 | MFMA #2 (af) | refill X
        constexpr int UVT = MT >= 2 ? MT - 2 : 0;   // MFMA #2 tile after which the next group's leftover-rank rows are read
        const unsigned kbase = nnf_lds_addr(ldsK) + (unsigned)lane * 16u;
        const unsigned abase = nnf_lds_addr(ldsA) + (unsigned)lane * 16u;
        const unsigned ubase = nnf_lds_addr(ldsA) + (unsigned)g * 256u;      // lanes of a row group share one address
        for (int q = 0; q < nchunk; ++q) {
            const unsigned kb = kbase + (unsigned)(q & 1) * (MT * 4096u);
            const unsigned ab = abase + (unsigned)(q & 1) * (MTA * 4096u);
            const unsigned ub = ubase + (unsigned)(q & 1) * (MTA * 4096u);
            f32x4 sa[MTA], sk[MT];
            stageA_bload<MTA>(stg, i_end, i_begin + 64 * (int64_t)(q + 1), a_vec_ok, sa);
            const int soff_q = q * 64 * ldx4;
            const int64_t left64 = (i_end - i_begin) - 64 * (int64_t)q;
            const int rows_left = left64 > 64 ? 64 : (int)left64;          // rows of this chunk inside the split
            f32x4 ak[MT], af[MT], uv[NR];
            nnf_static_for<0, REM>([&](auto rr) { nnf_lds_read4<(MT * 4 + 0) * 1024 + rr * 16>(uv[rr], ub); });
            nnf_static_for<0, MT>([&](auto s4) { nnf_lds_read4<(0 * MT + s4) * 1024>(ak[s4], kb); });
            nnf_static_for<0, 4>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                f32x4 accP[4];
                if constexpr (REM > 0) {
                    nnf_lds_wait<(t == 0 ? MT : MT - 1 - UVT), REM>(uv);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {   // (scalar FMAs: a packed form keeps a splat pair per V value)
                            float p0 = uv[0][reg] * vrem[0][cc];
#pragma unroll
                            for (int rr = 1; rr < REM; ++rr) p0 = fmaf(uv[rr][reg], vrem[rr][cc], p0);
                            accP[cc][reg] = p0;
                        }
                } else {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) accP[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                nnf_static_for<0, MT>([&](auto s4c) {
                    constexpr int s4 = decltype(s4c)::value;
                    nnf_lds_wait<MT - 1 - s4, 1>(&ak[s4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (mu_kstep_on<MT, true>(s4, c, KS, tail4)) {
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) accP[cc] = MFMA16(ak[s4][c], vfr[4 * s4 + c][cc], accP[cc]);
                        }
                    }
                });
                nnf_static_for<0, MT>([&](auto mt) { nnf_lds_read4<(mt * 4 + t) * 1024>(af[mt], ab); });
                const int rowrem = rows_left - 16 * t - 4 * g;
                f32x4 R1[4], R2[1];
                elementwise(xb[t & 1], accP, rowrem, R1, R2);
                if constexpr (REM > 0) {
#pragma unroll
                    for (int rr = 0; rr < REM; ++rr)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) ev[rr][cc] = fmaf(uv[rr][reg], R1[cc][reg], ev[rr][cc]);
                    // finish the chains HERE: left alone, LLVM sinks them (and every R they consume) to the end of the chunk
#pragma unroll
                    for (int rr = 0; rr < REM; ++rr) asm volatile("" : "+v"(ev[rr]));
                }
                // MFMA #2: num[rk][j] += Ut[rk][i] * R[i][j], k = the block's 16 rows; tile by tile, and the K-image read of
                // the next group's rank tile mt goes out as soon as af[mt] is done with (its registers are free again; the
                // accumulator rides through so that the read stays behind this tile's MFMAs).  The leftover-rank rows of the
                // next group follow the last tile but one: 16 MFMAs cover their latency, and they are live for those only.
                nnf_static_for<0, MT>([&](auto mtc) {
                    constexpr int mt = decltype(mtc)::value;
                    nnf_lds_wait<(t < 3 ? MT - 1 + (mt > UVT ? REM : 0) : MT - 1 - mt), 1>(&af[mt]);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) num[mt][cc] = MFMA16(af[mt][reg], R1[cc][reg], num[mt][cc]);
                    if constexpr (t < 3) {
                        nnf_lds_read4_after<((t + 1) * MT + mt) * 1024>(ak[mt], kb, num[mt][3]);
                        if constexpr (mt == UVT)
                            nnf_static_for<0, REM>([&](auto rr) { nnf_lds_read4<(MT * 4 + t + 1) * 1024 + rr * 16>(uv[rr], ub); });
                    }
                });
#pragma unroll
                for (int c = 0; c < 4; ++c) xb[t & 1][c] = nnf_bload4<VEC>(rs, voff, soff_q + (16 * (t + 2) + c) * ldx4);
                // next chunk's images: A loaded at the top of the chunk and written here in group 1, K loaded in group 2 and
                // written after group 3 -- the fences between the groups keep the two staging sets from being live together
                if constexpr (t == 1) stageA_store<MTA>(ldsA + (size_t)((q + 1) & 1) * MTA * 256, sa);
                if constexpr (t == 2) stageK_bload<MT>(stg, i_end, i_begin + 64 * (int64_t)(q + 1), sk);
                MU_STEP_FENCE();
            });
            stageK_store<MT>(ldsK + (size_t)((q + 1) & 1) * MT * 256, sk);
            __syncthreads();
        }
    } else {
    for (int q = 0; q < nchunk; ++q) {
        const f32x4* imgA = ldsA + (size_t)(q & 1) * MTA * 256;
        const f32x4* imgK = ldsK + (size_t)(q & 1) * MT * 256;
        // next chunk's operand images: global loads now, LDS writes after this chunk's MFMAs (past the end: zeros)
        // the two images are staged through registers one after the other (A during groups 0-1, K during groups 2-3):
        // half the staging registers of loading both up front
        f32x4 sa[MTA], sk[MT];
        stageA_bload<MTA>(stg, i_end, i_begin + 64 * (int64_t)(q + 1), a_vec_ok, sa);
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
                        const f32x4 bv = ldsVf[(size_t)w * KS * 64 + (4 * s4 + c) * 64 + lane];
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) accP[cc] = MFMA16(ak[c], bv[cc], accP[cc]);
                    }
                }
            }
            const int64_t left64 = (i_end - i_begin) - 64 * (int64_t)q;
            const int rowrem = (left64 > 64 ? 64 : (int)left64) - 16 * t - 4 * g;
            f32x4 R1[4], R2[BM == BM_GEN ? 4 : 1];
            elementwise(xb[t & 1], accP, rowrem, R1, R2);
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
            if (t == 1) stageA_store<MTA>(ldsA + (size_t)((q + 1) & 1) * MTA * 256, sa);
            if (t == 2) stageK_bload<MT>(stg, i_end, i_begin + 64 * (int64_t)(q + 1), sk);
            MU_STEP_FENCE();
        }
        stageK_store<MT>(ldsK + (size_t)((q + 1) & 1) * MT * 256, sk);
        __syncthreads();
    }
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
    if constexpr (REM > 0) {   // sum the four row groups (lanes l, l^16, l^32, l^48), lanes of group 0 store
        float* sn = snum + (int64_t)ks * r * ldp;
#pragma unroll
        for (int rr = 0; rr < REM; ++rr) {
            f32x4 e = ev[rr];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = e[c];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                e[c] = x;
            }
            const int rk = 16 * MT + rr;
            if (g == 0 && rk < r && jl < ldp) *reinterpret_cast<f32x4*>(sn + (int64_t)rk * ldp + jl) = e;
        }
    }
}

// =========================================================================================================
// left update: workgroup = 64*NT rows of X (wave: NT 16-row N tiles), sweeping all columns; no split.
// =========================================================================================================
// NT = 16-row tiles per wave: a workgroup covers 64*NT rows starting at row0 (see nnf_xht_kernel for why the host mixes
// workgroups of NTH and NTH-1 tiles: one balanced round of resident workgroups instead of 391 on 512 slots).
// REM > 0: leftover ranks 16MT .. 16MT+REM-1 on the VALU pipe, as in the right kernel (not for the general-beta form).
template <int MT, int REM, int BM, bool VEC, int NT>
__device__ __forceinline__ void nnf_mu_left_body(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                 const float* __restrict__ Ut, int64_t ldu,
                                                 const float* __restrict__ V, int64_t ldv, int r, float beta,
                                                 const double* __restrict__ den_vec, float gamma,
                                                 float* __restrict__ Ut_out, int64_t lduo, int a_vec_ok, int64_t row0,
                                                 char* smem, const mu_left_extra& ex) {
    static_assert(REM == 0 || BM != BM_GEN, "leftover ranks on the VALU pipe: not for the general-beta form");
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);                      // tiles of the F_A image
    constexpr int NR = REM > 0 ? REM : 1;
    const int KS = REM > 0 ? 4 * MT : ((r + 3) >> 2);
    constexpr bool REGF = (BM != BM_GEN);                            // resident fragments in registers / in LDS
    const bool tail4 = REM > 0 || KS > 4 * (MT - 1) + 2;
    float csum = 0.f;                                                // BM_FROB: this lane's share of sum (X - UV)^2
    f32x4* ldsUf = reinterpret_cast<f32x4*>(smem);                 // !REGF: [4][KS][64]: comps nt: Ut[4s+g][i0w+16nt+ii]
    f32x4* ldsA = ldsUf + (REGF ? 0 : (size_t)4 * KS * 64);          // [2][MTA*256]  F_A image of the V chunk
    f32x4* ldsK = ldsA + (size_t)2 * MTA * 256;                      // [2][MT*256]   F_K image of the V chunk
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
    // ufr[s][nt] = Ut[4s+g][i0w + 16nt + ii], s < KS (zero beyond r x m);  REM: urem[rr][nt] = Ut[16MT+rr][i0w + 16nt + ii]
    f32x4 ufr[REGF ? 4 * MT : 1];
    f32x4 urem[NR];
    if constexpr (REGF) {
        if (BM == BM_FROB && ex.Fb != nullptr) {   // (only the cost + partial pass of NTF brings a second factor)
            // Khatri-Rao left factor generated on the fly (loop-invariant: once per wave): row i of the (A*B) x K view is
            // (i / nb, i % nb), entry k of it Ut[k][i / nb] * Fb[k][i % nb].  Buffer loads with hardware bounds checking
            // (rank rows >= r and tensor rows >= m: offset outside the descriptor -> 0), ALL of them issued before the first
            // product: the pointer form with its `k < r` / `i < m` branches ended every product in s_waitcnt vmcnt(0) --
            // two dozen dependent L2 round trips in front of a workgroup's first MFMA.
            const rsrc_t ra = nnf_make_rsrc(Ut, (uint32_t)((((int64_t)r - 1) * ldu + (m + ex.nb - 1) / ex.nb) * 4));
            const rsrc_t rb = nnf_make_rsrc(ex.Fb, (uint32_t)((((int64_t)r - 1) * ex.ldb + ex.nb) * 4));
            const int ldu4 = (int)(ldu * 4), ldb4 = (int)(ex.ldb * 4);
            int ka4[NT], kb4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t i = i0w + 16 * nt + ii;
                const int64_t a = i / ex.nb;
                ka4[nt] = (i < m) ? (int)(a * 4) : (int)0x7ffffff0;
                kb4[nt] = (i < m) ? (int)((i - a * ex.nb) * 4) : (int)0x7ffffff0;
            }
            f32x4 fa[4 * MT + NR], fb[4 * MT + NR];
#pragma unroll
            for (int s_ = 0; s_ < 4 * MT + (REM > 0 ? NR : 0); ++s_) {
                const int k = s_ < 4 * MT ? 4 * s_ + g : 16 * MT + (s_ - 4 * MT);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bool ok = nt < NT && k < r && ka4[nt < NT ? nt : 0] != (int)0x7ffffff0;
                    const int oa = ok ? k * ldu4 + ka4[nt < NT ? nt : 0] : (int)0x7ffffff0;
                    const int ob = ok ? k * ldb4 + kb4[nt < NT ? nt : 0] : (int)0x7ffffff0;
                    fa[s_][nt] = (nt < NT) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, oa, 0, 0)) : 0.f;
                    fb[s_][nt] = (nt < NT) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, ob, 0, 0)) : 0.f;
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < 4 * MT; ++s_) ufr[s_] = fa[s_] * fb[s_];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) urem[rr] = (REM > 0) ? fa[4 * MT + rr] * fb[4 * MT + rr] : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
        auto u_row = [&](int k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < r) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t i = i0w + 16 * nt + ii;
                    if (i < m) v[nt] = Ut[(int64_t)k * ldu + i];
                }
            }
            return v;
        };
#pragma unroll
        for (int s_ = 0; s_ < 4 * MT; ++s_) ufr[s_] = u_row(4 * s_ + g);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) urem[rr] = (REM > 0) ? u_row(16 * MT + rr) : f32x4{0.f, 0.f, 0.f, 0.f};
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
    f32x4 ev[NR];    // REM: numerator rows 16MT+rr of this lane's NT rows of U, partial over this lane's columns
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) ev[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xb[2][4];  // ring of two 16-column groups [group parity][nt]: X[i0w+16nt+ii][16*gi+4g .. +3]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[t][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldx4 + 64 * t);
    const mu_stage stg = mu_stage_make(V, ldv, r, n);
    {
        f32x4 sa[MTA], sk[MT];
        stageA_bload<MTA>(stg, n, 0, a_vec_ok, sa);
        stageK_bload<MT>(stg, n, 0, sk);
        stageA_store<MTA>(ldsA, sa);
        stageK_store<MT>(ldsK, sk);
    }
    __syncthreads();

    // element-wise phase of one 16-column group: R = op(X, P) (+ this group's share of the cost), masked outside the matrix
    auto elementwise = [&](const f32x4 (&x)[4], const f32x4 (&accP)[4], int colrem, f32x4 (&R1)[4],
                           f32x4 (&R2)[BM == BM_GEN ? 4 : 1]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool rowok = (16 * nt + ii) < rows;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float r1, r2;
                mu_elem<BM>(x[nt][reg], accP[nt][reg], beta, r1, r2);
                const bool ok = rowok && (reg < colrem);
                if constexpr (BM == BM_FROB) {
                    const float dd = ok ? (x[nt][reg] - accP[nt][reg]) : 0.f;
                    csum = fmaf(dd, dd, csum);
                }
                if constexpr (BM == BM_KLC) {   // beta_divergence(X, UV, 1) of the factors this update starts from
                    const float term = nnf_cost_term<NNF_COST_KL>(x[nt][reg], accP[nt][reg], 1.f);
                    csum += ok ? term : 0.f;
                }
                R1[nt][reg] = ok ? r1 : 0.f;
                if constexpr (BM == BM_GEN) R2[nt][reg] = ok ? r2 : 0.f;
            }
            // finish this tile's share of the sum HERE: left alone, LLVM sinks the whole dependent chain of a chunk (and the
            // differences it consumes) to the chunk's last block -- 256 VGPRs + spills instead of ~180; the divergence term
            // (a 12-term series and a logarithm per entry) is kept to one tile's worth of temporaries the same way
            if constexpr (BM == BM_KLC) asm volatile("" : "+v"(csum));
        }
        if constexpr (BM == BM_FROB) asm volatile("" : "+v"(csum));
    };

    if constexpr (REGF) {
        // ---- hand-issued LDS reads, one phase early: same schedule as the right kernel (see there) ----
        constexpr int UVT = MT >= 2 ? MT - 2 : 0;
        const unsigned kbase = nnf_lds_addr(ldsK) + (unsigned)lane * 16u;
        const unsigned abase = nnf_lds_addr(ldsA) + (unsigned)lane * 16u;
        const unsigned ubase = nnf_lds_addr(ldsA) + (unsigned)g * 256u;      // lanes of a column group share one address
        for (int q = 0; q < nchunk; ++q) {
            const unsigned kb = kbase + (unsigned)(q & 1) * (MT * 4096u);
            const unsigned ab = abase + (unsigned)(q & 1) * (MTA * 4096u);
            const unsigned ub = ubase + (unsigned)(q & 1) * (MTA * 4096u);
            f32x4 sa[MTA], sk[MT];
            stageA_bload<MTA>(stg, n, 64 * (int64_t)(q + 1), a_vec_ok, sa);
            const int64_t left64 = n - 64 * (int64_t)q;
            const int cols_left = left64 > 64 ? 64 : (int)left64;            // columns of this chunk inside the matrix
            f32x4 ak[MT], af[MT], vv[NR];
            nnf_static_for<0, REM>([&](auto rr) { nnf_lds_read4<(MT * 4 + 0) * 1024 + rr * 16>(vv[rr], ub); });
            nnf_static_for<0, MT>([&](auto s4) { nnf_lds_read4<(0 * MT + s4) * 1024>(ak[s4], kb); });
            nnf_static_for<0, 4>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                // MFMA #1 (transposed product): accP[nt][reg] = P[i0w+16nt+ii][64q+16t+4g+reg], seeded with the leftover ranks
                f32x4 accP[4];
                if constexpr (REM > 0) {
                    nnf_lds_wait<(t == 0 ? MT : MT - 1 - UVT), REM>(vv);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            float p0 = vv[0][reg] * urem[0][nt];
#pragma unroll
                            for (int rr = 1; rr < REM; ++rr) p0 = fmaf(vv[rr][reg], urem[rr][nt], p0);
                            accP[nt][reg] = p0;
                        }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) accP[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                nnf_static_for<0, MT>([&](auto s4c) {
                    constexpr int s4 = decltype(s4c)::value;
                    nnf_lds_wait<MT - 1 - s4, 1>(&ak[s4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (mu_kstep_on<MT, true>(s4, c, KS, tail4)) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) accP[nt] = MFMA16(ak[s4][c], ufr[4 * s4 + c][nt], accP[nt]);
                        }
                    }
                });
                nnf_static_for<0, MT>([&](auto mt) { nnf_lds_read4<(mt * 4 + t) * 1024>(af[mt], ab); });
                const int colrem = cols_left - 16 * t - 4 * g;
                f32x4 R1[4], R2[1];
                elementwise(xb[t & 1], accP, colrem, R1, R2);
                if constexpr (REM > 0) {
#pragma unroll
                    for (int rr = 0; rr < REM; ++rr)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) ev[rr][nt] = fmaf(vv[rr][reg], R1[nt][reg], ev[rr][nt]);
#pragma unroll
                    for (int rr = 0; rr < REM; ++rr) asm volatile("" : "+v"(ev[rr]));   // (see the right kernel)
                }
                // MFMA #2: num[rk][i] += V[rk][j] * R[j][i], k = the block's 16 columns; tile by tile (see the right kernel)
                nnf_static_for<0, MT>([&](auto mtc) {
                    constexpr int mt = decltype(mtc)::value;
                    nnf_lds_wait<(t < 3 ? MT - 1 + (mt > UVT ? REM : 0) : MT - 1 - mt), 1>(&af[mt]);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) num[mt][nt] = MFMA16(af[mt][reg], R1[nt][reg], num[mt][nt]);
                    if constexpr (t < 3) {
                        nnf_lds_read4_after<((t + 1) * MT + mt) * 1024>(ak[mt], kb, num[mt][NT - 1]);
                        if constexpr (mt == UVT)
                            nnf_static_for<0, REM>([&](auto rr) { nnf_lds_read4<(MT * 4 + t + 1) * 1024 + rr * 16>(vv[rr], ub); });
                    }
                });
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    xb[t & 1][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldx4 + 256 * q + 64 * (t + 2));
                if constexpr (t == 1) stageA_store<MTA>(ldsA + (size_t)((q + 1) & 1) * MTA * 256, sa);
                if constexpr (t == 2) stageK_bload<MT>(stg, n, 64 * (int64_t)(q + 1), sk);
                MU_STEP_FENCE();
            });
            stageK_store<MT>(ldsK + (size_t)((q + 1) & 1) * MT * 256, sk);
            __syncthreads();
        }
    } else {
    for (int q = 0; q < nchunk; ++q) {
        const f32x4* imgA = ldsA + (size_t)(q & 1) * MTA * 256;
        const f32x4* imgK = ldsK + (size_t)(q & 1) * MT * 256;
        f32x4 sa[MTA], sk[MT];   // staged one after the other (see the right kernel)
        stageA_bload<MTA>(stg, n, 64 * (int64_t)(q + 1), a_vec_ok, sa);
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
                        const f32x4 bu = ldsUf[(size_t)w * KS * 64 + (4 * s4 + c) * 64 + lane];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) accP[nt] = MFMA16(ak[c], bu[nt], accP[nt]);
                    }
                }
            }
            const int64_t left64 = n - 64 * (int64_t)q;
            const int colrem = (left64 > 64 ? 64 : (int)left64) - 16 * t - 4 * g;
            f32x4 R1[4], R2[BM == BM_GEN ? 4 : 1];
            elementwise(xb[t & 1], accP, colrem, R1, R2);
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
            if (t == 1) stageA_store<MTA>(ldsA + (size_t)((q + 1) & 1) * MTA * 256, sa);
            if (t == 2) stageK_bload<MT>(stg, n, 64 * (int64_t)(q + 1), sk);
            MU_STEP_FENCE();
        }
        stageK_store<MT>(ldsK + (size_t)((q + 1) & 1) * MT * 256, sk);
        __syncthreads();
    }
    }
    // epilogue: tile (mt, nt): rk = 16mt+4g+reg, i = i0w+16nt+ii
    // The update needs the old factor entry and the row's denominator per output element.  ALL of them are loaded first,
    // from addresses clamped into the factor (no branch around a load), and pinned: with a load inside the `i < m` /
    // `rk < r` branches hipcc waited for each one alone (s_waitcnt vmcnt(0) twice per element: ~100 dependent round trips at
    // the end of a kernel whose workgroups all finish together -- nothing left to hide them behind).
    const bool raw = (BM == BM_FROB) || gamma < 0.f;   // raw numerator (nnf_mu_left_num_f32; wave-uniform flag)
    float uo[MT][NT][4], dn[MT][4], uor[NR][NT], dnr[NR];
    if (!raw) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rk = 16 * mt + 4 * g + reg, rkc = rk < r ? rk : r - 1;
                if constexpr (BM != BM_GEN) dn[mt][reg] = (float)den_vec[rkc];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t i = i0w + 16 * nt + ii, ic = i < m ? i : m - 1;
                    uo[mt][nt][reg] = Ut[(int64_t)rkc * ldu + ic];
                }
            }
        if constexpr (REM > 0) {
#pragma unroll
            for (int rr = 0; rr < REM; ++rr) {
                const int rkc = (16 * MT + rr) < r ? (16 * MT + rr) : r - 1;
                dnr[rr] = (float)den_vec[rkc];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t i = i0w + 16 * nt + ii, ic = i < m ? i : m - 1;
                    uor[rr][nt] = Ut[(int64_t)rkc * ldu + ic];
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if constexpr (BM != BM_GEN) asm volatile("" : "+v"(dn[mt][reg]));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(uo[mt][nt][reg]));
            }
        if constexpr (REM > 0) {
#pragma unroll
            for (int rr = 0; rr < REM; ++rr) {
                asm volatile("" : "+v"(dnr[rr]));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(uor[rr][nt]));
            }
        }
    }
    auto finish = [&](int rk, int64_t i, float nu, float de, float old) {
        if (raw) {
            Ut_out[(int64_t)rk * lduo + i] = nu;
            return;
        }
        float ratio = nu / de;
        if (gamma != 1.f) ratio = powf(ratio, gamma);
        Ut_out[(int64_t)rk * lduo + i] = fmaxf(old * ratio, 1e-12f);
    };
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
                        if constexpr (BM == BM_GEN) d = den[mt][nt][reg]; else d = dn[mt][reg];
                        finish(rk, i, num[mt][nt][reg], d, uo[mt][nt][reg]);
                    }
                }
        }
    }
    if constexpr (REM > 0) {   // sum the four column groups (lanes l, l^16, l^32, l^48); the lanes of group 0 finish the rows
#pragma unroll
        for (int rr = 0; rr < REM; ++rr) {
            const int rk = 16 * MT + rr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float x = ev[rr][nt];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                const int64_t i = i0w + 16 * nt + ii;
                if (g == 0 && rk < r && i < m) finish(rk, i, x, dnr[rr], uor[rr][nt]);
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

template <int MT, int REM, int BM, bool VEC>
__global__ __launch_bounds__(256, MU_LEFT_WGPC(MT, REM, BM)) void nnf_mu_left_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                             const float* __restrict__ Ut, int64_t ldu,
                                                             const float* __restrict__ V, int64_t ldv, int r, float beta,
                                                             const double* __restrict__ den_vec, float gamma,
                                                             float* __restrict__ Ut_out, int64_t lduo, int a_vec_ok, int n_hi,
                                                             int n_mid, mu_left_extra ex) {
    // workgroups [0, n_hi): 256 rows (4 row tiles per wave), [n_hi, n_hi + n_mid): 192 rows (3), the rest: 128 rows (2) --
    // the host picks the mix that fills whole rounds of resident workgroups (launch_mu_left)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = (int)blockIdx.x;
    if (b < n_hi)
        nnf_mu_left_body<MT, REM, BM, VEC, 4>(X, m, n, ldx, Ut, ldu, V, ldv, r, beta, den_vec, gamma, Ut_out, lduo, a_vec_ok,
                                              (int64_t)b * 256, smem, ex);
    else if (b < n_hi + n_mid)
        nnf_mu_left_body<MT, REM, BM, VEC, 3>(X, m, n, ldx, Ut, ldu, V, ldv, r, beta, den_vec, gamma, Ut_out, lduo, a_vec_ok,
                                              (int64_t)n_hi * 256 + (int64_t)(b - n_hi) * 192, smem, ex);
    else
        nnf_mu_left_body<MT, REM, BM, VEC, 2>(X, m, n, ldx, Ut, ldu, V, ldv, r, beta, den_vec, gamma, Ut_out, lduo, a_vec_ok,
                                              (int64_t)n_hi * 256 + (int64_t)n_mid * 192 + (int64_t)(b - n_hi - n_mid) * 128,
                                              smem, ex);
}
