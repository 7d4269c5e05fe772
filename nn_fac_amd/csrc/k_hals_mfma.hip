// HALS sweeps on the matrix cores: PUSH form of the Gauss-Seidel sweep (nn_fac/update_rules/nnls.py:156-196) for the
// many-column solves (the r x m "U side"), ranks 33..104.  Round 4.
//
// Why: the lane-per-column kernel (k_hals_fast.hip) feeds its r*r/2 packed FMAs per column and sweep with the Gram through the
// scalar cache.  At rank 100 the 40 KB Gram does not fit the 16 KB scalar cache, every 32-float block comes from L2 and the
// only usable wait on out-of-order scalar loads is lgkmcnt(0): 44 us per sweep at 125000 columns where the FMA rate allows
// ~29 (DESIGN_HISTORY "Rank 100 sweeps").  And v_pk_fma_f32 delivers 0.74-0.80 of the fp32 peak, the fp32 MFMA 0.99.
//
// Formulation.  Each column keeps its SCALED RESIDUAL  acc[i] = (UtM[i] - sp - sum_j UtU[i][j] v[j]) / UtU[i][i]  (all v[j]
// current) next to v.  The reference's row update  d = max(x, -v[k]); v[k] += d  has x = acc[k]; afterwards every residual
// moves by  acc[i] -= G'[i][k] d  (G' = D^-1 UtU).  Rows are taken four at a time (a "k-block"):
//   1. the 4 residuals of the block's rows are GATHERED from the accumulator tiles into a lane-per-column layout,
//   2. the 4 row updates run there on the VALU, in the reference's order, with the 6 in-block couplings G'[k0+i][k0+j], j < i
//      applied to temporaries (18 instructions for 64 columns),
//   3. the 4 steps go back as the B operand of v_mfma_f32_16x16x4_f32 and ONE rank-4 update  acc -= G'[:, k0:k0+4] d  pushes
//      them into all rows (including the block's own: G'[k][k] = 1 brings acc[k] to its new value) -- RT x 4 MFMAs.
// So the r*r multiply-adds per column and sweep run on the matrix cores with the Gram as the A operand (one VGPR per 16 x 4
// piece, read from an LDS image in fragment order: no scalar feed, no broadcast), exact fp32 (an MFMA is a k-ordered fmaf
// chain).  Rounding: each push rounds once relative to the residual itself (as in k_hals_wave.hip); the residual is formed from
// scratch (b' - G'v, the same MFMAs) before the first sweep and every NREF sweeps.
//
// Layouts (wave = 64 columns, workgroup = 4 waves = 256 columns, 2 workgroups per CU):
//   v[k]            lane l <-> column l of the wave              ("lane = column", 4*NKB registers)
//   acc[rt][ct][i]  MFMA C/D tile: lane l holds row 16 rt + 4 (l/16) + i of column 16 ct + l%16   (RT x 4 x 4 registers)
//   gather / scatter between the two = a 4 x 4 transpose over (register, 16-lane row) with v_permlane16_swap /
//   v_permlane32_swap (tr4 below): out[ct] row g = in[g] row ct.
// The stopping rule is the lane kernel's exchange (k_hals_common.h), decided on the sweep just done.
#include "k_hals_common.h"
#include <type_traits>

#ifndef MFMA_NREF_V
#define MFMA_NREF_V 32      // sweeps between two from-scratch residuals
#endif
#ifndef MFMA_DBG
#define MFMA_DBG 0          // timing-only ablations: 1 no MFMAs, 2 no gather/scatter transposes, 4 no in-block VALU
#endif
NNF_BUILD_FLAGS(k_hals_mfma, "MFMA_NREF_V=" NNF_STR(MFMA_NREF_V) " MFMA_DBG=" NNF_STR(MFMA_DBG))

// (hipcc 7.2: __builtin_bit_cast straight from an ext-vector element reads element 0 -- go through scalars)
__device__ __forceinline__ void pl_s16(float& x, float& y) {   // x rows 1,3 <-> y rows 0,2
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    x = __builtin_bit_cast(float, r0);
    y = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void pl_s32(float& x, float& y) {   // x rows 2,3 <-> y rows 0,1
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    x = __builtin_bit_cast(float, r0);
    y = __builtin_bit_cast(float, r1);
}
// 4 x 4 transpose over (register index, 16-lane row): afterwards t[a] row b = (old t[b]) row a
__device__ __forceinline__ void tr4(float (&t)[4]) {
    pl_s16(t[0], t[1]);
    pl_s16(t[2], t[3]);
    pl_s32(t[0], t[2]);
    pl_s32(t[1], t[3]);
}

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RT, int NKB>
struct mfma_lds {
    static constexpr int RTQ = (RT + 3) / 4;
    f32x4 img[NKB * RTQ * 64];   // [k-block][quarter q][lane] -> -G'[16 (4q+e) + lane%16][4 kb + lane/16], e = 0..3
};
// The in-block couplings come through the scalar cache (a 64-byte row per k-block: {-L10, -L20, -L21, -L30, -L31, -L32, 0, 0,
// nz0..nz3, 1/diag0..3}; 1.6 KB at rank 100: always resident) into SGPRs, fetched a block ahead; the A fragments through LDS.
// Both are hand-issued (k_hals_fast.hip explains why: hipcc sinks a load to its use, and here it also hoists every
// loop-invariant LDS read of the sweep out of the sweep loop -- 250 registers of Gram): `asm volatile` loads, one
// `s_waitcnt lgkmcnt(0)` per k-block that names every destination.  Straight-line code between issue and wait.
template <int RTQ>
__device__ __forceinline__ void mfma_issue_af(f32x4 (&af)[RTQ], unsigned addr, int kb) {
#pragma unroll
    for (int q = 0; q < RTQ; ++q)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[q]) : "v"(addr), "i"((kb * RTQ + q) * 1024));
}
template <int RTQ, class LT>
__device__ __forceinline__ void mfma_wait(f32x4 (&af)[RTQ], LT& lt) {
    static_assert(RTQ == 1 || RTQ == 2, "one or two quarters of row tiles");
    if constexpr (RTQ == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+s"(lt));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+s"(lt));
}
// acc += a b on the matrix core, the accumulator tile pinned to the ACCUMULATOR registers (AGPRs): as a builtin hipcc keeps the
// tiles in VGPRs, renames them at every MFMA (untied three-address form) and spills hundreds of registers at RT >= 4.  The asm
// form is opaque to hipcc's hazard recogniser: the callers keep the required distances themselves (mfma_pad below).
__device__ __forceinline__ void mfma_acc(f32x4& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// wait states: (a) a VALU / permlane result read as an MFMA A/B operand: 2; (b) an MFMA result read by v_accvgpr_read: the
// 8-pass v_mfma_f32_16x16x4_f32 needs 11 instructions (or nops) between -- every k-block puts (RT-1)*4 other MFMAs between a
// tile's MFMAs and the next block's gather of it (the tile the next block gathers from is pushed first); mfma_pad adds the rest.
template <int N>
__device__ __forceinline__ void mfma_nop() {
    if constexpr (N > 0) asm volatile("s_nop %0" ::"n"(N - 1));
}
template <int RTQ>
__device__ __forceinline__ void mfma_wait_af(f32x4 (&af)[RTQ]) {
    if constexpr (RTQ == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
}

// One Gauss-Seidel sweep over the wave's 64 columns.  Returns this lane's (= column's) sum of squared steps.
// ltab: the coupling table (global, 16 floats per k-block); img_addr: this lane's LDS byte address inside the image.
template <int RT, int NKB, bool GUARD>
__device__ __forceinline__ float mfma_sweep(f32x4 (&acc)[RT][4], float (&v)[4 * NKB], unsigned img_addr, const float* ltab) {
    constexpr int RTQ = (RT + 3) / 4;
    using LT = typename std::conditional<GUARD, f32x16, f32x8>::type;
    const uint64_t lbase = (uint64_t)ltab;
    float nd = 0.f;
    f32x4 af[RTQ];
    LT lt[2];
    auto issue_lt = [&](LT& dst, int kb) {
        if constexpr (GUARD) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(dst) : "s"(lbase), "i"(kb * 64));
        else asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(dst) : "s"(lbase), "i"(kb * 64));
    };
    issue_lt(lt[0], 0);
    mfma_issue_af<RTQ>(af, img_addr, 0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int rt = kb / 4, g = kb % 4, cur = kb & 1;
        mfma_wait<RTQ>(af, lt[cur]);
        mfma_nop<((RT - 1) * 4 < 12) ? 12 - (RT - 1) * 4 : 0>();
        if (kb + 1 < NKB) issue_lt(lt[cur ^ 1], kb + 1);
        const LT l = lt[cur];
        // 1. gather the block's residuals: w[i] lane (ct, n) = acc[rt][ct][i] lane (g, n)
        float w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t[4] = {acc[rt][0][i], acc[rt][1][i], acc[rt][2][i], acc[rt][3][i]};
            if (!(MFMA_DBG & 2)) tr4(t);
            w[i] = t[g];
        }
        // 2. the four row updates (nnls.py:162-170), in order; l[0..5] = -G'[k0+i][k0+j], j < i
        float d[4];
        if (!(MFMA_DBG & 4)) {
            asm("v_max_f32 %0, %1, -%2" : "=v"(d[0]) : "v"(w[0]), "v"(v[4 * kb]));
            if constexpr (GUARD) d[0] *= l[8];
            const float x1 = fmaf(l[0], d[0], w[1]);
            asm("v_max_f32 %0, %1, -%2" : "=v"(d[1]) : "v"(x1), "v"(v[4 * kb + 1]));
            if constexpr (GUARD) d[1] *= l[9];
            const float x2 = fmaf(l[2], d[1], fmaf(l[1], d[0], w[2]));
            asm("v_max_f32 %0, %1, -%2" : "=v"(d[2]) : "v"(x2), "v"(v[4 * kb + 2]));
            if constexpr (GUARD) d[2] *= l[10];
            const float x3 = fmaf(l[5], d[2], fmaf(l[4], d[1], fmaf(l[3], d[0], w[3])));
            asm("v_max_f32 %0, %1, -%2" : "=v"(d[3]) : "v"(x3), "v"(v[4 * kb + 3]));
            if constexpr (GUARD) d[3] *= l[11];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[4 * kb + i] += d[i];
                nd = fmaf(d[i], d[i], nd);
            }
            // finish the block's bookkeeping here (hipcc otherwise keeps every block's steps alive to the end of the sweep)
            asm volatile("" : "+v"(v[4 * kb]), "+v"(v[4 * kb + 1]), "+v"(v[4 * kb + 2]), "+v"(v[4 * kb + 3]), "+v"(nd));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = w[i] + v[4 * kb + i];
        }
        // 3. push: acc += (-G'[:, 4kb : 4kb+4]) d ; d[ct] becomes the B operand of column tile ct
        if (!(MFMA_DBG & 2)) tr4(d);
        if (!(MFMA_DBG & 1)) {
            asm volatile("s_nop 1" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                const int r2 = (((kb + 1) / 4) % RT + j) % RT;   // the tile the next block gathers from goes first
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) mfma_acc(acc[r2][ct], af[r2 / 4][r2 % 4], d[ct]);
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct][0] += d[ct] * af[0][0];
        }
        // the fragments of the next block, into the registers the MFMAs above have read
        if (kb + 1 < NKB) mfma_issue_af<RTQ>(af, img_addr, kb + 1);
    }
    return nd;
}

// acc = (UtM - sp) / diag - G' v from scratch (before the first sweep and every MFMA_NREF_V sweeps)
template <int RT, int NKB>
__device__ __forceinline__ void mfma_residual(f32x4 (&acc)[RT][4], const float (&v)[4 * NKB], unsigned img_addr, rsrc_t rb, int voff,
                                              int ldm4, const float* ltab, float sp) {
    constexpr int RTQ = (RT + 3) / 4;
    const uint64_t lbase = (uint64_t)ltab;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float w[16];
        f32x4 di[4];   // 1/diag of the tile's rows: entries 12..15 of the k-blocks' table rows (hand-issued: not hoisted)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            di[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (4 * rt + g < NKB) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(di[g]) : "s"(lbase), "i"((4 * rt + g) * 64 + 48));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(di[0]), "+s"(di[1]), "+s"(di[2]), "+s"(di[3]));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = 16 * rt + j;
            w[j] = 0.f;
            if (row < 4 * NKB)   // (rows >= r: outside the descriptor -> 0, and 1/diag = 0)
                w[j] = (__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, row * ldm4, 0)) - sp) * di[j / 4][j % 4];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t[4] = {w[i], w[4 + i], w[8 + i], w[12 + i]};
            tr4(t);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct][i] = t[ct];
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        f32x4 af[RTQ];
        mfma_issue_af<RTQ>(af, img_addr, kb);
        float d[4] = {v[4 * kb], v[4 * kb + 1], v[4 * kb + 2], v[4 * kb + 3]};
        tr4(d);
        mfma_wait_af<RTQ>(af);
        asm volatile("s_nop 3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));   // (also: v_accvgpr_write -> MFMA C)
#pragma unroll
        for (int r2 = 0; r2 < RT; ++r2)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) mfma_acc(acc[r2][ct], af[r2 / 4][r2 % 4], d[ct]);
    }
    mfma_nop<12>();
}

template <int RT, int NKB>
__global__ __launch_bounds__(256, 2) void nnf_hals_mfma_kernel(hals_args a) {
    constexpr int RTQ = (RT + 3) / 4, RP = 4 * NKB;
    __shared__ mfma_lds<RT, NKB> L;
    __shared__ double red2[2][2][4];
    __shared__ unsigned lds_flag;
    if (threadIdx.x == 0) lds_flag = 1u;
    const int nblocks = gridDim.x;
    const int lane = threadIdx.x & 63;
    const unsigned img_addr = (unsigned)(uintptr_t)&L.img[lane];
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    {   // stage the Gram images (prepared in fragment order by nnf_hals_mfma_prep_kernel)
        const f32x4* src = reinterpret_cast<const f32x4*>(a.Mimg);
        for (int e = threadIdx.x; e < NKB * RTQ * 64; e += 256) L.img[e] = src[e];
    }
    const bool all_live = a.dinv[2 * a.rp] != 0.f;   // wave-uniform: no zero on the Gram diagonal (prep kernel)
    const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, (int)(((int64_t)(a.r - 1) * a.ldv + a.ncols) * 4), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.UtM), 0,
                                                        (int)(((int64_t)(a.r - 1) * a.ldm + a.ncols) * 4), 0x00020000);
    const rsrc_t rvs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Vsrc), 0,
                                                         (int)(((int64_t)(a.r - 1) * a.ldvs + a.ncols) * 4), 0x00020000);
    const int ldv4 = (int)(a.ldv * 4), ldm4 = (int)(a.ldm * 4), ldvs4 = (int)(a.ldvs * 4);
    const int voff0 = gtid < a.ncols ? (int)(gtid * 4) : (int)0x7ffffff0;
    float v[RP];
    f32x4 acc[RT][4];
#pragma unroll
    for (int k = 0; k < RP; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvs, voff0, k * ldvs4, 0));
    __syncthreads();

    double eps0 = 0.0, eps = 1.0;
    int done = 0;
    bool ok = true;
    if (a.mode == 0 && a.sweep0 > 0 && !hals_take_over(a.status, a.sweep0, a.delta, eps0, eps)) return;
    hals_prefetch pf;
    pf.s = 0;
    // the whole sweep loop once per variant (a branch INSIDE the loop makes hipcc keep two copies of v and of the tiles)
    auto run = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll 1
        for (int s = 1; s <= a.max_sweeps; ++s) {
            if (((s - 1) % MFMA_NREF_V) == 0) mfma_residual<RT, NKB>(acc, v, img_addr, rb, voff0, ldm4, a.Mlt, a.sp);
            const float f = mfma_sweep<RT, NKB, GUARD>(acc, v, img_addr, a.Mlt);
            const double nd = gtid < a.ncols ? (double)f : 0.0;
            const double bs = hals_block_sum1<256>(nd, red2[s & 1][0]);
            if (a.mode == 1) {
                if (threadIdx.x == 0) a.sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
                if (a.snapshots != nullptr && gtid < a.ncols) {   // V after sweep s (fire-and-forget stores)
                    float* sp_ = a.snapshots + (size_t)(s - 1) * a.snap_stride + gtid;
                    int64_t step = a.ncols;
                    asm volatile("" : "+s"(step));   // (not loop-invariant: hipcc otherwise keeps RP row pointers in SGPRs across the sweeps)
#pragma unroll
                    for (int k = 0; k < RP; ++k) {
                        if (k < a.r) *sp_ = v[k];
                        sp_ += step;
                    }
                }
                done = s;
                continue;
            }
            hals_publish(a.sy, s, nblocks, bs);
            double tot;
            ok = hals_collect1(a.sy, s, nblocks, tot, red2[s & 1][1], &lds_flag, pf);
            if (!ok) break;
            if (s == 1 && a.sweep0 == 0) eps0 = tot;
            eps = tot;
            done = s;
            if (!(eps >= a.delta * eps0)) break;   // nnls.py:156: sweep s was the last one
        }
    };
    if (all_live) run(std::false_type{});
    else run(std::true_type{});
    if (a.max_sweeps >= 1) {
#pragma unroll
        for (int k = 0; k < RP; ++k) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), rv, voff0, k * ldv4, 0);
    }
    if (a.mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.max_sweeps >= 1) {
            a.status[NNF_HALS_ST_EPS] = eps;
            a.status[NNF_HALS_ST_CNT] = (double)(a.sweep0 + done + 1);
            a.status[NNF_HALS_ST_EPS0] = eps0;
        }
        if (!ok) a.status[NNF_HALS_ST_ERR] = 1.0;
    }
}

// Gram images in the kernel's fragment order.  img[kb][q][lane][e] = -UtU[row][k] / UtU[row][row], row = 16 (4q+e) + lane%16,
// k = 4 kb + lane/16 (0 outside r x r and in rows with a zero diagonal); lt[kb] = the in-block couplings and the rows' nz flags.
__global__ __launch_bounds__(256) void nnf_hals_mfma_prep_kernel(const float* __restrict__ UtU, int64_t ldg, int r, int RT, int NKB,
                                                                 float* __restrict__ img, float* __restrict__ lt) {
    const int RTQ = (RT + 3) / 4;
    const int total = NKB * RTQ * 64 * 4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int el = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % RTQ, kb = (e >> 8) / RTQ;
        const int row = 16 * (4 * q + el) + (lane & 15), k = 4 * kb + (lane >> 4);
        float val = 0.f;
        if (row < r && k < r) {
            const float dg = UtU[(int64_t)row * ldg + row];
            if (dg != 0.f) val = -(UtU[(int64_t)row * ldg + k] * (float)(1.0 / (double)dg));
        }
        img[e] = val;
    }
    for (int e = blockIdx.x * 256 + threadIdx.x; e < NKB * 16; e += gridDim.x * 256) {
        const int kb = e / 16, j = e % 16, k0 = 4 * kb;
        // j: 0 L10, 1 L20, 2 L21, 3 L30, 4 L31, 5 L32, 8..11 nz
        static const int ri[6] = {1, 2, 2, 3, 3, 3}, ci[6] = {0, 0, 1, 0, 1, 2};
        float val = 0.f;
        if (j < 6) {
            const int row = k0 + ri[j], k = k0 + ci[j];
            if (row < r) {
                const float dg = UtU[(int64_t)row * ldg + row];
                if (dg != 0.f) val = -(UtU[(int64_t)row * ldg + k] * (float)(1.0 / (double)dg));
            }
        } else if (j >= 8 && j < 12) {
            const int row = k0 + (j - 8);
            val = (row < r && UtU[(int64_t)row * ldg + row] != 0.f) ? 1.f : 0.f;
        } else if (j >= 12) {
            const int row = k0 + (j - 12);
            const float dg = row < r ? UtU[(int64_t)row * ldg + row] : 0.f;
            val = (dg != 0.f) ? (float)(1.0 / (double)dg) : 0.f;
        }
        lt[e] = val;
    }
}

struct mfma_shape { int rt, nkb; };
static bool mfma_shape_of(int RP, mfma_shape& s) {
    switch (RP) {
        case 48: s = {3, 12}; return true;
        case 50: case 52: s = {4, 13}; return true;
        case 56: s = {4, 14}; return true;
        case 64: s = {4, 16}; return true;
        case 80: s = {5, 20}; return true;
        case 96: s = {6, 24}; return true;
        case 100: s = {7, 25}; return true;
        default: return false;
    }
}
bool nnf_hals_mfma_supported(int RP) { mfma_shape s; return mfma_shape_of(RP, s); }
size_t nnf_hals_mfma_gram_floats(int RP) {
    mfma_shape s;
    if (!mfma_shape_of(RP, s)) return 0;
    return (size_t)s.nkb * ((s.rt + 3) / 4) * 256 + (size_t)s.nkb * 16 + 64;
}

template <int RT, int NKB>
static int mfma_launch(nnf_ctx* ctx, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    static int cached = 0;
    if (cached == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_mfma_kernel<RT, NKB>, 256, 0) != hipSuccess || nb < 1) return NNF_ERR_LAUNCH;
        cached = nb > 2 ? 2 : nb;
    }
    int64_t cap = (int64_t)cached * ctx->num_cus;
    if (cap > max_blocks_cap) cap = max_blocks_cap;
    if (a.ncols < 0) { *nblocks_out = (int)cap; return NNF_OK; }
    const int64_t need = nnf_cdiv(a.ncols, 256);
    if (need > cap) return NNF_ERR_UNSUPPORTED;
    *nblocks_out = (int)need;
    hipLaunchKernelGGL((nnf_hals_mfma_kernel<RT, NKB>), dim3((int)need), dim3(256), 0, st, a);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

// a.ncols < 0: capacity query (workgroups of 256 columns that stay resident).  NNF_ERR_UNSUPPORTED: rank or column count not
// covered (the caller takes the lane kernel).  `gram` = nnf_hals_mfma_gram_floats(RP) floats of workspace.
int nnf_hals_mfma_run(nnf_ctx* ctx, int RP, const float* UtU, int64_t ldg, float* gram, hals_args a, int max_blocks_cap, int* nblocks_out,
                      hipStream_t st) {
    mfma_shape s;
    if (!mfma_shape_of(RP, s)) return NNF_ERR_UNSUPPORTED;
    if (a.ncols >= 0) {
        const int rtq = (s.rt + 3) / 4;
        float* img = gram;
        float* lt = gram + (size_t)s.nkb * rtq * 256;
        a.Mimg = img;
        a.Mlt = lt;
        a.rp = RP;
        const int total = s.nkb * rtq * 256;
        hipLaunchKernelGGL(nnf_hals_mfma_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, st, UtU, ldg, a.r, s.rt, s.nkb, img, lt);
        NNF_CHECK_LAUNCH();
    }
#define MFMA_CASE(RT_, NKB_) if (s.rt == RT_ && s.nkb == NKB_) return mfma_launch<RT_, NKB_>(ctx, a, max_blocks_cap, nblocks_out, st);
    MFMA_CASE(3, 12) MFMA_CASE(4, 13) MFMA_CASE(4, 14) MFMA_CASE(4, 16) MFMA_CASE(5, 20) MFMA_CASE(6, 24) MFMA_CASE(7, 25)
#undef MFMA_CASE
    return NNF_ERR_UNSUPPORTED;
}
