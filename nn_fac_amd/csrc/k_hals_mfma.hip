// HALS sweeps on the matrix cores: PUSH form of the Gauss-Seidel sweep (nn_fac/update_rules/nnls.py:156-196) for the
// many-column solves (the r x m "U side"), ranks 48..100.  Round 4.
//
// Why: the lane-per-column kernel (k_hals_fast.hip) feeds its r*r/2 packed FMAs per column and sweep with the Gram through the
// scalar cache.  At rank 100 the 40 KB Gram does not fit the 16 KB scalar cache, every 32-float block comes from L2 and the
// only usable wait on out-of-order scalar loads is lgkmcnt(0): 44-46 us per sweep at 125000 columns where the FMA rate allows
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
// scratch (b' - G'v, the same MFMAs) before the first sweep and every MFMA_NREF_V sweeps.
// Leftover rows: a rank of 16 RT + REM (REM <= 4: 50 = 48 + 2, 100 = 96 + 4) keeps its last REM rows OFF the matrix cores --
// their residuals live in the lane-per-column layout and receive every block's steps through 4 REM FMAs with scalar operands
// (the 48 + 2 split of the streaming kernels): 24 instead of 28 MFMAs per block at rank 100, 12 instead of 16 at rank 50.
//
// Layouts (wave = 64 columns, workgroup = 4 waves = 256 columns, 2 workgroups per CU):
//   v[k]            lane l <-> column l of the wave              ("lane = column", 4*NKB registers)
//   acc[rt][ct][i]  MFMA C/D tile: lane l holds row 16 rt + 4 (l/16) + i of column 16 ct + l%16   (RT x 4 x 4 registers)
//   gather / scatter between the two = a 4 x 4 transpose over (register, 16-lane row) with v_permlane16_swap /
//   v_permlane32_swap: out[ct] row g = in[g] row ct.
// Schedule: software-pipelined by hand.  While the MFMAs of block kb issue (32 cycles each, 8 of them holding the vector issue
// port), the VALU work of block kb+1 -- gather from the tile that block kb pushed FIRST, row updates, scatter -- is slotted in
// between them in ten stages, pinned with empty `asm volatile` statements (hipcc otherwise clusters all of it in front of the
// MFMAs: 450 cycles of exposed VALU per 900 cycles of MFMA).  The MFMAs are `asm volatile` too, with the accumulator tied
// ("+v"): as builtins hipcc renames every tile at every MFMA (untied three-address form) and spills hundreds of registers.
// The stopping rule is the lane kernel's exchange (k_hals_common.h), decided on the sweep just done.
#include "k_hals_common.h"
#include <type_traits>

#ifndef MFMA_NREF_V
#define MFMA_NREF_V 32      // sweeps between two from-scratch residuals
#endif
#ifndef MFMA_EARLY
#define MFMA_EARLY 1        // one more from-scratch residual after the FIRST sweep of a solve (its steps are the large ones)
#endif
#ifndef MFMA_DBG
#define MFMA_DBG 0          // timing-only ablations: 1 no MFMAs, 4 no row updates (gather and scatter stay)
#endif
NNF_BUILD_FLAGS(k_hals_mfma, "MFMA_NREF_V=" NNF_STR(MFMA_NREF_V) " MFMA_EARLY=" NNF_STR(MFMA_EARLY) " MFMA_DBG=" NNF_STR(MFMA_DBG))

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
__device__ __forceinline__ void pin4(float (&t)[4]) { asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])); }

// acc += a b on the matrix core, accumulator tied.  Opaque to hipcc's hazard recogniser -- the callers keep the distances:
// (a) a VALU / permlane result read as an MFMA A/B operand: 2 wait states; (b) an MFMA result read by the VALU: the 8-pass
// v_mfma_f32_16x16x4_f32 needs ~11 -- the tile the next block gathers from is pushed first and at least four other MFMAs
// (128 cycles) follow before the gather reads it.
__device__ __forceinline__ void mfma_acc(f32x4& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void mfma_nop() {
    if constexpr (N > 0) asm volatile("s_nop %0" ::"n"(N - 1));
}

// Shape of an instantiation: RT MFMA row tiles (16 rows each), REM leftover rows (0..4) on the VALU, NKB k-blocks of 4 rows.
template <int RT, int REM, int NKB>
struct mfma_cfg {
    static_assert(REM >= 0 && REM <= 4, "leftover rows");
    static_assert(REM > 0 ? NKB == RT * 4 + 1 : (NKB <= RT * 4 && NKB > RT * 4 - 4), "k-blocks cover the tile rows (+ one block of leftover rows)");
    static constexpr int RTQ = (RT + 3) / 4;          // float4 pieces of a lane's A fragments per k-block
    static constexpr int NTB = REM > 0 ? NKB - 1 : NKB;   // k-blocks whose rows live in the tiles
    static constexpr int LTF = REM > 0 ? 32 : 16;     // floats per k-block in the coupling table
    static constexpr int IMG = NKB * RTQ * 64;        // float4 entries of the LDS image
};
// Coupling table (global memory, read through the scalar cache: <= 3.3 KB, always resident), LTF floats per k-block:
//   [0..5] -L10 -L20 -L21 -L30 -L31 -L32 (in-block couplings -G'[k0+i][k0+j], j < i)   [8..11] nz flags   [12..15] 1/diag
//   [16..31] (REM > 0) X[j][i] = -G'[16 RT + j][k0 + i]: the leftover rows' couplings to the block's steps
// A fragments: LDS image [k-block][q][lane] float4 -> -G'[16 (4q+e) + lane%16][4 kb + lane/16], e = 0..3.
// All of it hand-issued (`asm volatile` loads + one `s_waitcnt lgkmcnt(0)` per block naming every destination): hipcc sinks a
// load to its use, and hoists every loop-invariant LDS read of the sweep out of the sweep loop (250 registers of Gram).

template <int RT, int REM, int NKB, bool GUARD>
struct mfma_sweeper {
    using C = mfma_cfg<RT, REM, NKB>;
    static constexpr int RTQ = C::RTQ, NM = RT * 4;
    static constexpr int NBUF = RTQ == 1 ? 2 : 1;   // A fragments: double-buffered while they are one float4 per lane, else
                                                    // ONE set whose two pieces are refilled as the block's MFMAs release them
    using LT = typename std::conditional<GUARD, f32x16, f32x8>::type;
    f32x4 (&acc)[RT][4];
    float (&accx)[4];
    float (&v)[4 * NKB];
    unsigned img_addr;
    uint64_t lbase;
    unsigned ta, tg, ts;   // this lane's LDS byte addresses in the wave's 2 KB transpose scratch (see below)
    f32x4 af[NBUF][RTQ];
    LT lt[1];          // ONE set each: refilled for the next block right behind the row updates that read it (a second set
    f32x16 xt[1];      // overflows the scalar registers: 8 v_readlane reloads per k-block in the middle of the MFMA stream)
    f32x4 w, stp;          // the block's residuals / steps, lane = column
    float d2[2][4];        // B operands of the k-blocks, by block parity (the scatter of block kb+1 lands while block kb's are in use)
    float nd;

    // Gather and scatter go THROUGH LDS, not through the VALU (tools/probes/permlane_rate.hip: fp32 MFMAs and VALU instructions
    // never overlap on a SIMD -- even a v_mov adds its full issue time -- and a v_permlane swap costs three v_mov; the 16 copies
    // + 16 swaps per block of the register version were a third of the sweep).  The wave owns 2 KB of LDS:
    //   gather:  the 16 lanes of row group g -- the only ones that hold rows of this block; the stores run under that exec
    //            mask -- store their four registers of column tile ct at [ct][n] (ds_write_b128 x 4); lane (ct, n) reads its
    //            column's 4 residuals back with one ds_read_b128;                                           ta = lane * 16
    //   scatter: every lane stores its column's 4 steps at 1 KB + [lane] (one ds_write_b128); lane (k, n) reads step k of
    //            column 16 ct + n (ds_read_b32 x 4: 4 n + k covers 64 consecutive words).   tg = (lane % 16) * 16, ts = tg + 4 k
    // A wave's LDS operations execute in order, so no barrier and no second buffer are needed.
    __device__ __forceinline__ void issue_gather(int kb) {
        if (kb < C::NTB) {
            const int rt = kb / 4, g = kb % 4;
            uint64_t keep;   // (the wave is always fully active; the mask is saved and put back all the same)
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %6\n\ts_mov_b32 exec_hi, %7\n\t"
                         "ds_write_b128 %1, %2\n\tds_write_b128 %1, %3 offset:256\n\tds_write_b128 %1, %4 offset:512\n\t"
                         "ds_write_b128 %1, %5 offset:768\n\ts_mov_b64 exec, %0"
                         : "=&s"(keep)
                         : "v"(tg), "v"(acc[rt][0]), "v"(acc[rt][1]), "v"(acc[rt][2]), "v"(acc[rt][3]),
                           "i"(g == 0 ? 0xffffu : g == 1 ? 0xffff0000u : 0u), "i"(g == 2 ? 0xffffu : g == 3 ? 0xffff0000u : 0u));
            asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(ta));
        } else {
            w = f32x4{accx[0], accx[1], accx[2], accx[3]};
        }
    }
    __device__ __forceinline__ void issue_scatter(int kb) {   // kb: the block whose steps these are
        asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(ta), "v"(stp));
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d2[kb & 1][ct]) : "v"(ts), "i"(1024 + ct * 256));
    }
    __device__ __forceinline__ void issue_af(int buf, int q, int kb) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[buf][q]) : "v"(img_addr), "i"((kb * RTQ + q) * 1024));
    }
    __device__ __forceinline__ void issue_lt(int buf, int kb) {
        if constexpr (GUARD) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(lt[buf]) : "s"(lbase), "i"(kb * C::LTF * 4));
        else asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(lt[buf]) : "s"(lbase), "i"(kb * C::LTF * 4));
        if constexpr (REM > 0) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(xt[buf]) : "s"(lbase), "i"(kb * C::LTF * 4 + 64));
    }
    // every hand-issued load in flight lands here (scalar loads return out of order: lgkmcnt(0) is the only usable wait)
    __device__ __forceinline__ void wait_all() {
        f32x4& a0 = af[0][0];
        f32x4& a1 = af[NBUF - 1][RTQ - 1];
        if constexpr (REM > 0)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+s"(lt[0]), "+s"(xt[0]), "+v"(w), "+v"(d2[0][0]), "+v"(d2[0][1]), "+v"(d2[0][2]), "+v"(d2[0][3]),
                           "+v"(d2[1][0]), "+v"(d2[1][1]), "+v"(d2[1][2]), "+v"(d2[1][3]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+s"(lt[0]), "+v"(w), "+v"(d2[0][0]), "+v"(d2[0][1]), "+v"(d2[0][2]), "+v"(d2[0][3]), "+v"(d2[1][0]),
                           "+v"(d2[1][1]), "+v"(d2[1][2]), "+v"(d2[1][3]));
    }

    // The VALU work of k-block kb on its residuals w: the four row updates of nnls.py:162-170 in order, the bookkeeping, the
    // leftover rows' push.  Leaves the steps in stp.  l[0..5] = -G'[k0+i][k0+j], j < i.
    __device__ __forceinline__ void update(int kb) {
        const int lb = 0;
        const LT l = lt[lb];
        float s0, s1, s2, s3;
        asm("v_max_f32 %0, %1, -%2" : "=v"(s0) : "v"(w[0]), "v"(v[4 * kb]));
        if constexpr (GUARD) s0 *= l[8];
        const float x1 = fmaf(l[0], s0, w[1]);
        asm("v_max_f32 %0, %1, -%2" : "=v"(s1) : "v"(x1), "v"(v[4 * kb + 1]));
        if constexpr (GUARD) s1 *= l[9];
        const float x2 = fmaf(l[2], s1, fmaf(l[1], s0, w[2]));
        asm("v_max_f32 %0, %1, -%2" : "=v"(s2) : "v"(x2), "v"(v[4 * kb + 2]));
        if constexpr (GUARD) s2 *= l[10];
        const float x3 = fmaf(l[5], s2, fmaf(l[4], s1, fmaf(l[3], s0, w[3])));
        asm("v_max_f32 %0, %1, -%2" : "=v"(s3) : "v"(x3), "v"(v[4 * kb + 3]));
        if constexpr (GUARD) s3 *= l[11];
        stp = f32x4{s0, s1, s2, s3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * kb + i] += stp[i];
            nd = fmaf(stp[i], stp[i], nd);
        }
        if constexpr (REM > 0) {   // leftover rows: accx[j] += X[j][i] step[i]   (X = -G'; their own block included)
            const f32x16 x = xt[lb];
#pragma unroll
            for (int j = 0; j < REM; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accx[j] = fmaf(x[4 * j + i], stp[i], accx[j]);
            }
        }
        // finish here (hipcc otherwise keeps every block's steps alive to the end of the sweep)
        asm volatile("" : "+v"(v[4 * kb]), "+v"(v[4 * kb + 1]), "+v"(v[4 * kb + 2]), "+v"(v[4 * kb + 3]), "+v"(nd), "+v"(stp));
        if constexpr (REM > 0) pin4(accx);
    }

    // MFMA order of block kb: the tile the NEXT block gathers from first, then the other tiles of the same float4 piece of the
    // A fragments, then the other piece (so that a piece is released as early as possible).
    static constexpr int first_tile(int kb) { return (kb + 1 < C::NTB) ? (kb + 1) / 4 : 0; }
    static constexpr int first_piece(int kb) { return first_tile(kb) / 4; }
    static constexpr int piece_tiles(int p) { return (p == RTQ - 1) ? RT - 4 * p : 4; }
    static constexpr int tile_at(int kb, int j) {
        const int rn = first_tile(kb), p = rn / 4, np = piece_tiles(p);
        if (j == 0) return rn;
        if (j < np) {   // the other tiles of piece p, ascending
            int tl = 4 * p + (j - 1);
            return tl >= rn ? tl + 1 : tl;
        }
        return 4 * (1 - p) + (j - np);   // (RTQ == 2: the other piece)
    }

    // One Gauss-Seidel sweep over the wave's 64 columns, in two parts: head() = the first SPB k-blocks (and the row updates of
    // block SPB, which ride in block SPB-1's shadow), tail() = the rest; tail() returns this lane's (= column's) sum of squared
    // steps.  The persistent solve runs head() of sweep s+1 BEFORE it looks at the global sum of sweep s (lag-one speculation
    // on a prefix of the sweep: it touches rows [0, 4 (SPB+1)) of v, which the kernel backs up in LDS and puts back if sweep s
    // turns out to be the last one).  Nothing hand-issued is in flight between the two parts.
    // Block kb's MFMAs carry the next block's preparation in their shadow (what hides is LATENCY -- LDS round trips, scalar
    // loads -- not VALU time):   M0..M7 | gather(kb+1) out | M8..M11 | wait, row updates of kb+1, scatter out | M12.. | wait.
    static constexpr int SPB = NKB >= 12 ? 3 : 0;
    static constexpr int MG = 7;                           // gather goes out behind MFMA MG: 4 MFMAs behind the tile's own (hazard)
    static constexpr int MU = NM >= 16 ? 11 : NM - 3;      // row updates + scatter behind MFMA MU

    __device__ __forceinline__ void blocks(int first, int last) {
#pragma unroll
        for (int kb = first; kb < last; ++kb) {
            float (&d)[4] = d2[kb & 1];
            const int cb = NBUF == 2 ? (kb & 1) : 0;
            const bool more = kb + 1 < NKB;
            const int nfirst = 4 * piece_tiles(first_piece(kb));   // MFMAs on the first piece
            if (NBUF == 1 && kb > 0 && first_piece(kb) != first_piece(kb - 1)) wait_all();   // (that piece was issued last: rare)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int r2 = tile_at(kb, m / 4), ct = m % 4;
                if (!(MFMA_DBG & 1)) mfma_acc(acc[r2][ct], af[cb][r2 / 4][r2 % 4], d[ct]);
                else if (m == 0) acc[r2][ct][0] += d[ct] * af[cb][0][0];
                if (more && m == MG) issue_gather(kb + 1);
                if (more && m == MU) {
                    wait_all();
                    if (!(MFMA_DBG & 4)) update(kb + 1);
                    else stp = w;
                    issue_scatter(kb + 1);
                    if (kb + 2 < NKB) issue_lt(0, kb + 2);   // (its registers have just been read for the last time)
                }
                if (NBUF == 1 && m == nfirst - 1 && m != NM - 1) {   // the first piece is released: refill it for the next block
                    if (m < MU) wait_all();   // (else the wait above has already passed)
                    if (more) issue_af(0, first_piece(kb), kb + 1);
                }
            }
            wait_all();
            if (NBUF == 2) {
                if (kb + 2 < NKB) issue_af(cb, 0, kb + 2);
            } else if (more) {
                if (nfirst == NM) issue_af(0, first_piece(kb), kb + 1);   // (one piece only)
                else issue_af(0, 1 - first_piece(kb), kb + 1);
            }
        }
    }
    __device__ __forceinline__ void head() {
        nd = 0.f;
        issue_lt(0, 0);
#pragma unroll
        for (int q = 0; q < RTQ; ++q) issue_af(0, q, 0);
        if (NBUF == 2 && NKB > 1) issue_af(1, 0, 1);
        issue_gather(0);
        wait_all();
        update(0);
        issue_scatter(0);
        if (NKB > 1) issue_lt(0, 1);
        wait_all();
        blocks(0, SPB);
        wait_all();
    }
    __device__ __forceinline__ float tail() {
        blocks(SPB, NKB);
        mfma_nop<12>();
        return nd;
    }
};

// acc = (UtM - sp) / diag - G' v from scratch (before the first sweep and every MFMA_NREF_V sweeps)
template <int RT, int REM, int NKB>
__device__ __forceinline__ void mfma_residual(f32x4 (&acc)[RT][4], float (&accx)[4], const float (&v)[4 * NKB], unsigned img_addr,
                                              rsrc_t rb, int voff, int ldm4, const float* ltab, float sp) {
    using C = mfma_cfg<RT, REM, NKB>;
    constexpr int RTQ = C::RTQ;
    const uint64_t lbase = (uint64_t)ltab;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float w[16];
        f32x4 di[4];   // 1/diag of the tile's rows: entries 12..15 of the k-blocks' table rows (hand-issued: not hoisted)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            di[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (4 * rt + g < C::NTB) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(di[g]) : "s"(lbase), "i"((4 * rt + g) * C::LTF * 4 + 48));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(di[0]), "+s"(di[1]), "+s"(di[2]), "+s"(di[3]));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = 16 * rt + j;
            w[j] = 0.f;
            if (row < 4 * C::NTB)   // (rows >= r: outside the descriptor -> 0, and 1/diag = 0)
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
    if constexpr (REM > 0) {
        f32x4 di;
        asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(di) : "s"(lbase), "i"(C::NTB * C::LTF * 4 + 48));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            accx[j] = 0.f;
            if (j < REM)
                accx[j] = (__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, (16 * RT + j) * ldm4, 0)) - sp) * di[j];
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        f32x4 af[RTQ];
        f32x16 x;
#pragma unroll
        for (int q = 0; q < RTQ; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[q]) : "v"(img_addr), "i"((kb * RTQ + q) * 1024));
        if constexpr (REM > 0) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(x) : "s"(lbase), "i"(kb * C::LTF * 4 + 64));
        float d[4] = {v[4 * kb], v[4 * kb + 1], v[4 * kb + 2], v[4 * kb + 3]};
        if constexpr (REM > 0) {
            if constexpr (RTQ == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+s"(x));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+s"(x));
#pragma unroll
            for (int j = 0; j < REM; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) accx[j] = fmaf(x[4 * j + i], d[i], accx[j]);
        } else {
            if constexpr (RTQ == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]));
        }
        tr4(d);
        asm volatile("s_nop 3" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));   // (also: tile registers just written -> MFMA C)
#pragma unroll
        for (int r2 = 0; r2 < RT; ++r2)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) mfma_acc(acc[r2][ct], af[r2 / 4][r2 % 4], d[ct]);
    }
    mfma_nop<12>();
}

template <int RT, int REM, int NKB>
__global__ __launch_bounds__(256, 2) void nnf_hals_mfma_kernel(hals_args a) {
    using C = mfma_cfg<RT, REM, NKB>;
    constexpr int RP = 4 * NKB;
    __shared__ f32x4 img[C::IMG];
    __shared__ f32x4 tsc[4][2 * 64];   // per wave: gather [ct][n] (1 KB), scatter [lane] (1 KB)
    __shared__ f32x4 bkp[4][4][64];    // per wave: the rows of v the speculative head of a sweep moves
    __shared__ double red2[2][2][4];
    __shared__ unsigned lds_flag;
    if (threadIdx.x == 0) lds_flag = 1u;
    const int nblocks = gridDim.x;
    const int lane = threadIdx.x & 63;
    const unsigned img_addr = (unsigned)(uintptr_t)&img[lane];
    const unsigned tbase = (unsigned)(uintptr_t)&tsc[threadIdx.x >> 6][0];
    const unsigned t_ta = tbase + lane * 16, t_tg = tbase + (lane & 15) * 16, t_ts = tbase + (lane & 15) * 16 + (lane >> 4) * 4;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    {   // stage the Gram image (prepared in fragment order by nnf_hals_mfma_prep_kernel)
        const f32x4* src = reinterpret_cast<const f32x4*>(a.Mimg);
        for (int e = threadIdx.x; e < C::IMG; e += 256) img[e] = src[e];
    }
    const bool all_live = a.dinv[2 * a.rp] != 0.f;   // wave-uniform: no zero on the Gram diagonal (prep kernel)
    const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, (int)(((int64_t)(a.r - 1) * a.ldv + a.ncols) * 4), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.UtM), 0,
                                                        (int)(((int64_t)(a.r - 1) * a.ldm + a.ncols) * 4), 0x00020000);
    const rsrc_t rvs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Vsrc), 0,
                                                         (int)(((int64_t)(a.r - 1) * a.ldvs + a.ncols) * 4), 0x00020000);
    const int ldv4 = (int)(a.ldv * 4), ldm4 = (int)(a.ldm * 4), ldvs4 = (int)(a.ldvs * 4);
    const int voff0 = gtid < a.ncols ? (int)(gtid * 4) : (int)0x7ffffff0;
    if (a.max_sweeps < 1) return;   // (the entry point never launches such a solve)
    double eps0_in = 0.0, eps_in = 1.0;
    if (a.mode == 0 && a.sweep0 > 0 && !hals_take_over(a.status, a.sweep0, a.delta, eps0_in, eps_in)) return;
    __syncthreads();

    double eps0 = eps0_in, eps = eps_in;
    int done = 0;
    bool ok = true;
    hals_prefetch pf;
    pf.s = 0;
    // The whole solve once per variant, final store included (a branch INSIDE the loop, or a merge of the two variants' v
    // behind it, makes hipcc keep two copies of v and of the tiles).  The loop body runs at least once (checked above); the
    // from-scratch residual sits at its top behind a flag that hipcc cannot see through (it otherwise peels the first sweep).
    auto run = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        float v[RP];   // (each variant loads its own start values: shared ones stay live across the other variant's loop)
        f32x4 acc[RT][4];
        float accx[4] = {0.f, 0.f, 0.f, 0.f};
        int vo = voff0;
        asm volatile("" : "+v"(vo));   // (opaque: hipcc otherwise merges the two variants' loads back into one hoisted set)
#pragma unroll
        for (int k = 0; k < RP; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvs, vo, k * ldvs4, 0));
        // The residual state.  A launch starts from scratch unless the caller hands over the state its predecessor left
        // (nnf_hals_sweeps_ex_f32: blind chunks of ONE solve continue bit for bit where the last one stopped); from-scratch
        // residuals are scheduled on the ABSOLUTE sweep index (sweeps_done + s), so chunking does not move them.
        float* const st_out = a.resid_out ? a.resid_out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RT * 4 + 1) * 256 + lane * 4 : nullptr;
        int fresh = 1;
        if (a.resid_in != nullptr && (a.sweep0 % MFMA_NREF_V) != 0 && !(MFMA_EARLY && a.sweep0 == 1)) {
            const float* st_in = a.resid_in + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (RT * 4 + 1) * 256 + lane * 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = *reinterpret_cast<const f32x4*>(st_in + (rt * 4 + ct) * 256);
            const f32x4 ax = *reinterpret_cast<const f32x4*>(st_in + RT * 4 * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) accx[j] = ax[j];
            fresh = 0;
        }
        mfma_sweeper<RT, REM, NKB, GUARD> sw{acc, accx, v, img_addr, (uint64_t)a.Mlt, t_ta, t_tg, t_ts};
        constexpr int NBK = 4 * (sw.SPB + 1);   // rows of v that head() moves
        f32x4* const bk = &bkp[threadIdx.x >> 6][0][lane];
        // ONE instance of each big piece inside the loop (several call sites of head() / the residual make hipcc merge tiles
        // and v at every join: 500 spilled registers).  An iteration = [residual from scratch if scheduled] [head of sweep s]
        // [decision on sweep s-1] [tail of sweep s] [publish].  Residual and head run BEFORE the decision on the previous
        // sweep arrives -- lag-one speculation on a prefix of the sweep: the residual changes nothing that outlives the solve,
        // the head's rows of v are backed up in LDS and put back if sweep s-1 was the last one.
        int s = 1;
        bool pending = false;   // sweep s-1 is published, its global sum not looked at yet
        bool undo = false;
#pragma unroll 1
        for (;;) {
            asm volatile("" : "+v"(fresh));   // (opaque: hipcc otherwise specialises the first sweep)
            if (__builtin_amdgcn_readfirstlane(fresh)) mfma_residual<RT, REM, NKB>(acc, accx, v, img_addr, rb, voff0, ldm4, a.Mlt, a.sp);
#pragma unroll
            for (int b = 0; b < NBK / 4; ++b) bk[b * 64] = f32x4{v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]};
            sw.head();
            if (pending) {
                double tot;
                ok = hals_collect1(a.sy, s - 1, nblocks, tot, red2[(s - 1) & 1][1], &lds_flag, pf);
                if (ok) {
                    if (s == 2 && a.sweep0 == 0) eps0 = tot;
                    eps = tot;
                }
                if (!ok || !(eps >= a.delta * eps0)) {   // nnls.py:156: sweep s-1 was the last one -> sweep s does not take place:
                    undo = true;                          // its head's rows of v are taken from the backup when V is stored
                    break;                                // (assigning them to v here makes hipcc keep two copies of everything)
                }
            }
            const float f = sw.tail();
            const double nd = gtid < a.ncols ? (double)f : 0.0;
            const double bs = hals_block_sum1<256>(nd, red2[s & 1][0]);
            done = s;
            if (a.mode == 1) {
                if (threadIdx.x == 0) a.sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
                if (a.snapshots != nullptr && s > a.snap_first) {   // V after sweep s (fire-and-forget stores; rows >= r, idle lanes: outside the descriptor)
                    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.snapshots + (size_t)(s - 1 - a.snap_first) * a.snap_stride, 0,
                                                                        (int)((int64_t)a.r * a.ncols * 4), 0x00020000);
                    int so = 0;
                    const int step = (int)(a.ncols * 4);
#pragma unroll
                    for (int k = 0; k < RP; ++k) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), rs, voff0, so, 0);
                        so += step;
                        asm volatile("" : "+s"(so));   // (a running offset, not RP hoisted ones)
                    }
                }
            } else {
                hals_publish(a.sy, s, nblocks, bs);
                pending = true;
            }
            if (s >= a.max_sweeps) break;
            // from-scratch residuals are scheduled on the ABSOLUTE sweep index, so chunking does not move them
            fresh = ((a.sweep0 + s) % MFMA_NREF_V) == 0 || (MFMA_EARLY && a.sweep0 + s == 1);
            ++s;
        }
        if (a.mode == 0 && ok && done == s && pending) {   // ran to the sweep budget: the last sweep's sum is still due
            double tot;
            ok = hals_collect1(a.sy, s, nblocks, tot, red2[s & 1][1], &lds_flag, pf);
            if (ok) {
                if (s == 1 && a.sweep0 == 0) eps0 = tot;
                eps = tot;
            }
        }
        if (st_out != nullptr) {   // (the tiles are complete: the sweep ends with the MFMA -> reader distance)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<f32x4*>(st_out + (rt * 4 + ct) * 256) = acc[rt][ct];
            *reinterpret_cast<f32x4*>(st_out + RT * 4 * 256) = f32x4{accx[0], accx[1], accx[2], accx[3]};
        }
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            float val = v[k];
            if (k < NBK) {
                const float old = bk[(k / 4) * 64][k % 4];
                val = undo ? old : val;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rv, voff0, k * ldv4, 0);
        }
    };
    if (all_live) run(std::false_type{});
    else run(std::true_type{});
    if (a.mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        a.status[NNF_HALS_ST_EPS] = eps;
        a.status[NNF_HALS_ST_CNT] = (double)(a.sweep0 + done + 1);
        a.status[NNF_HALS_ST_EPS0] = eps0;
        if (!ok) a.status[NNF_HALS_ST_ERR] = 1.0;
    }
}

// Gram image + coupling table in the kernel's order (see above); g(row, k) = -UtU[row][k] / UtU[row][row], 0 outside r x r and
// in rows with a zero diagonal.
__global__ __launch_bounds__(256) void nnf_hals_mfma_prep_kernel(const float* __restrict__ UtU, int64_t ldg, int r, int RT, int REM, int NKB,
                                                                 float* __restrict__ img, float* __restrict__ lt) {
    const int RTQ = (RT + 3) / 4, LTF = REM > 0 ? 32 : 16;
    auto g = [&](int row, int k) -> float {
        if (row >= r || k >= r) return 0.f;
        const float dg = UtU[(int64_t)row * ldg + row];
        return dg != 0.f ? -(UtU[(int64_t)row * ldg + k] * (float)(1.0 / (double)dg)) : 0.f;
    };
    const int total = NKB * RTQ * 64 * 4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int el = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % RTQ, kb = (e >> 8) / RTQ;
        const int rt = 4 * q + el;
        img[e] = rt < RT ? g(16 * rt + (lane & 15), 4 * kb + (lane >> 4)) : 0.f;
    }
    for (int e = blockIdx.x * 256 + threadIdx.x; e < NKB * LTF; e += gridDim.x * 256) {
        const int kb = e / LTF, j = e % LTF, k0 = 4 * kb;
        const int ri[6] = {1, 2, 2, 3, 3, 3}, ci[6] = {0, 0, 1, 0, 1, 2};
        float val = 0.f;
        if (j < 6) {
            val = g(k0 + ri[j], k0 + ci[j]);
        } else if (j >= 8 && j < 12) {
            const int row = k0 + (j - 8);
            val = (row < r && UtU[(int64_t)row * ldg + row] != 0.f) ? 1.f : 0.f;
        } else if (j >= 12 && j < 16) {
            const int row = k0 + (j - 12);
            const float dg = row < r ? UtU[(int64_t)row * ldg + row] : 0.f;
            val = (dg != 0.f) ? (float)(1.0 / (double)dg) : 0.f;
        } else if (j >= 16) {
            val = g(16 * RT + (j - 16) / 4, k0 + (j - 16) % 4);
        }
        lt[e] = val;
    }
}

struct mfma_shape { int rt, rem, nkb; };
static bool mfma_shape_of(int RP, mfma_shape& s) {
    switch (RP) {
        case 48: s = {3, 0, 12}; return true;
        case 50: s = {3, 2, 13}; return true;
        case 52: s = {3, 4, 13}; return true;
        case 64: s = {4, 0, 16}; return true;
        case 80: s = {5, 0, 20}; return true;
        case 96: s = {6, 0, 24}; return true;
        case 100: s = {6, 4, 25}; return true;
        default: return false;
    }
}
bool nnf_hals_mfma_supported(int RP) { mfma_shape s; return mfma_shape_of(RP, s); }
size_t nnf_hals_mfma_gram_floats(int RP) {
    mfma_shape s;
    if (!mfma_shape_of(RP, s)) return 0;
    return (size_t)s.nkb * ((s.rt + 3) / 4) * 256 + (size_t)s.nkb * 32 + 64;
}

// floats of residual state a chunked solve of `ncols` columns carries from launch to launch (0: rank not covered)
size_t nnf_hals_mfma_resid_floats(int RP, int64_t ncols) {
    mfma_shape s;
    if (!mfma_shape_of(RP, s) || ncols < 1) return 0;
    return (size_t)nnf_cdiv(ncols, 256) * 4 * (s.rt * 4 + 1) * 256;
}

template <int RT, int REM, int NKB>
static int mfma_launch(nnf_ctx* ctx, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    static int cached = 0;
    if (cached == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_mfma_kernel<RT, REM, NKB>, 256, 0) != hipSuccess || nb < 1) return NNF_ERR_LAUNCH;
        cached = nb > 2 ? 2 : nb;
    }
    int64_t cap = (int64_t)cached * ctx->num_cus;
    if (cap > max_blocks_cap) cap = max_blocks_cap;
    if (a.ncols < 0) { *nblocks_out = (int)cap; return NNF_OK; }
    const int64_t need = nnf_cdiv(a.ncols, 256);
    if (need > cap) return NNF_ERR_UNSUPPORTED;
    *nblocks_out = (int)need;
    hipLaunchKernelGGL((nnf_hals_mfma_kernel<RT, REM, NKB>), dim3((int)need), dim3(256), 0, st, a);
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
        hipLaunchKernelGGL(nnf_hals_mfma_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, st, UtU, ldg, a.r, s.rt, s.rem, s.nkb, img, lt);
        NNF_CHECK_LAUNCH();
    }
#define MFMA_CASE(RT_, REM_, NKB_) \
    if (s.rt == RT_ && s.rem == REM_ && s.nkb == NKB_) return mfma_launch<RT_, REM_, NKB_>(ctx, a, max_blocks_cap, nblocks_out, st);
    MFMA_CASE(3, 0, 12) MFMA_CASE(3, 2, 13) MFMA_CASE(3, 4, 13) MFMA_CASE(4, 0, 16) MFMA_CASE(5, 0, 20) MFMA_CASE(6, 0, 24) MFMA_CASE(6, 4, 25)
#undef MFMA_CASE
    return NNF_ERR_UNSUPPORTED;
}
