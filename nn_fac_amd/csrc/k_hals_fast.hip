// Fast HALS sweep kernels: one lane = one column of V, column resident in VGPRs, Gram through the scalar cache.
// Compiled once per -DHALS_PART=0..3 (each part instantiates a subset of padded ranks) so the parts build in parallel.
// See k_hals.hip for the algorithm, the grid-exchange protocol and the entry points.
#include "k_hals_common.h"

#ifndef HALS_PART
#error "compile with -DHALS_PART=0..3"
#endif

NNF_BUILD_FLAGS(NNF_CAT(k_hals_fast, HALS_PART), "HALS_LATE_ISSUE=" NNF_STR(HALS_LATE_ISSUE) " HALS_MID_AT(R)=" NNF_STR(HALS_MID_AT(R)) " HALS_DBG=" NNF_STR(HALS_DBG))

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// Scalar (SMEM) loads issued by hand.  SMEM returns out of order, so the only usable wait is lgkmcnt(0) and hipcc cannot
// software-pipeline such loads itself (it sinks every load to its use: one exposed scalar-cache round trip per 16 values).
// Here a 32-float block of the Gram row is fetched while the previous block is being consumed: wait(current) ->
// issue(next) -> 16 x v_pk_fma_f32 with SGPR-pair operands.  The loads are invisible to hipcc's counters (so it adds no
// waits of its own); the "+s" wait statements carry the data dependence (cdna_hip_programming.md s.5.7 form (ii)).
#define NNF_SLOAD2(d0, d1, base, off) \
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4" : "=&s"(d0), "=&s"(d1) : "s"(base), "i"((off) * 4), "i"((off) * 4 + 64))
#define NNF_SLOAD2D(d0, d1, dd, base, off, doff) \
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dword %2, %3, %6" \
                 : "=&s"(d0), "=&s"(d1), "=&s"(dd) : "s"(base), "i"((off) * 4), "i"((off) * 4 + 64), "i"((doff) * 4))
#define NNF_SWAIT2(d0, d1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1))
#define NNF_SWAIT3(d0, d1, dd) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1), "+s"(dd))

// ---- scalar-cache operand buffers -------------------------------------------------------------------------
// W floats of a Gram row held in SGPRs as 16/8/4-dword pieces (s_load_dwordx16/x8/x4).  Scalar loads return out of
// order, so the only usable wait is lgkmcnt(0): the schedule below is built so that everything outstanding at a wait
// was issued at least ~half a row of FMAs earlier (tools/smem_probe.hip: ~60 ns for two x16 hits, ~105 ns from L2).
template <int W>
struct sgbuf {
    f32x16 a, b;
    f32x8 q;
    f32x4 r;
};
template <int W, bool TIED>
__device__ __forceinline__ void sg_issue(sgbuf<W>& d, uint64_t base, int off) {
    constexpr int N16 = W / 16, N8 = (W % 16) / 8, N4 = (W % 8) / 4;
    static_assert(W % 4 == 0 && N16 <= 2, "piece decomposition");
    // TIED: the destination registers are the ones that held the previous row's piece (consumed just above)
    if constexpr (TIED) {
        if constexpr (N16 >= 1) asm volatile("s_load_dwordx16 %0, %1, %2" : "+s"(d.a) : "s"(base), "i"(off * 4));
        if constexpr (N16 >= 2) asm volatile("s_load_dwordx16 %0, %1, %2" : "+s"(d.b) : "s"(base), "i"(off * 4 + 64));
        if constexpr (N8 == 1) asm volatile("s_load_dwordx8 %0, %1, %2" : "+s"(d.q) : "s"(base), "i"(off * 4 + 64 * N16));
        if constexpr (N4 == 1) asm volatile("s_load_dwordx4 %0, %1, %2" : "+s"(d.r) : "s"(base), "i"(off * 4 + 64 * N16 + 32 * N8));
    } else {
        if constexpr (N16 >= 1) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(d.a) : "s"(base), "i"(off * 4));
        if constexpr (N16 >= 2) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(d.b) : "s"(base), "i"(off * 4 + 64));
        if constexpr (N8 == 1) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(d.q) : "s"(base), "i"(off * 4 + 64 * N16));
        if constexpr (N4 == 1) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(d.r) : "s"(base), "i"(off * 4 + 64 * N16 + 32 * N8));
    }
}
// after s_waitcnt: tell the compiler the pieces now hold their loaded values (nothing may be read above this point)
template <int W>
__device__ __forceinline__ void sg_arrived(sgbuf<W>& d) {
    constexpr int N16 = W / 16, N8 = (W % 16) / 8, N4 = (W % 8) / 4;
    if constexpr (N16 >= 1) asm volatile("" : "+s"(d.a));
    if constexpr (N16 >= 2) asm volatile("" : "+s"(d.b));
    if constexpr (N8 == 1) asm volatile("" : "+s"(d.q));
    if constexpr (N4 == 1) asm volatile("" : "+s"(d.r));
}
template <int W>
__device__ __forceinline__ f32x2 sg_pair(const sgbuf<W>& d, int p) {   // floats 2p, 2p+1 of the buffer
    constexpr int N16 = W / 16, N8 = (W % 16) / 8;
    const int j = 2 * p;
    if (N16 >= 1 && j < 16) return f32x2{d.a[j], d.a[j + 1]};
    if (N16 >= 2 && j < 32) return f32x2{d.b[j - 16], d.b[j - 15]};
    const int j2 = j - 16 * N16;
    if (N8 == 1 && j2 < 8) return f32x2{d.q[j2], d.q[j2 + 1]};
    const int j3 = j2 - 8 * N8;
    return f32x2{d.r[j3], d.r[j3 + 1]};
}

// One Gauss-Seidel sweep for 32 < R <= 64 with the UtM column resident.  Operands are pre-scaled by 1/diag:
// Gs = diag(1/diag) UtU (rows with a zero diagonal are all zero) and b = (UtM - sp)/diag, so a row update is
//     x = b[k] - Gs[k,:].v ;  d = max(x, -v[k]) ;  v[k] += d ;  nodelta += d*d          (nnls.py:162-170)
// i.e. 6 vector instructions after the R/2 packed FMAs of the dot product.
// A Gram row is split into X = columns [0, XW) and Y = columns [XW, XW+24).  Y is double-buffered and fetched a whole
// row ahead; X is single-buffered and refilled for the next row as soon as its FMAs have issued, i.e. before the Y
// part and the row bookkeeping -- so the single lgkmcnt(0) per row finds both in (or nearly in) the registers.
// GUARD: some Gram diagonal is zero (rare).  Such a row must be left alone whatever it holds (nnls.py:160): d is
// multiplied by the row's nz flag (0/1) fetched with the Y part.  Without GUARD there is no per-row flag at all.
template <int R, bool GUARD, class MID>
__device__ __forceinline__ float hals_sweep_column_xy(f32x2 (&v2)[R / 2], const float (&b)[R], const float* __restrict__ Gs,
                                                      const float* __restrict__ nzp, const MID& mid) {
    constexpr int RS = 64, P = R / 2, YW = 24, XW = (R - YW + 3) & ~3, XP = XW / 2;
    static_assert(YW == 24, "the Y issue statement below is written for x16 + x8");
    static_assert(R > 32 && R <= 64 && R % 2 == 0 && XW + YW <= RS && XW >= 8, "row split");
    const uint64_t base = (uint64_t)Gs, nzb = (uint64_t)nzp;
    sgbuf<XW> X;
    sgbuf<YW> Y[2];
    f32x2 dv[2];   // (1/diag, nz) pair of the row (GUARD only)
    float nd = 0.f;
    sg_issue<XW, false>(X, base, 0);
    sg_issue<YW, false>(Y[0], base, XW);
    if constexpr (GUARD) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(dv[0]) : "s"(nzb), "i"(0));
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int cur = k & 1, nxt = cur ^ 1;
        if (k == HALS_MID_AT(R)) mid();
        if constexpr (GUARD) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dv[cur]));
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sg_arrived<XW>(X);
        sg_arrived<YW>(Y[cur]);
        if (k + 1 < R) {   // one statement, so that all of it is issued here and not wherever the scheduler sinks it
            if constexpr (GUARD)
                asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx8 %1, %3, %5\n\ts_load_dwordx2 %2, %6, %7"
                             : "=&s"(Y[nxt].a), "=&s"(Y[nxt].q), "=&s"(dv[nxt])
                             : "s"(base), "i"(((k + 1) * RS + XW) * 4), "i"(((k + 1) * RS + XW + 16) * 4), "s"(nzb),
                               "i"(8 * (k + 1)));
            else
                asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4"
                             : "=&s"(Y[nxt].a), "=&s"(Y[nxt].q)
                             : "s"(base), "i"(((k + 1) * RS + XW) * 4), "i"(((k + 1) * RS + XW + 16) * 4));
        }
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};   // two chains are enough: the other wave of the SIMD fills the FMA latency
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const f32x2 g = sg_pair<XW>(X, p);
            if (p & 1) a1 = __builtin_elementwise_fma(g, v2[p], a1);
            else a0 = __builtin_elementwise_fma(g, v2[p], a0);
        }
        asm volatile("" : "+v"(a0), "+v"(a1));   // X is consumed: its registers may be refilled
        if (k + 1 < R) sg_issue<XW, true>(X, base, (k + 1) * RS);
#pragma unroll
        for (int p = XP; p < P; ++p) {
            const f32x2 g = sg_pair<YW>(Y[cur], p - XP);
            if (p & 1) a1 = __builtin_elementwise_fma(g, v2[p], a1);
            else a0 = __builtin_elementwise_fma(g, v2[p], a0);
        }
        const f32x2 a = a0 + a1;
        const float vk = v2[k / 2][k & 1];
        const float x = b[k] - (a[0] + a[1]);
        float dl;   // max(x, -v[k]) in one instruction (fmaxf adds a canonicalising max for the negated operand)
        asm("v_max_f32 %0, %1, -%2" : "=v"(dl) : "v"(x), "v"(vk));
        if constexpr (GUARD) dl *= dv[cur][1];
        v2[k / 2][k & 1] = vk + dl;
        nd = fmaf(dl, dl, nd);
        asm volatile("" : "+v"(nd));  // finish this row's bookkeeping here (otherwise it is sunk to the end of the sweep)
    }
    return nd;
}

// One Gauss-Seidel sweep (nnls.py:158-170) over the column held in v2 = {(v[0],v[1]), (v[2],v[3]), ...}.
//   x     = (UtM[k] - UtU[k,:].v - sp) / UtU[k,k]
//   v[k] <- max(v[k] + x, 0)            (== v[k] + max(x, -v[k]) of the reference, same rounding)
//   step  = x if not clipped else -v[k]
// Gp: padded Gram, row stride RS = 32*ceil(R/32) floats (zeros past r), followed by R pairs (1/diag, nz): (0, 0) = skip row.
template <int R, bool KEEPB, class MID>
__device__ __forceinline__ float hals_sweep_column(f32x2 (&v2)[R / 2], const float (&b)[KEEPB ? R : 1], rsrc_t rb, int voff,
                                                   int ldm4, const float* __restrict__ Gp, float sp, const MID& mid) {
    constexpr int NBLK = (R + 31) / 32, RS = 32 * NBLK, P = R / 2, DOFF = R * RS;
    const uint64_t base = (uint64_t)Gp;
    f32x16 buf[2][2];
    float dv[2];
    float nd = 0.f;
    // UtM column not resident (R > 104): fetch it BPF rows ahead of its use (rows >= r: outside the descriptor -> 0)
    constexpr int BPF = 8;
    float bring[KEEPB ? 1 : BPF];
    if constexpr (!KEEPB) {
#pragma unroll
        for (int i = 0; i < BPF; ++i)
            bring[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, i * ldm4, 0));
    }
    NNF_SLOAD2D(buf[0][0], buf[0][1], dv[0], base, 0, DOFF);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (k == HALS_MID_AT(R)) mid();
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};   // four chains: FMA latency > 2 issues
        float di = 0.f;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int jb = k * NBLK + blk, cur = jb & 1, nxt = cur ^ 1;
            if (blk == 0) {
                NNF_SWAIT3(buf[cur][0], buf[cur][1], dv[k & 1]);
                di = dv[k & 1];
            } else {
                NNF_SWAIT2(buf[cur][0], buf[cur][1]);
            }
            // next block of the stream (next row's first block also brings that row's 1/diag)
            if (blk + 1 < NBLK) {
                NNF_SLOAD2(buf[nxt][0], buf[nxt][1], base, k * RS + 32 * (blk + 1));
            } else if (k + 1 < R) {
                NNF_SLOAD2D(buf[nxt][0], buf[nxt][1], dv[(k + 1) & 1], base, (k + 1) * RS, DOFF + 2 * (k + 1));
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                if (16 * blk + j < P)
                    a0 = __builtin_elementwise_fma(f32x2{buf[cur][0][2 * j], buf[cur][0][2 * j + 1]}, v2[16 * blk + j], a0);
                if (16 * blk + 8 + j < P)
                    a1 = __builtin_elementwise_fma(f32x2{buf[cur][1][2 * j], buf[cur][1][2 * j + 1]}, v2[16 * blk + 8 + j], a1);
                if (16 * blk + j + 1 < P)
                    a2 = __builtin_elementwise_fma(f32x2{buf[cur][0][2 * j + 2], buf[cur][0][2 * j + 3]}, v2[16 * blk + j + 1], a2);
                if (16 * blk + 8 + j + 1 < P)
                    a3 = __builtin_elementwise_fma(f32x2{buf[cur][1][2 * j + 2], buf[cur][1][2 * j + 3]}, v2[16 * blk + 8 + j + 1], a3);
            }
        }
        const f32x2 a = (a0 + a1) + (a2 + a3);
        const float dot = a[0] + a[1];
        float bk;
        if constexpr (KEEPB) {
            bk = b[k];
        } else {
            bk = bring[k % BPF];
            if (k + BPF < R)
                bring[k % BPF] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, (k + BPF) * ldm4, 0));
        }
        const float vk = v2[k / 2][k & 1];
        // KEEPB: the resident column already holds UtM - sp.  A zero Gram diagonal (or a padded row) has di = 0: then
        // x = 0 and the row must stay untouched whatever its sign (nnls.py:160) -> force "keep" with a scalar-side test.
        const float x = (KEEPB ? (bk - dot) : (bk - dot - sp)) * di;
        const float t = vk + x;
        const bool keep = (t > 0.f) | (__builtin_bit_cast(unsigned, di) == 0u);
        const float vn = keep ? t : 0.f;
        const float step = keep ? x : -vk;
        v2[k / 2][k & 1] = vn;
        nd = fmaf(step, step, nd);
        asm volatile("" : "+v"(nd));  // finish this row's bookkeeping here (otherwise it is sunk to the end of the sweep)
    }
    return nd;
}

template <int RP, bool RES>
__global__ __launch_bounds__(256, (RP > 104 ? 1 : 2)) void nnf_hals_kernel(hals_args a) {
    constexpr bool KEEPB = (RP <= 104);   // UtM column resident in VGPRs next to the V column (fits 256 registers)
    constexpr bool XY = (RP > 32 && RP <= 52);   // row-split sweep on operands pre-scaled by 1/diag (its 3 buffers need RP+24 SGPRs: beyond 52 the allocator starts spilling in-flight buffers)
    const bool all_live = XY && (a.dinv[2 * RP] != 0.f);   // wave-uniform: no zero on the Gram diagonal (prep kernel)
    __shared__ double red[4 * 3];
    __shared__ double red2[2][2][4];   // [sweep parity][block sum | collect][wave]: see hals_block_sum1
    __shared__ unsigned lds_flag;
    if (threadIdx.x == 0) lds_flag = 1u;   // time-out flag of the collects, armed once (read after a barrier)
    const int nblocks = gridDim.x;
    const int64_t gthreads = (int64_t)nblocks * 256;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // descriptors over the whole matrices: rows >= r / columns of idle lanes fall outside num_records
    // (loads return 0, stores are dropped by the hardware range check)
    const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, (int)(((int64_t)(a.r - 1) * a.ldv + a.ncols) * 4), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.UtM), 0,
                                                        (int)(((int64_t)(a.r - 1) * a.ldm + a.ncols) * 4), 0x00020000);
    const int ldv4 = (int)(a.ldv * 4), ldm4 = (int)(a.ldm * 4);
    // start values: a.Vsrc (resident kernel only: the column is read ONCE, before the first sweep, and V is written once, at
    // the end -- so a solve that starts from another matrix, nmf.py:415 `hals_nnls_acc(..., U_in^T)`, needs no copy of it first;
    // the streaming kernel re-reads V every sweep and gets Vsrc == V from the entry point)
    const rsrc_t rvs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Vsrc), 0,
                                                         (int)(((int64_t)(a.r - 1) * a.ldvs + a.ncols) * 4), 0x00020000);
    const int ldvs4 = (int)(a.ldvs * 4);
    f32x2 v2[RP / 2];
    float b[KEEPB ? RP : 1];
    auto sweep = [&](int voff, const auto& mid) -> float {
        if constexpr (XY) {
            return all_live ? hals_sweep_column_xy<RP, false>(v2, b, a.Gs, a.dinv, mid) : hals_sweep_column_xy<RP, true>(v2, b, a.Gs, a.dinv, mid);
        } else {
            return hals_sweep_column<RP, KEEPB>(v2, b, rb, voff, ldm4, a.Gp, a.sp, mid);
        }
    };
    auto load_col = [&](int voff) {
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            v2[k / 2][k & 1] = RES ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rvs, voff, k * ldvs4, 0))
                                   : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, k * ldv4, 0));
            if constexpr (KEEPB) {   // the sparsity constant is folded in once (rows >= r: di = 0, value irrelevant)
                b[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, k * ldm4, 0)) - a.sp;
                if constexpr (XY) b[k] *= a.dinv[2 * k];   // scaled operands (hals_sweep_column_xy)
            }
        }
    };
    auto store_col = [&](int voff) {
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            // NB: __builtin_bit_cast straight from an ext-vector element reads element 0 (hipcc 7.2): go through a scalar
            const float val = v2[k / 2][k & 1];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rv, voff, k * ldv4, 0);
        }
    };
    const int voff0 = gtid < a.ncols ? (int)(gtid * 4) : (int)0x7ffffff0;
    if constexpr (RES) load_col(voff0);
    // Sweep loop.  mode 1: fixed count, per-sweep local partials, nothing to decide.
    // mode 0, resident columns, RP <= 64 (240 VGPRs at RP = 64, no spills): lag-one speculation.  After sweep s the workgroup publishes its partial and goes
    // straight on to sweep s+1 in registers; only then does it collect the global sum of sweep s (published by every
    // workgroup a whole sweep ago, its granule loads issued a sweep ago: the wait is normally free).  If that sum says
    // "stop" (nnls.py:156) the registers hold one sweep too many and V is taken from a register copy made before the
    // sweep; V is written to memory once, at the end.  Cost: one wasted sweep at the end instead of a stall per sweep.
    // Larger ranks have no registers for the copy; storing V after every confirmed sweep instead costs more than it saves
    // (125000 x 100: 100 vs 45 us per sweep), so they -- and the strided mode -- wait for the sum of the sweep just done.
    constexpr bool BACKUP = RES && RP <= 64;
    constexpr bool SPEC = BACKUP;
    f32x2 vb[BACKUP ? RP / 2 : 1];

    double eps0 = 0.0, eps = 1.0;
    int done = 0;
    bool ok = true, stopped = false;
    if (a.mode == 0 && a.sweep0 > 0 && !hals_take_over(a.status, a.sweep0, a.delta, eps0, eps)) return;
    hals_prefetch pf;
    pf.s = 0;
    for (int s = 1; s <= a.max_sweeps; ++s) {
        double nd = 0.0;
        if constexpr (BACKUP) {
            if (a.mode == 0) {
#pragma unroll
                for (int k = 0; k < RP / 2; ++k) vb[k] = v2[k];   // V after sweep s-1
            }
        }
        if constexpr (RES) {
            float f;
            if constexpr (SPEC && (RP > 52 || !HALS_LATE_ISSUE)) {   // (RP > 52: 240+ VGPRs already, no room for the four address registers)
                if (a.mode == 0 && s >= 2) hals_collect_issue(a.sy, s - 1, nblocks, pf);   // consumed after this sweep
                f = sweep(voff0, hals_mid_none{});
            } else if constexpr (SPEC) {
                // granules of sweep s-1, consumed after this sweep: their loads go out in the middle of it (hals_mid_issue);
                // nothing to collect (fixed-count mode, first sweep): row 0 of the region is read and ignored (pf.s = 0)
                const bool want = a.mode == 0 && s >= 2;
                const unsigned long long* row = reinterpret_cast<const unsigned long long*>(a.sy.sslots) +
                                                (size_t)(want ? s - 1 : 0) * nblocks * 2;
                const int b0 = (int)threadIdx.x < nblocks ? (int)threadIdx.x : 0;
                const int b1 = (int)threadIdx.x + 256 < nblocks ? (int)threadIdx.x + 256 : 0;
                pf.s = want ? s - 1 : 0;
                f = sweep(voff0, hals_mid_issue{(unsigned long long)(row + 2 * (size_t)b0), (unsigned long long)(row + 2 * (size_t)b1), pf});
                hals_mid_wait(pf);
            } else {
                f = sweep(voff0, hals_mid_none{});
            }
            nd = gtid < a.ncols ? (double)f : 0.0;
        } else {
            for (int64_t col = gtid; col < a.ncols; col += gthreads) {
                const int voff = (int)(col * 4);
                load_col(voff);
                nd += (double)sweep(voff, hals_mid_none{});
                store_col(voff);
            }
        }
        double bs;
        bool merged = false;    // SPEC: this sweep's block sum and the collect of sweep s-1 share one barrier
        double tot_m = 0.0;
        if constexpr (RES && SPEC) {
            if (a.mode == 0 && s >= 2 && !(HALS_DBG & 6)) {
                ok = hals_sum_collect1<256>(a.sy, nd, bs, s - 1, nblocks, tot_m, red2[s & 1][0], red2[s & 1][1], &lds_flag, pf);
                merged = true;
            }
        }
        if (!merged) {
            if constexpr (RES) bs = hals_block_sum1<256>(nd, red2[s & 1][0]);   // valid in every thread; one barrier
            else bs = nnf_block_sum_f64(nd, red);
        }
        if (a.mode == 1) {
            if (threadIdx.x == 0) a.sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
            if constexpr (RES) {
                if (a.snapshots != nullptr && gtid < a.ncols && s > a.snap_first) {   // V after sweep s (fire-and-forget stores)
                    float* sp_ = a.snapshots + (size_t)(s - 1 - a.snap_first) * a.snap_stride + gtid;
#pragma unroll
                    for (int k = 0; k < RP; ++k)
                        if (k < a.r) sp_[(int64_t)k * a.ncols] = v2[k / 2][k & 1];
                }
            }
            done = s;
            continue;
        }
        if (!(HALS_DBG & 4)) hals_publish(a.sy, s, nblocks, bs);
        const int c = SPEC ? s - 1 : s;         // sweep whose global sum is examined now
        if (c >= 1 && !(HALS_DBG & 6)) {
            double tot;
            if (merged) tot = tot_m;
            else if constexpr (RES) ok = hals_collect1(a.sy, c, nblocks, tot, red2[s & 1][1], &lds_flag, pf);
            else ok = hals_collect(a.sy, c, nblocks, tot, red, &lds_flag, &pf);
            if (!ok) break;
            if (c == 1 && a.sweep0 == 0) eps0 = tot;
            eps = tot;
            done = c;
            if (!(eps >= a.delta * eps0) && !HALS_DBG) { stopped = true; break; }   // nnls.py:156: sweep c was the last one
        }
    }
    if constexpr (RES) {
        if (a.mode == 0) {
            if constexpr (BACKUP) {
                if (stopped || !ok) {   // the registers hold one sweep too many: fall back to the copy
#pragma unroll
                    for (int k = 0; k < RP / 2; ++k) v2[k] = vb[k];
                }
            }
            if (a.max_sweeps >= 1) store_col(voff0);
        }
    }
    if (a.mode == 1) {
        if constexpr (RES) store_col(voff0);
    } else if (SPEC && ok && !stopped && a.max_sweeps >= 1 && !HALS_DBG) {
        double tot;                             // ran to the sweep budget: the last sweep's sum is still due
        ok = hals_collect(a.sy, a.max_sweeps, nblocks, tot, red, &lds_flag);
        if (ok) {
            if (a.max_sweeps == 1 && a.sweep0 == 0) eps0 = tot;
            eps = tot;
            done = a.max_sweeps;
        }
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

template <int RP>
static int launch_rp(nnf_ctx* ctx, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    static int cached_bpc[2] = {0, 0};
    auto bpc_of = [&](bool res) -> int {
        int& c = cached_bpc[res ? 1 : 0];
        if (c == 0) {
            int nb = 0;
            hipError_t e = res ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_kernel<RP, true>, 256, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_kernel<RP, false>, 256, 0);
            if (e != hipSuccess || nb < 1) return -1;
            int b = nb >= 3 ? nb - 1 : nb;  // margin: the occupancy API can over-report by one block per CU
            if (b > 3) b = 3;
            c = b;
        }
        return c;
    };
    int bpc = bpc_of(true);
    if (bpc < 1) return NNF_ERR_LAUNCH;
    int64_t cap = (int64_t)bpc * ctx->num_cus;
    if (cap > max_blocks_cap) cap = max_blocks_cap;
    if (a.ncols < 0) {   // capacity query (nnf_hals_resident_columns): workgroups of 256 columns the resident kernel keeps on the chip
        *nblocks_out = (int)cap;
        return NNF_OK;
    }
    const int64_t need = nnf_cdiv(a.ncols, 256);
    if (need <= cap) {
        *nblocks_out = (int)need;
        hipLaunchKernelGGL((nnf_hals_kernel<RP, true>), dim3((int)need), dim3(256), 0, st, a);
    } else {
        if (a.snapshots != nullptr) return NNF_ERR_UNSUPPORTED;   // per-sweep snapshots are written by the resident kernel only
        bpc = bpc_of(false);
        if (bpc < 1) return NNF_ERR_LAUNCH;
        cap = (int64_t)bpc * ctx->num_cus;
        if (cap > max_blocks_cap) cap = max_blocks_cap;
        *nblocks_out = (int)cap;
        hipLaunchKernelGGL((nnf_hals_kernel<RP, false>), dim3((int)cap), dim3(256), 0, st, a);
    }
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

#define HALS_CASE(N) \
    case N:          \
        return launch_rp<N>(ctx, a, max_blocks_cap, nblocks_out, st);

#if HALS_PART == 0
int nnf_hals_fast_part0(nnf_ctx* ctx, int RP, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    switch (RP) { HALS_CASE(8) HALS_CASE(16) HALS_CASE(24) HALS_CASE(32) HALS_CASE(40) HALS_CASE(48) default: return NNF_ERR_UNSUPPORTED; }
}
#elif HALS_PART == 1
int nnf_hals_fast_part1(nnf_ctx* ctx, int RP, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    switch (RP) { HALS_CASE(50) HALS_CASE(52) HALS_CASE(56) HALS_CASE(64) default: return NNF_ERR_UNSUPPORTED; }
}
#elif HALS_PART == 2
int nnf_hals_fast_part2(nnf_ctx* ctx, int RP, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    switch (RP) { HALS_CASE(80) HALS_CASE(96) HALS_CASE(100) HALS_CASE(104) default: return NNF_ERR_UNSUPPORTED; }
}
#else
int nnf_hals_fast_part3(nnf_ctx* ctx, int RP, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    switch (RP) { HALS_CASE(112) HALS_CASE(128) default: return NNF_ERR_UNSUPPORTED; }
}
#endif
