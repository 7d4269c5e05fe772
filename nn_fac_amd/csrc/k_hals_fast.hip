// Fast HALS sweep kernels: one lane = one column of V, column resident in VGPRs, Gram through the scalar cache.
// Compiled once per -DHALS_PART=0..3 (each part instantiates a subset of padded ranks) so the parts build in parallel.
// See k_hals.hip for the algorithm, the grid-exchange protocol and the entry points.
#include "k_hals_common.h"

#ifndef HALS_PART
#error "compile with -DHALS_PART=0..3"
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Scalar (SMEM) loads issued by hand.  SMEM returns out of order, so the only usable wait is lgkmcnt(0) and hipcc cannot
// software-pipeline such loads itself (it sinks every load to its use: one exposed scalar-cache round trip per 16 values).
// Here a 32-float block of the Gram row is fetched while the previous block is being consumed: wait(current) ->
// issue(next) -> 16 x v_pk_fma_f32 with SGPR-pair operands.  The loads are invisible to hipcc's counters (so it adds no
// waits of its own); the "+s" wait statements carry the data dependence (cdna_hip_programming.md s.5.7 form (ii)).
#define NNF_SLOAD2(d0, d1, base, off) \
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4" : "=s"(d0), "=s"(d1) : "s"(base), "i"((off) * 4), "i"((off) * 4 + 64))
#define NNF_SLOAD2D(d0, d1, dd, base, off, doff) \
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dword %2, %3, %6" \
                 : "=s"(d0), "=s"(d1), "=s"(dd) : "s"(base), "i"((off) * 4), "i"((off) * 4 + 64), "i"((doff) * 4))
#define NNF_SWAIT2(d0, d1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1))
#define NNF_SWAIT3(d0, d1, dd) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1), "+s"(dd))

// One Gauss-Seidel sweep (nnls.py:158-170) over the column held in v2 = {(v[0],v[1]), (v[2],v[3]), ...}.
//   x     = (UtM[k] - UtU[k,:].v - sp) / UtU[k,k]
//   v[k] <- max(v[k] + x, 0)            (== v[k] + max(x, -v[k]) of the reference, same rounding)
//   step  = x if not clipped else -v[k]
// Gp: padded Gram, row stride RS = 32*ceil(R/32) floats (zeros past r), followed by 1/diag (R floats, 0 = skip row).
template <int R, bool KEEPB>
__device__ __forceinline__ float hals_sweep_column(f32x2 (&v2)[R / 2], const float (&b)[KEEPB ? R : 1], rsrc_t rb, int voff,
                                                   int ldm4, const float* __restrict__ Gp, float sp) {
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
                NNF_SLOAD2D(buf[nxt][0], buf[nxt][1], dv[(k + 1) & 1], base, (k + 1) * RS, DOFF + k + 1);
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
    __shared__ double red[4 * 3];
    __shared__ unsigned lds_flag;
    const int nblocks = gridDim.x;
    const int64_t gthreads = (int64_t)nblocks * 256;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // descriptors over the whole matrices: rows >= r / columns of idle lanes fall outside num_records
    // (loads return 0, stores are dropped by the hardware range check)
    const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, (int)(((int64_t)(a.r - 1) * a.ldv + a.ncols) * 4), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.UtM), 0,
                                                        (int)(((int64_t)(a.r - 1) * a.ldm + a.ncols) * 4), 0x00020000);
    const int ldv4 = (int)(a.ldv * 4), ldm4 = (int)(a.ldm * 4);
    f32x2 v2[RP / 2];
    float b[KEEPB ? RP : 1];
    auto load_col = [&](int voff) {
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            v2[k / 2][k & 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, k * ldv4, 0));
            if constexpr (KEEPB)   // the sparsity constant is folded in once (rows >= r: di = 0, value irrelevant)
                b[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, k * ldm4, 0)) - a.sp;
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
    // mode 0, resident columns: lag-one speculation.  After sweep s the workgroup publishes its partial and goes straight
    // on to sweep s+1 in registers; only then does it collect the global sum of sweep s (published by every workgroup a
    // whole sweep ago, so the wait is normally free).  If that sum says "stop" (nnls.py:156) the registers are dropped:
    // memory still holds V after sweep s, because V is stored only after a sweep has been confirmed.  Cost: one wasted
    // sweep at the end instead of a grid-barrier stall in every sweep.  Strided mode keeps the blocking exchange.
    double eps0 = 0.0, eps = 1.0;
    int done = 0;
    bool ok = true, stopped = false;
    for (int s = 1; s <= a.max_sweeps; ++s) {
        double nd = 0.0;
        if constexpr (RES) {
            const float f = hals_sweep_column<RP, KEEPB>(v2, b, rb, voff0, ldm4, a.Gp, a.sp);
            nd = gtid < a.ncols ? (double)f : 0.0;
        } else {
            for (int64_t col = gtid; col < a.ncols; col += gthreads) {
                const int voff = (int)(col * 4);
                load_col(voff);
                nd += (double)hals_sweep_column<RP, KEEPB>(v2, b, rb, voff, ldm4, a.Gp, a.sp);
                store_col(voff);
            }
        }
        const double bs = nnf_block_sum_f64(nd, red);
        if (a.mode == 1) {
            if (threadIdx.x == 0) a.sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
            if constexpr (RES) {
                if (a.snapshots != nullptr && gtid < a.ncols) {   // V after sweep s (fire-and-forget stores)
                    float* sp_ = a.snapshots + (size_t)(s - 1) * a.snap_stride + gtid;
#pragma unroll
                    for (int k = 0; k < RP; ++k)
                        if (k < a.r) sp_[(int64_t)k * a.ncols] = v2[k / 2][k & 1];
                }
            }
            done = s;
            continue;
        }
        hals_publish(a.sy, s, nblocks, bs);
        const int c = RES ? s - 1 : s;          // sweep whose global sum is examined now
        if (c >= 1) {
            double tot;
            ok = hals_collect(a.sy, c, nblocks, tot, red, &lds_flag);
            if (!ok) break;
            if (c == 1) eps0 = tot;
            eps = tot;
            done = c;
            if (!(eps >= a.delta * eps0)) { stopped = true; break; }   // nnls.py:156: sweep c was the last one
        }
        if constexpr (RES) store_col(voff0);    // V after sweep s (sweep s-1 said "go on")
    }
    if (a.mode == 1) {
        if constexpr (RES) store_col(voff0);
    } else if (RES && ok && !stopped && a.max_sweeps >= 1) {
        double tot;                             // ran to the sweep budget: the last sweep's sum is still due
        ok = hals_collect(a.sy, a.max_sweeps, nblocks, tot, red, &lds_flag);
        if (ok) {
            if (a.max_sweeps == 1) eps0 = tot;
            eps = tot;
            done = a.max_sweeps;
        }
    }
    if (a.mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.max_sweeps >= 1) {
            a.status[NNF_HALS_ST_EPS] = eps;
            a.status[NNF_HALS_ST_CNT] = (double)(done + 1);
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
    const int64_t need = nnf_cdiv(a.ncols, 256);
    int bpc = bpc_of(true);
    if (bpc < 1) return NNF_ERR_LAUNCH;
    int64_t cap = (int64_t)bpc * ctx->num_cus;
    if (cap > max_blocks_cap) cap = max_blocks_cap;
    if (need <= cap) {
        *nblocks_out = (int)need;
        hipLaunchKernelGGL((nnf_hals_kernel<RP, true>), dim3((int)need), dim3(256), 0, st, a);
    } else {
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
    switch (RP) { HALS_CASE(52) HALS_CASE(56) HALS_CASE(64) default: return NNF_ERR_UNSUPPORTED; }
}
#elif HALS_PART == 2
int nnf_hals_fast_part2(nnf_ctx* ctx, int RP, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    switch (RP) { HALS_CASE(80) HALS_CASE(96) HALS_CASE(104) default: return NNF_ERR_UNSUPPORTED; }
}
#else
int nnf_hals_fast_part3(nnf_ctx* ctx, int RP, const hals_args& a, int max_blocks_cap, int* nblocks_out, hipStream_t st) {
    switch (RP) { HALS_CASE(112) HALS_CASE(128) default: return NNF_ERR_UNSUPPORTED; }
}
#endif
