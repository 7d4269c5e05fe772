// HALS-NNLS sweeps for FEW columns (the r x n "V side" of NMF, the I_mode x R factors of NTF): four lanes per column.
//
// With one lane per column (k_hals_fast.hip) a 2000-column solve is 32 lone waves, each issuing ~r*(r/2 + 16) vector
// instructions per sweep at the single-wave issue rate (tools/valu_probe.hip: 2.7 ns per v_pk_fma_f32) while 248 CUs
// idle.  Here a column is spread over the four lanes of a quad (CH = ceil(r/4)): lane q holds rows [q*CH, (q+1)*CH) of the column (and of
// its UtM column), so a wave covers 16 columns, the row dot product costs CH/2 packed FMAs per lane plus a two-step DPP
// quad reduction, and there are 4x as many waves, one per workgroup, spread over 4x as many CUs.
//
//   * the Gram operand differs between the lanes of a quad, so it cannot come from SGPRs: the row-scaled Gram
//     G' = diag(1/diag) * UtU sits in LDS (RQ x RQ floats) and every lane reads its CH-float chunk of row k with
//     CH/4 ds_read_b128 (four distinct addresses per instruction, broadcast over the 16 columns: conflict-free), one
//     row ahead of its use (hand-issued, like the scalar loads of the wide kernel);
//   * 1/diag is folded into G' and into the resident UtM chunk, so a row update is (nnls.py:162-170)
//         t  = G'[k,:].v - b'[k]               (quad-reduced; identical in the four lanes, fp add is commutative)
//         d  = max(-t, -v[k]) = -min(t, v[k])  (v[k] broadcast from its owner lane with a DPP quad_perm)
//         v[k] += d, nodelta += d*d            (owner lane only: the other three multiply by 0)
//   * rows with a zero Gram diagonal, and the padding rows k >= r, are skipped by a wave-uniform branch on a bit mask
//     (nnls.py:160), so no per-row guard arithmetic.
// Stopping rule, lag-one speculation, tagged exchange, fixed-sweep mode and snapshots are those of k_hals_fast.hip.
#include "k_hals_common.h"

typedef float f32x4q __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

#if !defined(QUAD_PART) || QUAD_PART == 0
// prep: LDS image of the row-scaled Gram.  Row k (k < RQ = 4*CH) is four chunks of CHP = roundup(CH, 4) floats; chunk q holds
// G'[k][q*CH + jj] = UtU[k][q*CH + jj] / UtU[k][k], jj < CH (0 in the padding, outside r x r and in rows with a zero
// diagonal).  Then 1/diag per row (0 = skip row), zeroed barrier word and status.
// UtU2 != nullptr: the Gram is the Hadamard product UtU .* UtU2 (the `cross` of ntf.py:442-445, formed here instead of by a
// launch of its own).
__global__ void nnf_hals_prep_quad_kernel(const float* __restrict__ UtU, const float* __restrict__ UtU2, int64_t ldg, int r, int CH,
                                          float* __restrict__ Gq, float* __restrict__ dinvq, unsigned* counter, double* status) {
    const int RQ = 4 * CH, CHP = (CH + 3) & ~3, RS = 4 * CHP;
    const int k = blockIdx.x;                   // one workgroup per image row, 0 .. RQ-1
    auto gram = [&](int a, int b) -> float {
        const float g = UtU[(int64_t)a * ldg + b];
        return UtU2 ? g * UtU2[(int64_t)a * ldg + b] : g;
    };
    const float d = (k < r) ? gram(k, k) : 0.f;
    const float di = (d != 0.f) ? (float)(1.0 / (double)d) : 0.f;
    for (int c = threadIdx.x; c < RS; c += blockDim.x) {
        const int q = c / CHP, jj = c - q * CHP, j = q * CH + jj;
        Gq[k * RS + c] = (k < r && jj < CH && j < r) ? gram(k, j) * di : 0.f;
    }
    if (threadIdx.x == 0) dinvq[k] = di;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *counter = 0u;
        if (status) {
            status[NNF_HALS_ST_EPS] = 1.0;
            status[NNF_HALS_ST_CNT] = 1.0;
            status[NNF_HALS_ST_EPS0] = 0.0;
            status[NNF_HALS_ST_ERR] = 0.0;
        }
    }
}

#endif

// Row update shared by both sweep variants: g = this lane's chunk of G'[k,:] (NP float4), q0/j = owner lane / slot of row k.
template <int CH, int K>
__device__ __forceinline__ void quad_row(const f32x4q (&g)[((CH + 3) & ~3) / 4], float (&v)[CH], const float (&b)[CH],
                                         const float (&own)[4], float& nd) {
    constexpr int q0 = K / CH, j = K % CH;
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int p = 0; 2 * p < CH; ++p) {   // pair p = chunk floats 2p, 2p+1 (an odd CH leaves a zero-padded half pair)
        const f32x2 gp = {g[p / 2][2 * (p & 1)], g[p / 2][2 * (p & 1) + 1]};
        const f32x2 vp = {v[2 * p], (2 * p + 1 < CH) ? v[2 * p + 1] : 0.f};
        if (p & 1) a1 = __builtin_elementwise_fma(gp, vp, a1);
        else a0 = __builtin_elementwise_fma(gp, vp, a0);
    }
    const f32x2 a = a0 + a1;
    float t = fmaf(-b[j], own[q0], a[0] + a[1]);   // the owner lane brings in -b'[k]
    t += dpp_quad<0xB1>(t);                          // quad_perm [1,0,3,2]
    t += dpp_quad<0x4E>(t);                          // quad_perm [2,3,0,1]: G'[k,:].v - b'[k] in all four lanes
    const float vk = dpp_quad<0x55 * q0>(v[j]);      // v[k] from its owner lane: quad_perm [q0,q0,q0,q0]
    float mn;   // d = max(-t, -v[k]) = -min(t, v[k]); one instruction (fminf adds a canonicalising max)
    asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(t), "v"(vk));
    const float dm = -mn * own[q0];
    v[j] += dm;
    nd = fmaf(dm, dm, nd);
    asm volatile("" : "+v"(nd));
}

// One Gauss-Seidel sweep over the 16 columns of the wave, every Gram diagonal non-zero (the normal case): straight-line
// code, the lane's chunk of rows k+1 and k+2 in flight behind row k (LDS returns in order: lgkmcnt(NP) = "row k is in").
// The hand-issued loads define their registers long before the data lands, which is only safe without control flow
// between issue and wait (a branch merge lets the register allocator copy a buffer that is still in flight) -- hence no
// per-row tests here: the <= 3 padding rows k >= r run as no-ops (G' row, b' and v are 0 there).
#ifndef QUAD_MID_SEL
#define QUAD_MID_SEL 1     // row at which the exchange prefetch goes out: 1 the last one, 2 the middle one, 0 never (A/B builds)
#endif
#define QUAD_MID_AT(RQ) (QUAD_MID_SEL == 1 ? (RQ) - 1 : (QUAD_MID_SEL == 2 ? (RQ) / 2 : 99999))
template <int CH, int K>
struct quad_rows {
    static constexpr int RQ = 4 * CH, CHP = (CH + 3) & ~3, NP = CHP / 4, RS4 = 4 * CHP * 4;
    template <class MID>
    static __device__ __forceinline__ void run(f32x4q (&g)[3][NP], float (&v)[CH], const float (&b)[CH], const float (&own)[4],
                                               unsigned laddr, float& nd, const MID& mid) {
        constexpr int cur = K % 3, nx2 = (K + 2) % 3;
        if constexpr (K == QUAD_MID_AT(RQ)) mid();   // exchange prefetch of the previous sweep's sums (k_hals_common.h)
        if constexpr (K + 1 < RQ) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NP) : "memory");   // row K+1 may be in flight
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(g[cur][i]));
        if constexpr (K + 2 < RQ) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[nx2][i]) : "v"(laddr), "i"((K + 2) * RS4 + i * 16));
        }
        quad_row<CH, K>(g[cur], v, b, own, nd);
        if constexpr (K + 1 < RQ) quad_rows<CH, K + 1>::run(g, v, b, own, laddr, nd, mid);
    }
};
template <int CH, class MID>
__device__ __forceinline__ float quad_sweep_all_live(float (&v)[CH], const float (&b)[CH], const float (&own)[4], unsigned laddr,
                                                     const MID& mid) {
    constexpr int RQ = 4 * CH, CHP = (CH + 3) & ~3, NP = CHP / 4, RS4 = 4 * CHP * 4;
    f32x4q g[3][NP];
    float nd = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[0][i]) : "v"(laddr), "i"(i * 16));
    if constexpr (RQ > 1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[1][i]) : "v"(laddr), "i"(RS4 + i * 16));
    }
    quad_rows<CH, 0>::run(g, v, b, own, laddr, nd, mid);
    return nd;
}

// Same sweep when some Gram diagonal is zero (rare): those rows are skipped by a wave-uniform branch (nnls.py:160), and
// because of the branches the chunk is read with ordinary LDS loads (compiler-managed waits, no hand prefetch).
template <int CH, int K>
struct quad_rows_checked {
    static constexpr int RQ = 4 * CH, CHP = (CH + 3) & ~3, NP = CHP / 4, RS = 4 * CHP;
    static __device__ __forceinline__ void run(const float* lrow0, float (&v)[CH], const float (&b)[CH], const float (&own)[4],
                                               uint64_t nz_lo, uint64_t nz_hi, float& nd) {
        const bool live = (K < 64) ? ((nz_lo >> (K & 63)) & 1ull) != 0 : ((nz_hi >> (K & 63)) & 1ull) != 0;
        if (live) {
            f32x4q g[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) g[i] = *reinterpret_cast<const f32x4q*>(lrow0 + K * RS + 4 * i);
            quad_row<CH, K>(g, v, b, own, nd);
        }
        if constexpr (K + 1 < RQ) quad_rows_checked<CH, K + 1>::run(lrow0, v, b, own, nz_lo, nz_hi, nd);
    }
};

template <int CH>
__global__ __launch_bounds__(64) void nnf_hals_quad_kernel(hals_args a) {
    constexpr int RQ = 4 * CH, CHP = (CH + 3) & ~3, RS = 4 * CHP;
    extern __shared__ __attribute__((aligned(16))) float lg[];   // RQ * RS floats (dynamic: 64 KB at r = 128)
    const int lane = threadIdx.x, q = lane & 3;
    const int nblocks = gridDim.x;
    const int64_t col = (int64_t)blockIdx.x * 16 + (lane >> 2);
    const bool valid = col < a.ncols;
    for (int e = lane; e < RQ * RS / 4; e += 64)
        reinterpret_cast<f32x4q*>(lg)[e] = reinterpret_cast<const f32x4q*>(a.Gp)[e];
    // live-row mask: bit k set <=> k < r and UtU[k][k] != 0
    const uint64_t nz_lo = __ballot(lane < RQ && a.dinv[lane < RQ ? lane : 0] != 0.f);
    const uint64_t nz_hi = (RQ > 64) ? __ballot(lane + 64 < RQ && a.dinv[lane + 64 < RQ ? lane + 64 : 0] != 0.f) : 0ull;

    const uint64_t want_lo = a.r >= 64 ? ~0ull : ((1ull << a.r) - 1ull);
    const uint64_t want_hi = a.r <= 64 ? 0ull : ((1ull << (a.r - 64)) - 1ull);
    const bool all_live = (nz_lo == want_lo) && (nz_hi == want_hi);   // wave-uniform

    const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, (int)(((int64_t)(a.r - 1) * a.ldv + a.ncols) * 4), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.UtM), 0,
                                                        (int)(((int64_t)(a.r - 1) * a.ldm + a.ncols) * 4), 0x00020000);
    const int ldv4 = (int)(a.ldv * 4), ldm4 = (int)(a.ldm * 4);
    // per-lane byte offset of row q*CH of its column; rows >= r and idle lanes land outside the descriptor (0 / dropped)
    const int voffv = valid ? (int)(col * 4 + (int64_t)q * CH * ldv4) : (int)0x7ffffff0;
    // start values: a.Vsrc (== a.V for an in-place solve)
    const rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Vsrc), 0,
                                                         (int)(((int64_t)(a.r - 1) * a.ldvs + a.ncols) * 4), 0x00020000);
    const int ldvs4 = (int)(a.ldvs * 4);
    const int voffs = valid ? (int)(col * 4 + (int64_t)q * CH * ldvs4) : (int)0x7ffffff0;
    const int voffb = valid ? (int)(col * 4 + (int64_t)q * CH * ldm4) : (int)0x7ffffff0;
    float v[CH], b[CH], own[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) own[i] = (q == i) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int kk = q * CH + j;
        v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, voffs, j * ldvs4, 0));
        const float bm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voffb, j * ldm4, 0));
        b[j] = (bm - a.sp) * a.dinv[kk];
    }
    auto store_col = [&]() {
#pragma unroll
        for (int j = 0; j < CH; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), rv, voffv, j * ldv4, 0);
    };
    __syncthreads();
    const unsigned laddr = (unsigned)(uintptr_t)lg + (unsigned)(q * CHP * 4);   // this lane's chunk in row 0

    double eps0 = 0.0, eps = 1.0;
    int done = 0;
    bool ok = true, stopped = false;
    if (a.mode == 0 && a.sweep0 > 0 && !hals_take_over(a.status, a.sweep0, a.delta, eps0, eps)) return;
    hals_prefetch pf;
    pf.s = 0;
    float vb[CH];   // V after the last confirmed sweep (a late "stop" falls back to it; V is stored once, at the end)
    for (int s = 1; s <= a.max_sweeps; ++s) {
        if (a.mode == 0) {
#pragma unroll
            for (int j = 0; j < CH; ++j) vb[j] = v[j];
        }
        float f;
        if (all_live && HALS_LATE_ISSUE && nblocks <= 128) {
            // the granules of sweep s-1 (consumed after this sweep) are fetched late IN the sweep: issued before it, most
            // come back stale -- the other workgroups finish their sweep s-1 at the same moment (k_hals_common.h)
            const bool want = a.mode == 0 && s >= 2;
            const unsigned long long* row = reinterpret_cast<const unsigned long long*>(a.sy.sslots) +
                                            (size_t)(want ? s - 1 : 0) * nblocks * 2;
            const int b0 = lane < nblocks ? lane : 0, b1 = lane + 64 < nblocks ? lane + 64 : 0;
            pf.s = want ? s - 1 : 0;
            f = quad_sweep_all_live<CH>(v, b, own, laddr, hals_mid_issue{(unsigned long long)(row + 2 * (size_t)b0),
                                                                         (unsigned long long)(row + 2 * (size_t)b1), pf});
            hals_mid_wait(pf);
        } else if (all_live) {
            if (a.mode == 0 && s >= 2) hals_collect_issue(a.sy, s - 1, nblocks, pf);   // consumed after this sweep
            f = quad_sweep_all_live<CH>(v, b, own, laddr, hals_mid_none{});
        } else {
            if (a.mode == 0 && s >= 2) hals_collect_issue(a.sy, s - 1, nblocks, pf);
            f = 0.f;
            quad_rows_checked<CH, 0>::run(lg + q * CHP, v, b, own, nz_lo, nz_hi, f);
        }
        const double bs = nnf_wave_sum_f64(valid ? (double)f : 0.0);   // the workgroup IS one wave: no LDS, no barrier
        if (a.mode == 1) {
            if (threadIdx.x == 0) a.sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
            if (a.snapshots != nullptr && valid && s > a.snap_first) {   // V after sweep s (fire-and-forget stores)
                float* sp_ = a.snapshots + (size_t)(s - 1 - a.snap_first) * a.snap_stride + col;
#pragma unroll
                for (int j = 0; j < CH; ++j)
                    if (q * CH + j < a.r) sp_[(int64_t)(q * CH + j) * a.ncols] = v[j];
            }
            done = s;
            continue;
        }
        hals_publish(a.sy, s, nblocks, bs);
        const int c = s - 1;   // lag-one speculation (k_hals_fast.hip): sweep whose global sum is examined now
        if (c >= 1) {
            double tot;
            ok = hals_collect_wave(a.sy, c, nblocks, tot, &pf);
            if (!ok) break;
            if (c == 1 && a.sweep0 == 0) eps0 = tot;
            eps = tot;
            done = c;
            if (!(eps >= a.delta * eps0)) { stopped = true; break; }   // nnls.py:156: sweep c was the last one
        }
    }
    if (a.mode == 0 && (stopped || !ok)) {   // the registers hold one sweep too many
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = vb[j];
    }
    if (a.max_sweeps >= 1) store_col();
    if (a.mode == 0 && ok && !stopped && a.max_sweeps >= 1) {
        double tot;   // ran to the sweep budget: the last sweep's sum is still due
        ok = hals_collect_wave(a.sy, a.max_sweeps, nblocks, tot);
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

static int quad_ch(int r) { return r <= 128 ? (r + 3) / 4 : 0; }   // rows per lane; 0: not built
static size_t quad_lds(int ch) { return (size_t)(4 * ch) * (size_t)(4 * ((ch + 3) & ~3)) * 4; }

template <int CH>
static int quad_cap(nnf_ctx* ctx) {   // workgroups that can be co-resident (all of them must be: persistent kernel)
    static int cached = 0;
    if (cached == 0) {
        int nb = 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_hals_quad_kernel<CH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)quad_lds(CH));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_quad_kernel<CH>, 64, quad_lds(CH)) != hipSuccess || nb < 1)
            return -1;
        int b = nb >= 3 ? nb - 1 : nb;   // margin: the occupancy API can over-report by one block per CU
        if (b > 8) b = 8;
        cached = b;
    }
    return cached * ctx->num_cus;
}

// The 32 instantiations are compiled as four translation units (-DQUAD_PART=0..3, like k_hals_fast.hip); each part exports
// one residency query and one launcher for its range of CH, part 0 also holds the prep kernel's launcher and the host logic.
#define QUAD_CASE(N, FN, ...) \
    case N:                   \
        return FN<N>(__VA_ARGS__);
#ifndef QUAD_PART
#define QUAD_PART 0
#endif
NNF_BUILD_FLAGS(NNF_CAT(k_hals_quad, QUAD_PART), "QUAD_MID_SEL=" NNF_STR(QUAD_MID_SEL) " HALS_LATE_ISSUE=" NNF_STR(HALS_LATE_ISSUE))
#if QUAD_PART == 0
#define QUAD_CASES(FN, ...)                                                                                             \
    QUAD_CASE(1, FN, __VA_ARGS__) QUAD_CASE(2, FN, __VA_ARGS__) QUAD_CASE(3, FN, __VA_ARGS__) QUAD_CASE(4, FN, __VA_ARGS__)     \
    QUAD_CASE(5, FN, __VA_ARGS__) QUAD_CASE(6, FN, __VA_ARGS__) QUAD_CASE(7, FN, __VA_ARGS__) QUAD_CASE(8, FN, __VA_ARGS__)     \
    QUAD_CASE(9, FN, __VA_ARGS__) QUAD_CASE(10, FN, __VA_ARGS__) QUAD_CASE(11, FN, __VA_ARGS__) QUAD_CASE(12, FN, __VA_ARGS__)  \
    QUAD_CASE(13, FN, __VA_ARGS__) QUAD_CASE(14, FN, __VA_ARGS__)
#define QUAD_PART_FN(name) name##0
#elif QUAD_PART == 1
#define QUAD_CASES(FN, ...)                                                                                             \
    QUAD_CASE(15, FN, __VA_ARGS__) QUAD_CASE(16, FN, __VA_ARGS__) QUAD_CASE(17, FN, __VA_ARGS__) QUAD_CASE(18, FN, __VA_ARGS__) \
    QUAD_CASE(19, FN, __VA_ARGS__) QUAD_CASE(20, FN, __VA_ARGS__) QUAD_CASE(21, FN, __VA_ARGS__)
#define QUAD_PART_FN(name) name##1
#elif QUAD_PART == 2
#define QUAD_CASES(FN, ...)                                                                                             \
    QUAD_CASE(22, FN, __VA_ARGS__) QUAD_CASE(23, FN, __VA_ARGS__) QUAD_CASE(24, FN, __VA_ARGS__) QUAD_CASE(25, FN, __VA_ARGS__) \
    QUAD_CASE(26, FN, __VA_ARGS__) QUAD_CASE(27, FN, __VA_ARGS__)
#define QUAD_PART_FN(name) name##2
#else
#define QUAD_CASES(FN, ...)                                                                                             \
    QUAD_CASE(28, FN, __VA_ARGS__) QUAD_CASE(29, FN, __VA_ARGS__) QUAD_CASE(30, FN, __VA_ARGS__) QUAD_CASE(31, FN, __VA_ARGS__) \
    QUAD_CASE(32, FN, __VA_ARGS__)
#define QUAD_PART_FN(name) name##3
#endif

template <int CH>
static int quad_launch(const hals_args& a, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL((nnf_hals_quad_kernel<CH>), dim3(nblocks), dim3(64), quad_lds(CH), st, a);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
int QUAD_PART_FN(nnf_hals_quad_cap_part)(nnf_ctx* ctx, int ch) {
    switch (ch) { QUAD_CASES(quad_cap, ctx) default: return -1; }
}
int QUAD_PART_FN(nnf_hals_quad_launch_part)(int ch, const hals_args& a, int nblocks, hipStream_t st) {
    switch (ch) { QUAD_CASES(quad_launch, a, nblocks, st) default: return NNF_ERR_UNSUPPORTED; }
}

#if QUAD_PART == 0
int nnf_hals_quad_cap_part1(nnf_ctx*, int);
int nnf_hals_quad_cap_part2(nnf_ctx*, int);
int nnf_hals_quad_cap_part3(nnf_ctx*, int);
int nnf_hals_quad_launch_part1(int, const hals_args&, int, hipStream_t);
int nnf_hals_quad_launch_part2(int, const hals_args&, int, hipStream_t);
int nnf_hals_quad_launch_part3(int, const hals_args&, int, hipStream_t);
static int quad_cap_dispatch(nnf_ctx* ctx, int ch) {
    return ch <= 14 ? nnf_hals_quad_cap_part0(ctx, ch) : ch <= 21 ? nnf_hals_quad_cap_part1(ctx, ch)
         : ch <= 27 ? nnf_hals_quad_cap_part2(ctx, ch) : nnf_hals_quad_cap_part3(ctx, ch);
}

// Heuristic + residency: the quad kernel wins while its waves stay at <= 2 per SIMD (ncols <= 32768 on 256 CUs).
bool nnf_hals_quad_fits(nnf_ctx* ctx, int r, int64_t ncols, int max_blocks_cap) {
    const int ch = quad_ch(r);
    if (ch == 0) return false;
    const int64_t need = nnf_cdiv(ncols, 16);
    if (need > (int64_t)8 * ctx->num_cus || need > max_blocks_cap) return false;
    const int cap = quad_cap_dispatch(ctx, ch);
    return cap > 0 && need <= cap;
}

size_t nnf_hals_quad_gram_floats(int r) {
    const int ch = quad_ch(r), rq = 4 * ch, rs = 4 * ((ch + 3) & ~3);
    return (size_t)rq * rs + rq;
}

// Gq: workspace of nnf_hals_quad_gram_floats(r) floats.  a.Gp / a.dinv are set here.
int nnf_hals_quad_run(nnf_ctx* ctx, const float* UtU, const float* UtU2, int64_t ldg, float* Gq, unsigned* counter, hals_args a,
                      int* nblocks_out, hipStream_t st) {
    const int ch = quad_ch(a.r), rq = 4 * ch, rs = 4 * ((ch + 3) & ~3);
    if (ch == 0) return NNF_ERR_UNSUPPORTED;
    float* dinvq = Gq + (size_t)rq * rs;
    hipLaunchKernelGGL(nnf_hals_prep_quad_kernel, dim3(4 * ch), dim3(64), 0, st, UtU, UtU2, ldg, a.r, ch, Gq, dinvq, counter,
                       (a.mode == 0 && a.sweep0 == 0) ? a.status : (double*)nullptr);
    NNF_CHECK_LAUNCH();
    if (a.max_sweeps == 0) {
        // no sweep: the result is the start value (nnls.py:147 returns in_V.copy() when the loop does not run); the other
        // layouts have made that copy already, this one reads its start values inside the sweep kernel, which is not launched
        if (a.Vsrc != a.V && hipMemcpy2DAsync(a.V, (size_t)a.ldv * 4, a.Vsrc, (size_t)a.ldvs * 4, (size_t)a.ncols * 4, (size_t)a.r,
                                               hipMemcpyDeviceToDevice, st) != hipSuccess)
            return NNF_ERR_LAUNCH;
        return NNF_OK;
    }
    a.Gp = Gq;
    a.dinv = dinvq;
    const int nblocks = (int)nnf_cdiv(a.ncols, 16);
    *nblocks_out = nblocks;
    nnf_probe(ctx, NNF_PROBE_HALS, 0, st);
    const int rc = ch <= 14 ? nnf_hals_quad_launch_part0(ch, a, nblocks, st) : ch <= 21 ? nnf_hals_quad_launch_part1(ch, a, nblocks, st)
                 : ch <= 27 ? nnf_hals_quad_launch_part2(ch, a, nblocks, st) : nnf_hals_quad_launch_part3(ch, a, nblocks, st);
    nnf_probe(ctx, NNF_PROBE_HALS, 1, st);
    return rc;
}
#endif   // QUAD_PART == 0
