// Fast HALS sweep kernels: one lane = one column of V, column resident in VGPRs, Gram through the scalar cache.
// Compiled once per -DHALS_PART=0..3 (each part instantiates a subset of padded ranks) so the parts build in parallel.
// See k_hals.hip for the algorithm, the grid-exchange protocol and the entry points.
#include "k_hals_common.h"

#ifndef HALS_PART
#error "compile with -DHALS_PART=0..3"
#endif

typedef const __attribute__((address_space(4))) f32x2* cg2_t;   // constant address space -> s_load
typedef const __attribute__((address_space(4))) float* cf_t;

// One Gauss-Seidel sweep (nnls.py:158-170) over the column held in v2 = {(v[0],v[1]), (v[2],v[3]), ...}.
//   x     = (UtM[k] - UtU[k,:].v - sp) / UtU[k,k]
//   v[k] <- max(v[k] + x, 0)            (== v[k] + max(x, -v[k]) of the reference, same rounding)
//   step  = x if not clipped else -v[k]
// The per-row asm ties do two things hipcc would otherwise undo: (1) the scalar loads of Gram row k cannot be issued
// before row k-1 has finished (unconstrained, all RP^2 loads are clustered up front and thousands of SGPRs spill);
// (2) nothing about row k is precomputed rows ahead (keeps the live set at the column itself).
template <int R, bool KEEPB>
__device__ __forceinline__ float hals_sweep_column(f32x2 (&v2)[R / 2], const float (&b)[KEEPB ? R : 1], rsrc_t rb, int voff,
                                                   int ldm4, const float* __restrict__ Gp, const float* __restrict__ dinv,
                                                   float sp) {
    float nd = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        uint64_t pz = (uint64_t)Gp, dz = (uint64_t)dinv;
        if (k > 0)
            asm volatile("" : "+s"(pz), "+s"(dz), "+v"(v2[(k - 1) / 2]));
        else
            asm volatile("" : "+s"(pz), "+s"(dz));
        cg2_t G2 = (cg2_t)(pz) + (k * R) / 2;
        const float di = ((cf_t)dz)[k];
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j + 1 < R / 2; j += 2) {
            a0 = __builtin_elementwise_fma(G2[j], v2[j], a0);
            a1 = __builtin_elementwise_fma(G2[j + 1], v2[j + 1], a1);
        }
        if constexpr ((R / 2) & 1) a0 = __builtin_elementwise_fma(G2[R / 2 - 1], v2[R / 2 - 1], a0);
        const f32x2 a = a0 + a1;
        const float dot = a[0] + a[1];
        float bk;
        if constexpr (KEEPB)
            bk = b[k];
        else  // rows >= r lie outside the descriptor: the load returns 0
            bk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, k * ldm4, 0));
        const float vk = v2[k / 2][k & 1];
        const float x = (bk - dot - sp) * di;
        const float t = vk + x;
        const bool keep = t > 0.f;
        float vn = keep ? t : 0.f;
        float step = keep ? x : -vk;
        if (di == 0.f) {  // zero Gram diagonal (or padded row): row skipped (nnls.py:160)
            vn = vk;
            step = 0.f;
        }
        v2[k / 2][k & 1] = vn;
        nd = fmaf(step, step, nd);
    }
    return nd;
}

template <int RP, bool RES>
__global__ __launch_bounds__(256, 2) void nnf_hals_kernel(hals_args a) {
    constexpr bool KEEPB = (RP <= 64);
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
            if constexpr (KEEPB)
                b[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff, k * ldm4, 0));
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

    double eps0 = 0.0, eps = 1.0;
    int done = 0;
    bool ok = true;
    for (int s = 1; s <= a.max_sweeps; ++s) {
        double nd = 0.0;
        if constexpr (RES) {
            const float f = hals_sweep_column<RP, KEEPB>(v2, b, rb, voff0, ldm4, a.Gp, a.dinv, a.sp);
            nd = gtid < a.ncols ? (double)f : 0.0;
        } else {
            for (int64_t col = gtid; col < a.ncols; col += gthreads) {
                const int voff = (int)(col * 4);
                load_col(voff);
                nd += (double)hals_sweep_column<RP, KEEPB>(v2, b, rb, voff, ldm4, a.Gp, a.dinv, a.sp);
                store_col(voff);
            }
        }
        done = s;
        const double bs = nnf_block_sum_f64(nd, red);
        if (a.mode == 1) {
            if (threadIdx.x == 0) a.sweep_partials[(size_t)(s - 1) * nblocks + blockIdx.x] = bs;
        } else {
            double mine[1] = {bs}, tot[1];
            ok = grid_exchange<1>(a.sy, (unsigned)s, nblocks, mine, tot, red, &lds_flag);
            if (!ok) break;
            if (s == 1) eps0 = tot[0];
            eps = tot[0];
            if (!(eps >= a.delta * eps0)) break;  // nnls.py:156
        }
    }
    if constexpr (RES) store_col(voff0);
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
