// Device helpers shared by the streaming MFMA kernels (k_stream.hip, k_mu.hip, k_mttkrp.hip).
#pragma once
#include "nnf_internal.h"
#include <type_traits>

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t nnf_make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <bool VEC>
__device__ __forceinline__ f32x4 nnf_bload4(rsrc_t rs, int voff, int soff) {
    if constexpr (VEC) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    } else {  // rows not 16-byte aligned: four dword loads (small / odd shapes only)
        f32x4 v;
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
        v[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4, soff, 0));
        v[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 8, soff, 0));
        v[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 12, soff, 0));
        return v;
    }
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------
// Hand-issued LDS reads.  hipcc places an LDS read of an MFMA operand right in front of its first use ("read, wait,
// multiply": the whole LDS latency exposed, once per operand group); issuing the reads a phase early and waiting on the
// counter keeps them behind MFMAs that are already queued.  Rules that make this safe (k_hals_quad.hip, nnf_cost_kernel):
//   * every read and every wait is `asm volatile` (they keep their order among themselves);
//   * a wait names the registers it releases as in/out operands, so no consumer can be scheduled above it;
//   * LDS operations return in order: extra compiler-issued ones in between only make a wait more conservative;
//   * nothing is in flight across a barrier or a loop back-edge.
// ---------------------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void nnf_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        nnf_static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ unsigned nnf_lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }
template <int OFF>
__device__ __forceinline__ void nnf_lds_read4(f32x4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF));
}
// the same read, kept behind the instructions that produce `ride` (an accumulator rides through as an in/out operand)
template <int OFF>
__device__ __forceinline__ void nnf_lds_read4_after(f32x4& d, unsigned addr, f32x4& ride) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %2 offset:%3" : "=&v"(d), "+v"(ride) : "v"(addr), "n"(OFF));
}
// wait until at most N LDS operations are outstanding; releases d[0 .. CNT)
template <int N, int CNT>
__device__ __forceinline__ void nnf_lds_wait(f32x4* d) {
    static_assert(CNT >= 1 && CNT <= 4, "");
    if constexpr (CNT == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d[0]) : "n"(N));
    else if constexpr (CNT == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(d[0]), "+v"(d[1]) : "n"(N));
    else if constexpr (CNT == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "n"(N));
}

// ---------------------------------------------------------------------------------------------------------
// A-operand staging.  LDS image of one 64-deep chunk of a row-major r x K matrix A:
//   img[(mt*4 + t)*64 + lane].c = A[16*mt + (lane&15)][k0 + 16*t + 4*(lane>>4) + c]      (zero outside r x K)
// 256 threads: thread -> (t = tid>>6, lane = tid&63), MT float4 each.
// ---------------------------------------------------------------------------------------------------------
template <int MT>
__device__ __forceinline__ void stageA_load(const float* __restrict__ A, int64_t lda, int r, int64_t K, int64_t k0,
                                            bool vec_ok, f32x4 (&regs)[MT]) {
    const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
    const int64_t kk = k0 + 16 * t + 4 * (L >> 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = 16 * mt + (L & 15);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < r && kk < K) {
            const float* p = A + (int64_t)row * lda + kk;
            if (vec_ok && kk + 3 < K) {
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
                v[0] = p[0];
                if (kk + 1 < K) v[1] = p[1];
                if (kk + 2 < K) v[2] = p[2];
                if (kk + 3 < K) v[3] = p[3];
            }
        }
        regs[mt] = v;
    }
}
template <int MT>
__device__ __forceinline__ void stageA_store(f32x4* __restrict__ img, const f32x4 (&regs)[MT]) {
    const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) img[(mt * 4 + t) * 64 + L] = regs[mt];
}

// ---- buffer-addressed staging of the two chunk images -----------------------------------------------------------------
// One buffer resource spans the whole r x K factor ([r][lda], bytes = ((r-1)*lda + K)*4: rank rows >= r fall outside and read
// as zero), the lane's share of the address is ONE VGPR and the tile / row / chunk part an SGPR offset.  The pointer form
// (stageA_load / stageK_load: a 64-bit address per staged row) had its 12-16 row bases hoisted out of the chunk loop, spilled,
// and reloaded in the middle of it behind s_waitcnt vmcnt(0) -- a full drain of the X prefetch ring once per chunk.
// Columns >= K are masked per lane (they are allocated padding of the earlier rows, possibly not finite: 0 * NaN in MFMA #2).
struct mu_stage {
    rsrc_t rs;
    int lda4;          // row pitch in bytes
    int offA, offK;    // lane parts: ((L&15)*lda + 16t + 4(L>>4))*4  and  ((L>>4)*lda + 16t + (L&15))*4,  t = wave index
    int colA, colK;    // lane's first column inside a chunk: 16t + 4(L>>4)  and  16t + (L&15)
};
__device__ __forceinline__ mu_stage mu_stage_make(const float* A, int64_t lda, int r, int64_t K) {
    mu_stage s;
    const int t = threadIdx.x >> 6, L = threadIdx.x & 63;
    s.rs = nnf_make_rsrc(A, (uint32_t)((((int64_t)r - 1) * lda + K) * 4));
    s.lda4 = (int)(lda * 4);
    s.colA = 16 * t + 4 * (L >> 4);
    s.colK = 16 * t + (L & 15);
    s.offA = (int)(((int64_t)(L & 15) * lda + s.colA) * 4);
    s.offK = (int)(((int64_t)(L >> 4) * lda + s.colK) * 4);
    return s;
}
#define MU_OOB 0x7ffffff0
template <int MT>
__device__ __forceinline__ void stageA_bload(const mu_stage& s, int64_t K, int64_t k0, bool vec_ok, f32x4 (&regs)[MT]) {
    const int64_t left = K - (k0 + s.colA);                 // columns of this lane's float4 that exist (<= 0: none)
    const int k04 = (int)(k0 * 4);
    if (vec_ok) {
        const int vo = left > 0 ? s.offA : MU_OOB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rs, vo, 16 * mt * s.lda4 + k04, 0));
            if (left < 4) {
                v[1] = left > 1 ? v[1] : 0.f;
                v[2] = left > 2 ? v[2] : 0.f;
                v[3] = left > 3 ? v[3] : 0.f;
            }
            regs[mt] = v;
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.rs, left > c ? s.offA + 4 * c : MU_OOB,
                                                                                      16 * mt * s.lda4 + k04, 0));
            regs[mt] = v;
        }
    }
}
template <int MT>
__device__ __forceinline__ void stageK_bload(const mu_stage& s, int64_t K, int64_t k0, f32x4 (&regs)[MT]) {
    const int vo = (k0 + s.colK < K) ? s.offK : MU_OOB;
    const int k04 = (int)(k0 * 4);
#pragma unroll
    for (int s4 = 0; s4 < MT; ++s4) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.rs, vo, (16 * s4 + 4 * c) * s.lda4 + k04, 0));
        regs[s4] = v;
    }
}



static inline bool x_vec_ok(const float* X, int64_t ldx) { return (((uintptr_t)X) & 15) == 0 && (ldx & 3) == 0; }

// ---------------------------------------------------------------------------------------------------------
// Element-wise terms of the cost functions (beta_divergence.py:45-52), written so that fp32 does not cancel:
//   h(t) = t - log1p(t)   (series below 1/4)
//   KL  : x log(x/p) - x + p = x h((p-x)/x)            (x = 0 -> p)
//   IS  : x/p - log(x/p) - 1 = h((x-p)/p)
//   gen : (x^b + (b-1) p^b - b x p^(b-1)) / (b(b-1)) = p^b phi((x-p)/p),
//         phi(u) = ((1+u)^b - 1 - b u)/(b(b-1))        (binomial series below 3/10)
// Inputs must be strictly positive for the divergences, as in the reference.
// ---------------------------------------------------------------------------------------------------------
enum { NNF_COST_FROB = 0, NNF_COST_KL = 1, NNF_COST_IS = 2, NNF_COST_GEN = 3,
       NNF_RATIO_KL = 4, NNF_RATIO_GEN = 5,     // 4, 5: write X.*P^(beta-2) [and P^(beta-1)] instead of summing a cost
       NNF_PROD = 6 };                          // 6: write the model P itself (a rank chunk's share of it, ranks above 128)

// hardware transcendental units (v_log_f32 = log2, v_exp_f32 = exp2, v_rcp_f32): ~1 ulp, a few issue slots each
__device__ __forceinline__ float nnf_ln(float x) { return 0.69314718056f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float nnf_pow(float x, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x)); }

// h(t) = t - log1p(t) with q = 1 + t handed over separately: callers form q as a plain ratio (>= 0 by construction),
// because 1 + (a-b)/b can round to a tiny NEGATIVE number when a << b (data entries of 1e-12 next to a model of order 1:
// the clamped zeros of an NNDSVD start, multilayer NMF's inner layers) and the logarithm then returns NaN.
__device__ __forceinline__ float nnf_h(float t, float q) {
    // small |t|: alternating series (no cancellation); otherwise t - ln(q) directly (no cancellation left to fear)
    float s = 1.f / 14.f;
#pragma unroll
    for (int k = 11; k >= 0; --k) s = fmaf(s, -t, 1.f / (float)(k + 2));
    const float series = t * t * s;
    const float direct = t - nnf_ln(q);
    return fabsf(t) < 0.25f ? series : direct;
}

// KL term written on the reciprocal of the MODEL entry: x ln(x/p) - x + p = p g(rho), rho = x/p (>= 0 by construction),
//   g(rho) = rho ln(rho) - rho + 1 = (1 + s) ln(1 + s) - s,  s = (x - p)/p
//   |s| < 1/8: s^2 sum_k (-s)^k / ((k+1)(k+2))  (coefficients fall like 1/k^2: 6 terms for fp32, no cancellation);
//   otherwise rho ln(rho) - s directly (rho = 0, a zero data entry: 0 * ln(tiny) - (-1) = 1, the term is p as in the reference).
// 1/p is the reciprocal the fused KL update needs anyway (R = x/p): one transcendental and three FMAs per entry less than
// the form on 1/x (x h((p-x)/x)) -- the divergence rides on the same fp32 pipe as the MFMAs of nnf_mu_left_kl_cost_f32.
__device__ __forceinline__ float nnf_kl_term(float x, float p) {
    const float rp = __builtin_amdgcn_rcpf(p);
    const float rho = x * rp, s = (x - p) * rp;
    // |s| < 1/8: six terms reach fp32 (term k = s^k 2/((k+1)(k+2)) relative to the first: 1.4e-7 at k = 6) -- the divergence
    // rides on the same fp32 pipe as the MFMAs of nnf_mu_left_kl_cost_f32, every instruction per entry is 4.5 us at config C;
    // beyond it rho ln(rho) - s has lost at most four of its 24 bits to cancellation (|s| >= 1/8: the two terms are >= 0.1,
    // their difference >= s^2/2 (1 - |s|/3) >= 7e-3).
    float a = 1.f / 42.f;                         // k = 5
    a = fmaf(a, s, -1.f / 30.f);
    a = fmaf(a, s, 1.f / 20.f);
    a = fmaf(a, s, -1.f / 12.f);
    a = fmaf(a, s, 1.f / 6.f);
    a = fmaf(a, -s, 0.5f);
    const float series = s * s * a;
    const float direct = fmaf(rho * 0.69314718056f, __builtin_amdgcn_logf(fmaxf(rho, 1e-37f)), -s);
    return p * (fabsf(s) < 0.125f ? series : direct);
}

template <int OP>
__device__ __forceinline__ float nnf_cost_term(float x, float p, float beta) {
    if constexpr (OP == NNF_COST_FROB) {
        const float d = x - p;
        return d * d;
    } else if constexpr (OP == NNF_COST_KL) {
        return nnf_kl_term(x, p);
    } else if constexpr (OP == NNF_COST_IS) {
        const float rp = __builtin_amdgcn_rcpf(p);
        return nnf_h((x - p) * rp, x * rp);
    } else {
        const float rp = __builtin_amdgcn_rcpf(p);
        const float u = (x - p) * rp;
        float s = 1.f;
#pragma unroll
        for (int k = 16; k >= 3; --k) s = fmaf(s, (beta - (float)(k - 1)) * u * (1.f / (float)k), 1.f);
        const float series = 0.5f * u * u * s;
        // (1+u)^beta from the ratio x/p itself (>= 0), not from 1 + u (see nnf_h); beta > 0 here, so 0^beta = 0
        const float direct = (nnf_pow(x * rp, beta) - 1.f - beta * u) / (beta * (beta - 1.f));
        const float phi = fabsf(u) < 0.3f ? series : direct;
        return nnf_pow(p, beta) * phi;
    }
}
