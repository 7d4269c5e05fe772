// Internal declarations shared by the kernel translation units of libnnfac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/nnfac_hip.h"

struct nnf_ctx {
    int device;
    int num_cus;
    size_t ws_bytes;
    char* ws;          // device scratch (split-K slabs, partial sums, barrier words); zeroed once at creation
    unsigned hals_epoch;  // salt of the HALS exchange tags (k_hals_common.h)
};

#define NNF_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return NNF_ERR_LAUNCH;        \
    } while (0)

static inline int64_t nnf_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t nnf_rup(int64_t a, int64_t b) { return nnf_cdiv(a, b) * b; }

// workspace carve helper: 256-byte aligned bump allocator over ctx->ws
struct nnf_ws_cursor {
    char* base;
    size_t cap, off;
    __host__ nnf_ws_cursor(nnf_ctx* c) : base(c->ws), cap(c->ws_bytes), off(0) {}
    __host__ void* take(size_t bytes) {
        size_t a = (off + 255) & ~size_t(255);
        if (a + bytes > cap) return nullptr;
        off = a + bytes;
        return base + a;
    }
    __host__ size_t remaining() const {
        const size_t a = (off + 255) & ~size_t(255);
        return a < cap ? cap - a : 0;
    }
};

// ---- device helpers -------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double nnf_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// Block-wide fp64 sum in a fixed order (wave shuffle tree, then waves in index order).  Result in thread 0.
// `red` must hold (blockDim.x/64) doubles.  All threads must call.
__device__ __forceinline__ double nnf_block_sum_f64(double v, double* red) {
    v = nnf_wave_sum_f64(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) s += red[i];
    __syncthreads();
    return s;
}

// XCD-aware remap: workgroups that share `group` run with equal blockIdx%8 (observed round-robin placement;
// speed only, never correctness).  bid -> (group, member) with `members` members per group.
__device__ __forceinline__ void nnf_xcd_map(int bid, int members, int& group, int& member) {
    const int x = bid & 7, q = bid >> 3;
    member = q % members;
    group = x + 8 * (q / members);
}

// kernels / launchers implemented in the .hip files
int nnf_launch_reduce_slabs(const float* slabs, int nslab, int64_t slab_stride, int rows, int64_t cols, int64_t lds,
                            float* out, int64_t ldo, hipStream_t st);

int nnf_xty_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                 int r, int64_t ldu, float* out, int64_t ldo, hipStream_t st);
int nnf_xht_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V,
                 int r, int64_t ldv, float* out, int64_t ldo, hipStream_t st);
int nnf_gram_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                  hipStream_t st);
