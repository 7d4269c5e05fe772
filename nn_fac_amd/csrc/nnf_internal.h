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
    unsigned hals_epoch;   // salt of the HALS exchange tags (k_hals_common.h)
    char* xch;             // HALS exchange granules: a region of its own that no other kernel ever carves from, so a word
    size_t xch_bytes;      // left there can only be an older HALS tag (never look-alike float data); zeroed at creation
                           // and whenever the epoch wraps
    hipEvent_t probe[2];   // optional caller-owned events recorded around ONE main kernel (nnf_ctx_set_probe[_kernel])
    int probe_id;          // which kernel the probe brackets (NNF_PROBE_*, include/nnfac_hip.h); default: W^T X
    unsigned* gc_ticket;   // nnf_nmf_gram_cost_f32: finishing ticket (256 zeroed bytes of its own; the kernel returns it to zero)
    hipEvent_t* ring;      // nnf_ctx_set_probe_ring: ring_n (begin, end) pairs, the next launch records pair ring_pos
    int ring_n, ring_pos;
    char* big;             // nnf_ctx_set_scratch: caller-owned device buffer for the m x n model of a rank above 128 (may be NULL)
    size_t big_bytes;
};

// Build-switch registry (nnf_build_flags): every translation unit that has timing-only ablation / A-B macros records the
// values it was compiled with at load time.
void nnf_register_build_flags(const char* unit, const char* flags);
#define NNF_STR2(x) #x
#define NNF_STR(x) NNF_STR2(x)
#define NNF_CAT2(a, b) a##b
#define NNF_CAT(a, b) NNF_CAT2(a, b)
#define NNF_BUILD_FLAGS_I(unit, str)                                                      \
    namespace {                                                                           \
    struct nnf_bf_##unit { nnf_bf_##unit() { nnf_register_build_flags(#unit, str); } };  \
    static nnf_bf_##unit nnf_bf_inst_##unit;                                              \
    }
#define NNF_BUILD_FLAGS(unit, str) NNF_BUILD_FLAGS_I(unit, str)   /* (arguments are macro-expanded first: NNF_CAT(base, PART)) */

#define NNF_HALS_MAX_SWEEPS 1000   // per launch: the exchange tag holds the sweep index in 10 bits (k_hals_common.h)
#define NNF_HALS_MAX_BLOCKS 2048   // workgroups of one persistent solve (3 * 256 CUs fits)

// measurement hook: record the caller's event `which` (0 begin, 1 end) if the probe is armed for kernel `id`
static inline void nnf_probe(nnf_ctx* c, int id, int which, hipStream_t st) {
    if (c->probe_id != id) return;
    if (c->ring) {
        if (c->ring_pos < c->ring_n) {
            (void)hipEventRecord(c->ring[2 * c->ring_pos + which], st);
            if (which == 1) ++c->ring_pos;
        }
        return;
    }
    if (c->probe[which]) (void)hipEventRecord(c->probe[which], st);
}

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

// v + (v moved by the DPP control CTRL); lanes of rows outside ROW_MASK add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double nnf_dpp_add_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
// Sum over the 64 lanes in a fixed order, all on the VALU (DPP; no LDS crossbar round trips): xor-1, xor-2, half-row and
// row mirrors give every lane its 16-lane row sum, row_bcast15/31 chain the four rows into lane 63, which is broadcast.
// The result is wave-uniform (valid in every lane).
__device__ __forceinline__ double nnf_wave_sum_f64(double v) {
    v = nnf_dpp_add_f64<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = nnf_dpp_add_f64<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = nnf_dpp_add_f64<0x141, 0xf>(v);   // row_half_mirror
    v = nnf_dpp_add_f64<0x140, 0xf>(v);   // row_mirror
    v = nnf_dpp_add_f64<0x142, 0xa>(v);   // row_bcast15 into rows 1, 3
    v = nnf_dpp_add_f64<0x143, 0xc>(v);   // row_bcast31 into rows 2, 3
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
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
                            float* out, int64_t ldo, hipStream_t st, double* out64 = nullptr);   // out64: the sums before rounding (rows x cols, contiguous)

// out[z] = A (p x q) B[z] (q x cols), z < batch: a rank-sized left operand staged in LDS (k_mu.hip)
int nnf_small_gemm_launch(const float* A, int64_t lda, int p, int q, const float* B, int64_t ldb, int64_t cols, float* out,
                          int64_t ldo, int64_t batch, int64_t bstride, int64_t ostride, hipStream_t st);

// out[0] = scale * sum of `count` doubles, index order, one workgroup
int nnf_launch_sum_f64(const double* partial, int64_t count, double scale, double* out, hipStream_t st);

int nnf_xty_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                 int r, int64_t ldu, float* out, int64_t ldo, hipStream_t st);
int nnf_xht_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V,
                 int r, int64_t ldv, float* out, int64_t ldo, hipStream_t st);
int nnf_gram_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                  hipStream_t st, double* G64 = nullptr);
