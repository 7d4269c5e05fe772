// Context management and small deterministic helpers of libnnfac_hip.so.
#include "nnf_internal.h"
#include <new>

#include <map>
#include <string>
#include <string.h>

extern "C" int nnf_version(void) { return 101; }  // 0.1.1

static std::map<std::string, std::string>& build_flag_registry() {
    static std::map<std::string, std::string> r;   // constructed on first use: safe from any unit's static initialiser
    return r;
}
void nnf_register_build_flags(const char* unit, const char* flags) { build_flag_registry()[unit] = flags; }

extern "C" size_t nnf_build_flags(char* buf, size_t cap) {
    std::string all;
    for (const auto& kv : build_flag_registry()) {
        if (!all.empty()) all += "; ";
        all += kv.first + ": " + kv.second;
    }
    if (buf && cap > 0) {
        const size_t n = all.size() < cap - 1 ? all.size() : cap - 1;
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return all.size();
}

extern "C" const char* nnf_status_string(int status) {
    switch (status) {
        case NNF_OK: return "ok";
        case NNF_ERR_ARG: return "invalid argument";
        case NNF_ERR_LAUNCH: return "HIP launch/runtime error";
        case NNF_ERR_UNSUPPORTED: return "shape not supported by the built kernels";
        case NNF_ERR_WORKSPACE: return "context workspace too small";
        case NNF_ERR_DEVICE: return "device unavailable";
        default: return "unknown status";
    }
}

extern "C" int nnf_ctx_create(nnf_ctx** out_ctx, int device, size_t workspace_bytes) {
    if (!out_ctx) return NNF_ERR_ARG;
    *out_ctx = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return NNF_ERR_DEVICE;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return NNF_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return NNF_ERR_DEVICE;
    nnf_ctx* c = new (std::nothrow) nnf_ctx();
    if (!c) return NNF_ERR_DEVICE;
    c->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    c->num_cus = cus;
    c->ws_bytes = workspace_bytes ? workspace_bytes : (size_t)256 << 20;
    c->ws = nullptr;
    c->hals_epoch = 0u;
    c->probe[0] = c->probe[1] = nullptr;
    c->probe_id = NNF_PROBE_XTY;
    c->ring = nullptr;
    c->ring_n = c->ring_pos = 0;
    c->gc_ticket = nullptr;
    c->big = nullptr;
    c->big_bytes = 0;
    c->xch = nullptr;
    c->xch_bytes = (size_t)(NNF_HALS_MAX_SWEEPS + 2) * NNF_HALS_MAX_BLOCKS * 16;
    hipError_t e = hipMalloc((void**)&c->ws, c->ws_bytes);
    if (e == hipSuccess) e = hipMemset(c->ws, 0, c->ws_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->xch, c->xch_bytes);
    if (e == hipSuccess) e = hipMemset(c->xch, 0, c->xch_bytes);   // exchange words must not start as look-alike tags
    if (e == hipSuccess) e = hipMalloc((void**)&c->gc_ticket, 256);
    if (e == hipSuccess) e = hipMemset(c->gc_ticket, 0, 256);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (c->ws) (void)hipFree(c->ws);
        if (c->xch) (void)hipFree(c->xch);
        if (c->gc_ticket) (void)hipFree(c->gc_ticket);
        delete c;
        return NNF_ERR_WORKSPACE;
    }
    *out_ctx = c;
    return NNF_OK;
}

// Caller-owned device scratch for calls whose temporaries grow with the DATA (the m x n model of nnf_frob_resid_f32 /
// nnf_betadiv_f32 at a rank above NNF_MAX_RANK: 4*m*ldp bytes, ldp = n rounded up to 4).  Not freed by the library; the
// caller keeps it alive until the calls that use it have finished on their streams.  (NULL, 0) withdraws it.
extern "C" int nnf_ctx_set_scratch(nnf_ctx* ctx, void* device_buf, size_t bytes) {
    if (!ctx || (device_buf == nullptr) != (bytes == 0) || (((uintptr_t)device_buf) & 15) != 0) return NNF_ERR_ARG;
    ctx->big = (char*)device_buf;
    ctx->big_bytes = bytes;
    return NNF_OK;
}

extern "C" int nnf_ctx_destroy(nnf_ctx* ctx) {
    if (!ctx) return NNF_ERR_ARG;
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->xch) (void)hipFree(ctx->xch);
    if (ctx->gc_ticket) (void)hipFree(ctx->gc_ticket);
    delete[] ctx->ring;
    delete ctx;
    return NNF_OK;
}

extern "C" size_t nnf_ctx_workspace_bytes(const nnf_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

extern "C" int nnf_ctx_set_probe(nnf_ctx* ctx, void* ev_begin, void* ev_end) {
    if (!ctx || ((ev_begin == nullptr) != (ev_end == nullptr))) return NNF_ERR_ARG;
    ctx->probe[0] = (hipEvent_t)ev_begin;
    ctx->probe[1] = (hipEvent_t)ev_end;
    return NNF_OK;
}

extern "C" int nnf_ctx_set_probe_ring(nnf_ctx* ctx, void* const* events, int npairs) {
    if (!ctx || npairs < 0 || (npairs > 0 && !events)) return NNF_ERR_ARG;
    delete[] ctx->ring;
    ctx->ring = nullptr;
    ctx->ring_n = ctx->ring_pos = 0;
    if (npairs == 0) return NNF_OK;
    for (int i = 0; i < 2 * npairs; ++i)
        if (!events[i]) return NNF_ERR_ARG;
    ctx->ring = new (std::nothrow) hipEvent_t[2 * (size_t)npairs];
    if (!ctx->ring) return NNF_ERR_DEVICE;
    for (int i = 0; i < 2 * npairs; ++i) ctx->ring[i] = (hipEvent_t)events[i];
    ctx->ring_n = npairs;
    return NNF_OK;
}

extern "C" int nnf_ctx_probe_ring_count(const nnf_ctx* ctx) { return (ctx && ctx->ring) ? ctx->ring_pos : 0; }

extern "C" int nnf_ctx_set_probe_kernel(nnf_ctx* ctx, int kernel_id) {
    if (!ctx || kernel_id < 0 || kernel_id >= NNF_PROBE_COUNT) return NNF_ERR_ARG;
    ctx->probe_id = kernel_id;
    return NNF_OK;
}

// ---- dot: sum_ij A[i,j]*B[i,j] in fp64, fixed order ------------------------------------------------------
__global__ __launch_bounds__(256) void nnf_dot_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                      int64_t ldb, int64_t rows, int64_t cols, double* __restrict__ partial) {
    __shared__ double red[4];
    const int64_t total = rows * cols;
    double s = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / cols, j = e - i * cols;
        s += (double)A[i * lda + j] * (double)B[i * ldb + j];
    }
    const double t = nnf_block_sum_f64(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void nnf_sum_f64_kernel(const double* __restrict__ partial, int64_t count,
                                                          double* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int64_t e = threadIdx.x; e < count; e += 8 * 256) {   // eight loads in flight, added in index order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[e + 256 * u < count ? e + 256 * u : count - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (e + 256 * u < count) ? v[u] : 0.0;
    }
    const double t = nnf_block_sum_f64(s, red);
    if (threadIdx.x == 0) out[0] = t;
}

extern "C" int nnf_dot_f32(nnf_ctx* ctx, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t rows,
                           int64_t cols, double* out_f64, void* stream) {
    if (!ctx || !A || !B || !out_f64 || rows < 1 || cols < 1 || lda < cols || ldb < cols) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t grid = nnf_cdiv(rows * cols, 256 * 8);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    nnf_ws_cursor cur(ctx);
    double* partial = (double*)cur.take((size_t)grid * 8);
    if (!partial) return NNF_ERR_WORKSPACE;
    hipLaunchKernelGGL(nnf_dot_kernel, dim3((int)grid), dim3(256), 0, st, A, lda, B, ldb, rows, cols, partial);
    NNF_CHECK_LAUNCH();
    hipLaunchKernelGGL(nnf_sum_f64_kernel, dim3(1), dim3(256), 0, st, partial, grid, out_f64);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

__global__ void nnf_hadamard_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                    int64_t count) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x)
        C[e] = A[e] * B[e];
}
extern "C" int nnf_hadamard_f32(nnf_ctx* ctx, const float* A, const float* B, float* C, int64_t count, void* stream) {
    if (!ctx || !A || !B || !C || count < 1) return NNF_ERR_ARG;
    int64_t grid = nnf_cdiv(count, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(nnf_hadamard_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, A, B, C, count);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

// ---- Frobenius cost of an NMF iterate through the Gram identity (include/nnfac_hip.h) ------------------------------------
// 16 threads per column j: thread t takes the rows a = t, t + 16, ...:  t_a = sum_b UtU[a][b] v_b, then its shares of
//   A = sum v_a UtM[a][j],  A2 = sum (v_a UtM[a][j])^2,  B = sum v_a t_a,  V2 = sum v_a^2      (all fp64)
// block sums in a fixed order -> partial[wg][4]; the LAST workgroup to finish (a ticket) adds the partials in index order
// -- the same bits whichever workgroup that is -- and writes {cost, flag, estimate}.  One launch.
template <typename GT>
__global__ __launch_bounds__(256) void nnf_gram_cost_kernel(const float* __restrict__ V, int64_t ldv, const float* __restrict__ UtM,
                                                            int64_t ldm, const float* __restrict__ G, const float* __restrict__ G2,
                                                            const double* __restrict__ G64, int64_t ldg, int r, int64_t n,
                                                            double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                            const double* __restrict__ normx2, double* __restrict__ out,
                                                            double sigma_a, double bias_a, int g_global, double sigma_g) {
    extern __shared__ __attribute__((aligned(16))) float gc_sh[];
    // the Gram: staged in LDS (r x r; GT = double when the caller has the sums before their rounding to fp32, G64), or -- ranks
    // above NNF_MAX_RANK, where it no longer fits -- read where it lies (L2)
    GT* gsh = reinterpret_cast<GT*>(gc_sh);
    const GT* gsrc = reinterpret_cast<const GT*>(std::is_same<GT, double>::value ? (const void*)G64 : (const void*)G);
    const GT* g = g_global ? gsrc : gsh;
    const int64_t gs = g_global ? (std::is_same<GT, double>::value ? (int64_t)r : ldg) : r;
    float* vc = gc_sh + (g_global ? (size_t)0 : (size_t)r * r * (sizeof(GT) / 4));   // 16 columns x r
    __shared__ double red[4];
    __shared__ unsigned last;
    if (!g_global)
        for (int e = threadIdx.x; e < r * r; e += 256) {      // G2: the Gram is a Hadamard product (NTF: ntf.py:442-445)
            const int64_t o = (int64_t)(e / r) * ldg + (e % r);
            if constexpr (std::is_same<GT, double>::value) gsh[e] = G64[e];
            else gsh[e] = G2 ? G[o] * G2[o] : G[o];
        }
    const int tc = threadIdx.x >> 4, t = threadIdx.x & 15;
    const int64_t j = (int64_t)blockIdx.x * 16 + tc;
    for (int a = t; a < r; a += 16) vc[tc * r + a] = (j < n) ? V[(int64_t)a * ldv + j] : 0.f;
    __syncthreads();
    double pA = 0.0, pA2 = 0.0, pB = 0.0, pV2 = 0.0;
    if (j < n) {
        const float* vj = vc + tc * r;
        for (int a = t; a < r; a += 16) {
            const GT* ga = g + (size_t)a * gs;
            double ta = 0.0;
            if (g_global && G2 != nullptr) {       // Hadamard Gram read in place (NTF at a rank above 128)
                const float* gb = G2 + (size_t)a * ldg;
                for (int b = 0; b < r; ++b) ta = __builtin_fma((double)((float)ga[b] * gb[b]), (double)vj[b], ta);
            } else
            for (int b = 0; b < r; ++b) ta = __builtin_fma((double)ga[b], (double)vj[b], ta);
            const double va = (double)vj[a], p = va * (double)UtM[(int64_t)a * ldm + j];
            pA += p;
            pA2 = __builtin_fma(p, p, pA2);
            pB = __builtin_fma(va, ta, pB);
            pV2 = __builtin_fma(va, va, pV2);
        }
    }
    const double sA = nnf_block_sum_f64(pA, red), sA2 = nnf_block_sum_f64(pA2, red), sB = nnf_block_sum_f64(pB, red),
                 sV2 = nnf_block_sum_f64(pV2, red);
    if (threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(partial) + (size_t)blockIdx.x * 4;
        const double vals[4] = {sA, sA2, sB, sV2};
        for (int i = 0; i < 4; ++i)
            __hip_atomic_store(o + i, __builtin_bit_cast(unsigned long long, vals[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // release: the partials are visible to whoever sees this workgroup's ticket
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (tk == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (last == 0u) return;
    // ---- the last workgroup: everything is published ----
    const int nwg = (int)gridDim.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const unsigned long long* pp = reinterpret_cast<const unsigned long long*>(partial);
    for (int e = threadIdx.x; e < nwg; e += 256)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s[i] += __builtin_bit_cast(double, __hip_atomic_load(pp + (size_t)e * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    double tot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tot[i] = nnf_block_sum_f64(s[i], red);
    float gm = 0.f;
    for (int e = threadIdx.x; e < r * r; e += 256) {
        float ge = (float)g[(size_t)(e / r) * gs + (e % r)];
        if (g_global && G2 != nullptr) ge *= G2[(size_t)(e / r) * ldg + (e % r)];
        gm = fmaxf(gm, fabsf(ge));
    }
    gm = fmaxf(gm, __shfl_xor(gm, 1, 64));
    gm = fmaxf(gm, __shfl_xor(gm, 2, 64));
    gm = fmaxf(gm, __shfl_xor(gm, 4, 64));
    gm = fmaxf(gm, __shfl_xor(gm, 8, 64));
    gm = fmaxf(gm, __shfl_xor(gm, 16, 64));
    gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
    __shared__ float gmax[4];
    if ((threadIdx.x & 63) == 0) gmax[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) {
        gm = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
        const double cost = normx2[0] - 2.0 * tot[0] + tot[2];
        // fp32 storage of UtM (relative rms 2^-24/sqrt(3) = 3.4e-8, measured 5.5e-8 with the accumulation at config B) and of UtU:
        //   sigma_A = 2 * sigma_a * || V .* UtM ||_F,   sigma_B <= 4e-8 * max|UtU| * ||V||_F^2  (= trace of V V^T >= ||V V^T||_F)
        // sigma_a is the caller's figure for the relative rms rounding of a UtM entry -- it grows with the rows one workgroup of
        // the cross-product kernel sums in fp32 (tools/probes/accum_error_probe.py: 5.7e-8 at 100000 x 2000 rank 50, 9.5e-7 at
        // 1e6 x 4000 rank 100, there with a MEAN of -2.2e-7) -- and bias_a its figure for the relative mean: a bias does not
        // average down over the entries, it enters with the whole inner product.
        // sigma_g: relative rms rounding of a Gram entry as handed over -- 4e-8 for fp32 storage, the caller's figure for G64
        const double sa = 2.0 * sigma_a * sqrt(tot[1]), sb = sigma_g * (double)gm * tot[3];
        const double est = 4.0 * sqrt(sa * sa + sb * sb) + 2.0 * 2.0 * bias_a * fabs(tot[0]);
        out[0] = cost;
        out[1] = (est <= 5e-4 * cost) ? 0.0 : 1.0;      // (a NaN or a non-positive cost lands on 1)
        out[2] = est;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call (stream-ordered)
    }
}
static int gram_cost_launch(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU,
                            const float* UtU_b, const double* UtU64, int64_t ldg, int r, int64_t n, const double* normx2_f64,
                            double sigma_a, double bias_a, double sigma_g, double* out_f64, void* stream) {
    if (!ctx || !V || !UtM || !UtU || !normx2_f64 || !out_f64 || r < 1 || n < 1 || ldv < n || ldm < n || ldg < r) return NNF_ERR_ARG;
    if (!(sigma_a >= 0.0) || !(bias_a >= 0.0) || !(sigma_g >= 0.0)) return NNF_ERR_ARG;
    if (UtU64 != nullptr && UtU_b != nullptr) return NNF_ERR_ARG;
    const int g_global = r > NNF_MAX_RANK ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nwg = nnf_cdiv(n, 16);
    if (nwg > (int64_t)1 << 24) return NNF_ERR_UNSUPPORTED;
    nnf_ws_cursor cur(ctx);
    double* partial = (double*)cur.take((size_t)nwg * 4 * 8);
    if (!partial) return NNF_ERR_WORKSPACE;
    // the ticket: 256 bytes of the context's own, zero at creation, returned to zero by the kernel itself (stream-ordered)
    unsigned* ticket = ctx->gc_ticket;
    if (!ticket) return NNF_ERR_WORKSPACE;
    const size_t shm = ((g_global ? (size_t)0 : (size_t)r * r * (UtU64 ? 2 : 1)) + (size_t)16 * r) * 4;
    if (shm > 150 * 1024) return NNF_ERR_UNSUPPORTED;
    if (UtU64) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_gram_cost_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      150 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(nnf_gram_cost_kernel<double>, dim3((int)nwg), dim3(256), shm, st, V, ldv, UtM, ldm, UtU, UtU_b, UtU64, ldg, r, n,
                           partial, ticket, normx2_f64, out_f64, sigma_a, bias_a, g_global, sigma_g);
    } else {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_gram_cost_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      96 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(nnf_gram_cost_kernel<float>, dim3((int)nwg), dim3(256), shm, st, V, ldv, UtM, ldm, UtU, UtU_b, UtU64, ldg, r, n,
                           partial, ticket, normx2_f64, out_f64, sigma_a, bias_a, g_global, sigma_g);
    }
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
extern "C" int nnf_nmf_gram_cost_f32(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU,
                                     const float* UtU_b, int64_t ldg, int r, int64_t n, const double* normx2_f64, double* out_f64,
                                     void* stream) {
    return gram_cost_launch(ctx, V, ldv, UtM, ldm, UtU, UtU_b, nullptr, ldg, r, n, normx2_f64, 6e-8, 0.0, 4e-8, out_f64, stream);
}
extern "C" int nnf_nmf_gram_cost_cal_f32(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU,
                                         const float* UtU_b, int64_t ldg, int r, int64_t n, const double* normx2_f64, double sigma_a,
                                         double bias_a, double* out_f64, void* stream) {
    return gram_cost_launch(ctx, V, ldv, UtM, ldm, UtU, UtU_b, nullptr, ldg, r, n, normx2_f64, sigma_a, bias_a, 4e-8, out_f64, stream);
}
// The quadratic form on the Gram BEFORE its rounding to fp32 (UtU64: r x r doubles from nnf_gram_f64_f32; UtU, the fp32 Gram the
// solve used, still gives max|UtU| of the estimate): sigma_g = the caller's figure for the relative rms error of a UtU64 entry
// (what the fp32 accumulation inside a split leaves: Engine.cross_rounding measures it), in place of the 4e-8 of fp32 storage.
extern "C" int nnf_nmf_gram_cost_g64_f32(nnf_ctx* ctx, const float* V, int64_t ldv, const float* UtM, int64_t ldm, const float* UtU,
                                         const double* UtU64, int64_t ldg, int r, int64_t n, const double* normx2_f64, double sigma_a,
                                         double bias_a, double sigma_g, double* out_f64, void* stream) {
    if (!UtU64) return NNF_ERR_ARG;
    return gram_cost_launch(ctx, V, ldv, UtM, ldm, UtU, nullptr, UtU64, ldg, r, n, normx2_f64, sigma_a, bias_a, sigma_g, out_f64, stream);
}
