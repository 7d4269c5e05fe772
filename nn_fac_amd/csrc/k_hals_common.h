// Shared by k_hals.hip (entry points, generic path) and k_hals_fast.hip (register-resident fast path).
#pragma once
#include "nnf_internal.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HALS_SPIN_LIMIT (1u << 22)


struct hals_sync {
    unsigned* counter;   // generic path: monotonic arrival counter (zeroed by the prep kernel)
    double* slots;       // generic path: [2][nblocks][4] partials, parity-double-buffered
    double* sslots;      // fast path: [max_sweeps + 2][nblocks][2] tagged 8-byte granules (hals_publish)
    unsigned epoch;      // fast path: per-call tag salt (nnf_ctx::hals_epoch)
};

// Exchange up to 3 doubles between all workgroups; returns sums of v0, v1 and the max of v2 in out[0..2].
// `epoch` counts barrier episodes from 1.  Returns false on timeout (wave-uniform after the barrier).
template <int NV>
__device__ __forceinline__ bool grid_exchange(const hals_sync& sy, unsigned epoch, int nblocks, const double (&mine)[NV],
                                              double (&out)[NV], double* red, unsigned* lds_flag) {
    double* slot = sy.slots + ((size_t)(epoch & 1) * nblocks + blockIdx.x) * 4;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + i), __builtin_bit_cast(unsigned long long, mine[i]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)nblocks;
        unsigned spins = 0, ok = 1;
        while (__hip_atomic_load(sy.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > HALS_SPIN_LIMIT) { ok = 0; break; }
        }
        *lds_flag = ok;
    }
    __syncthreads();
    const bool ok = (*lds_flag != 0);
    // every workgroup sums every partial in the same order
    double s[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = (i == 2) ? -1.0e300 : 0.0;
    const double* base = sy.slots + (size_t)(epoch & 1) * nblocks * 4;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double x = __builtin_bit_cast(
                double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(base + (size_t)b * 4 + i),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (i == 2) s[i] = x > s[i] ? x : s[i]; else s[i] += x;
        }
    }
    // block reduce (fixed order) and broadcast through LDS
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double v = s[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double y = __shfl_down(v, o, 64);
            if (i == 2) v = y > v ? y : v; else v += y;
        }
        if ((threadIdx.x & 63) == 0) red[w * NV + i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double v = red[i];
        for (int k = 1; k < nw; ++k) {
            const double y = red[k * NV + i];
            if (i == 2) v = y > v ? y : v; else v += y;
        }
        out[i] = v;
    }
    __syncthreads();
    return ok;
}


// Fast path, non-blocking form of the exchange (cdna_hip_programming.md Guideline 16, form R2: the data IS the flag).
// A workgroup's fp64 partial of sweep s travels as two 8-byte granules {tag, lo32} {tag, hi32}, each ONE write-through
// (sc1) store: fire and forget, no drain, no counter.  tag = epoch*1024 + s with a per-call epoch from the context, so
// words left behind by earlier solves never match (the workspace is zeroed once at context creation).
// collect: every thread re-reads its share of the granules (sc1 loads) until both tags match (bounded), then all
// workgroups sum all partials in index order -> the same double everywhere, bit for bit.
__device__ __forceinline__ void hals_publish(const hals_sync& sy, int s, int nblocks, double mine) {
    if (threadIdx.x == 0) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, mine);
        const unsigned long long tag = (unsigned long long)(sy.epoch * 1024u + (unsigned)s) << 32;
        unsigned long long* g = reinterpret_cast<unsigned long long*>(sy.sslots) + ((size_t)s * nblocks + blockIdx.x) * 2;
        __hip_atomic_store(g, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// The granule loads of a collect can be issued a whole sweep ahead of their use (lag-one speculation: the sum of sweep
// s-1 is only needed after sweep s has been computed): hals_collect_issue() starts them, hals_collect() consumes them
// and falls back to re-reading (bounded spin) any granule whose tag had not arrived yet.  Every workgroup adds the
// partials in the same order (thread-strided by index, DPP wave sum, waves in index order): same double everywhere.
constexpr int HALS_PF = 2;   // granule pairs per thread that can be in flight (covers nblocks <= 2 * blockDim.x)
struct hals_prefetch {
    unsigned long long g0[HALS_PF], g1[HALS_PF];
    int s;   // sweep the granules belong to (0: nothing issued)
};
__device__ __forceinline__ void hals_collect_issue(const hals_sync& sy, int s, int nblocks, hals_prefetch& pf) {
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    pf.s = s;
#pragma unroll
    for (int i = 0; i < HALS_PF; ++i) {
        const int b = threadIdx.x + i * blockDim.x;
        pf.g0[i] = pf.g1[i] = 0ull;
        if (b < nblocks) {
            pf.g0[i] = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pf.g1[i] = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ bool hals_collect(const hals_sync& sy, int s, int nblocks, double& total, double* red,
                                             unsigned* lds_flag, const hals_prefetch* pf = nullptr) {
    if (threadIdx.x == 0) *lds_flag = 1u;
    __syncthreads();
    const unsigned tag = sy.epoch * 1024u + (unsigned)s;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    double v = 0.0;
    int i = 0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x, ++i) {
        unsigned long long g0 = 0ull, g1 = 0ull;
        if (pf != nullptr && pf->s == s) {   // prefetched copy (tag 0 never matches: epochs start at 1)
#pragma unroll
            for (int u = 0; u < HALS_PF; ++u)
                if (u == i) { g0 = pf->g0[u]; g1 = pf->g1[u]; }
        }
        unsigned spins = 0;
        while (!((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag)) {
            if (spins > 0) __builtin_amdgcn_s_sleep(1);
            if (++spins > HALS_SPIN_LIMIT) { *lds_flag = 0u; break; }
            g0 = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g1 = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        v += __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xffffffffull));
    }
    v = nnf_wave_sum_f64(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    const bool ok = (*lds_flag != 0);
    double t = red[0];
    for (int k = 1; k < nw; ++k) t += red[k];
    total = t;
    __syncthreads();
    return ok;
}

// Issue point of the exchange prefetch INSIDE a sweep.  With lag-one speculation the global sum of sweep s-1 is consumed after
// sweep s; its granule loads used to go out right before sweep s -- when the workgroups running a little behind have not
// published yet: 46 % of the granules came back stale (counted) and every sweep paid one or two more L2 round trips in the
// collect (9.4 against 8.05 us per sweep for the blind sweeps, whatever the rank).  Issued half a sweep later everybody's
// partial is there.  The issue is four hand-written vector loads from addresses prepared BEFORE the sweep: no scalar
// instruction, no branch, no exec mask in the middle of the hand-scheduled scalar loads (a C++ version there made hipcc
// shuffle in-flight scalar destinations: tools/check_sweep_spills.py).  Threads without a granule of their own read
// granule 0 and ignore it.  The consumer waits with vmcnt(0) (hals_mid_wait) before it looks at the registers.
#ifndef HALS_DBG
#define HALS_DBG 0             // timing-only ablations of the exchange (tools/probes/exchange_cost_probe.py): 1 granules taken as they
#endif                         // come (no tag wait), 2 no collect at all, 4 no publish either; decisions are ignored then
#ifndef HALS_LATE_ISSUE
#define HALS_LATE_ISSUE 1      // 0: the exchange prefetch goes out before the sweep (A/B builds)
#endif
#ifndef HALS_MID_AT
#define HALS_MID_AT(R) ((R) - 1)
#endif
struct hals_mid_none {
    __device__ __forceinline__ void operator()() const {}
};
struct hals_mid_issue {
    unsigned long long a0, a1;
    hals_prefetch& pf;
    __device__ __forceinline__ void operator()() const {
        static_assert(HALS_PF == 2, "two granule pairs per thread");
        asm volatile("global_load_dwordx2 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %4, off offset:8 sc1\n\t"
                     "global_load_dwordx2 %2, %5, off sc1\n\tglobal_load_dwordx2 %3, %5, off offset:8 sc1"
                     : "=&v"(pf.g0[0]), "=&v"(pf.g1[0]), "=&v"(pf.g0[1]), "=&v"(pf.g1[1])
                     : "v"(a0), "v"(a1));
    }
};
__device__ __forceinline__ void hals_mid_wait(hals_prefetch& pf) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf.g0[0]), "+v"(pf.g1[0]), "+v"(pf.g0[1]), "+v"(pf.g1[1]));
}

// Barrier-lean forms for the persistent sweep loop: `red` is a per-sweep-parity slot array ((blockDim/64) doubles each),
// so a slot is rewritten only two sweeps later, with a barrier in between -- no trailing barrier is needed, and the
// time-out flag is armed once before the loop instead of per call.  One __syncthreads each.
__device__ __forceinline__ double hals_block_sum1(double v, double* red_p) {
    v = nnf_wave_sum_f64(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red_p[w] = v;
    __syncthreads();
    double t = red_p[0];
    for (int k = 1; k < nw; ++k) t += red_p[k];
    return t;   // every thread
}
__device__ __forceinline__ bool hals_collect1(const hals_sync& sy, int s, int nblocks, double& total, double* red_p,
                                              unsigned* lds_flag, const hals_prefetch& pf) {
    const unsigned tag = sy.epoch * 1024u + (unsigned)s;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    double v = 0.0;
    int i = 0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x, ++i) {
        unsigned long long g0 = 0ull, g1 = 0ull;
        if (pf.s == s) {
#pragma unroll
            for (int u = 0; u < HALS_PF; ++u)
                if (u == i) { g0 = pf.g0[u]; g1 = pf.g1[u]; }
        }
        unsigned spins = 0;
        while (!((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag)) {
            if (spins > 0) __builtin_amdgcn_s_sleep(1);
            if (++spins > HALS_SPIN_LIMIT) { *lds_flag = 0u; break; }
            g0 = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g1 = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        v += __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xffffffffull));
    }
    v = nnf_wave_sum_f64(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red_p[w] = v;
    __syncthreads();
    double t = red_p[0];
    for (int k = 1; k < nw; ++k) t += red_p[k];
    total = t;
    return *lds_flag != 0u;
}

// Single-wave workgroups (k_hals_quad.hip): the same sum in the same order -- granules strided over the 64 lanes, DPP wave
// sum -- with no LDS and no barrier at all.  Returns false on a time-out (wave-uniform).
__device__ __forceinline__ bool hals_collect_wave(const hals_sync& sy, int s, int nblocks, double& total,
                                                  const hals_prefetch* pf = nullptr) {
    const unsigned tag = sy.epoch * 1024u + (unsigned)s;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    double v = 0.0;
    bool late = false;
    int i = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64, ++i) {
        unsigned long long g0 = 0ull, g1 = 0ull;
        if (pf != nullptr && pf->s == s) {
#pragma unroll
            for (int u = 0; u < HALS_PF; ++u)
                if (u == i) { g0 = pf->g0[u]; g1 = pf->g1[u]; }
        }
        unsigned spins = 0;
        while (!((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag)) {
            if (spins > 0) __builtin_amdgcn_s_sleep(1);
            if (++spins > HALS_SPIN_LIMIT) { late = true; break; }
            g0 = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g1 = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        v += __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xffffffffull));
    }
    total = nnf_wave_sum_f64(v);
    return __ballot(late) == 0ull;
}

// block sum of this sweep's partial AND collect of an earlier sweep's global sum behind ONE barrier: both are "every wave
// leaves a double in LDS, everybody adds the four up", so the two values of a wave travel together (red_p: two slot arrays
// of the current sweep parity).  Returns the block sum in `bs` (every thread), the global sum of sweep c in `total`.
// For workgroups of exactly NT = 256 threads (the resident lane kernel): the thread count is a compile-time constant --
// `blockDim.x` is a load from the implicit kernel arguments, i.e. a memory round trip per sweep in the middle of the
// exchange (s_load + global_load_ushort + s_waitcnt in the ISA of round 2's kernel) -- and the two granule pairs of a thread
// are handled by straight-line code; only a granule that has not arrived takes the re-read loop.
template <int NT>
__device__ __forceinline__ bool hals_sum_collect1(const hals_sync& sy, double nd, double& bs, int c, int nblocks, double& total,
                                                  double* red_bs, double* red_tot, unsigned* lds_flag,
                                                  const hals_prefetch& pf) {
    static_assert(NT == 256 && HALS_PF == 2, "two granule pairs per thread cover nblocks <= 512");
    const unsigned tag = sy.epoch * 1024u + (unsigned)c;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)c * nblocks * 2;
    // fast path: both pairs arrived with the prefetch -- two compares per pair, ONE wave-uniform test, no exec-mask branches;
    // a granule that is not there yet sends the whole wave through the re-read loop (rare; bounded)
    unsigned long long g0[HALS_PF], g1[HALS_PF];
    bool want[HALS_PF], need = false;
#pragma unroll
    for (int i = 0; i < HALS_PF; ++i) {
        want[i] = (int)threadIdx.x + NT * i < nblocks;
        g0[i] = (pf.s == c) ? pf.g0[i] : 0ull;
        g1[i] = (pf.s == c) ? pf.g1[i] : 0ull;
        need = need || (want[i] && !((unsigned)(g0[i] >> 32) == tag && (unsigned)(g1[i] >> 32) == tag));
    }
    if (!(HALS_DBG & 1) && __builtin_expect(__ballot(need) != 0ull, 0)) {
#pragma unroll 1
        for (int i = 0; i < HALS_PF; ++i) {
            const int b = (int)threadIdx.x + NT * i;
            unsigned long long a0 = i == 0 ? g0[0] : g0[HALS_PF - 1], a1 = i == 0 ? g1[0] : g1[HALS_PF - 1];
            if (b < nblocks) {
                unsigned spins = 0;
                while (!((unsigned)(a0 >> 32) == tag && (unsigned)(a1 >> 32) == tag)) {
                    if (spins > 0) __builtin_amdgcn_s_sleep(1);
                    if (++spins > HALS_SPIN_LIMIT) { *lds_flag = 0u; break; }
                    a0 = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a1 = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (i == 0) { g0[0] = a0; g1[0] = a1; } else { g0[HALS_PF - 1] = a0; g1[HALS_PF - 1] = a1; }
        }
    }
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < HALS_PF; ++i)
        v += want[i] ? __builtin_bit_cast(double, (g1[i] << 32) | (g0[i] & 0xffffffffull)) : 0.0;
    v = nnf_wave_sum_f64(v);
    nd = nnf_wave_sum_f64(nd);
    constexpr int nw = NT / 64;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red_bs[w] = nd; red_tot[w] = v; }
    __syncthreads();
    double t = red_tot[0], b2 = red_bs[0];
#pragma unroll
    for (int k = 1; k < nw; ++k) { t += red_tot[k]; b2 += red_bs[k]; }
    total = t;
    bs = b2;
    return *lds_flag != 0u;
}
template <int NT>
__device__ __forceinline__ double hals_block_sum1(double v, double* red_p) {
    v = nnf_wave_sum_f64(v);
    constexpr int nw = NT / 64;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red_p[w] = v;
    __syncthreads();
    double t = red_p[0];
#pragma unroll
    for (int k = 1; k < nw; ++k) t += red_p[k];
    return t;   // every thread
}

struct hals_args {
    const float* UtM; int64_t ldm;
    const float* Gp;      // padded Gram  RP x RP (zeros outside r x r), workspace
    const float* dinv;    // RP pairs (1/diag, nz) (0, 0 where the diagonal is 0 or padded), then the all-live flag; workspace
    const float* Gs;      // lane layout, 32 < RP <= 64: Gram rows scaled by 1/diag (k_hals_fast.hip), workspace
    float* V; int64_t ldv;
    int r; int64_t ncols;
    int max_sweeps; double delta; float sp;
    int mode;             // 0: solve (stopping rule on device)  1: fixed sweep count, per-sweep local partials
    hals_sync sy;
    double* status;
    double* sweep_partials;
    float* snapshots;     // mode 1, optional: V after every sweep, [sweep][r][ncols]
    int64_t snap_stride;
    int sweep0;           // mode 0: sweeps already done by earlier launches of the same solve (nnf_hals_solve_continue_f32)
    const float* Vsrc;    // start values (r x ncols, row stride ldvs); == V for an in-place solve (quad kernel; the other
    int64_t ldvs;         // layouts get a copy made by the entry point)
    const float* Mimg;    // k_hals_mfma.hip: Gram image in MFMA fragment order, in-block couplings; padded rank of the dinv table
    const float* Mlt;
    int rp;
    const float* resid_in;   // k_hals_mfma.hip, mode 1: residual state of the solve so far / where to leave it (may be NULL)
    float* resid_out;
    int snap_first;          // mode 1: the first sweep (0-based, within this launch) that writes a snapshot
};

// Continuation of a solve longer than one launch can tag (NNF_HALS_MAX_SWEEPS): `status` holds the state the previous launch
// left.  Returns false when that launch already ended the solve (stopping rule, error) -- the caller returns at once, V and
// the status block stay as they are; else eps0 / eps are taken over.  Every workgroup reads the same words (written by the
// previous launch, stream-ordered; rewritten by this one only after every workgroup has published its first sweep).
__device__ __forceinline__ bool hals_take_over(const double* status, int sweep0, double delta, double& eps0, double& eps) {
    const double pe = status[NNF_HALS_ST_EPS], pe0 = status[NNF_HALS_ST_EPS0];
    if ((int)status[NNF_HALS_ST_CNT] - 1 != sweep0 || !(pe >= delta * pe0) || status[NNF_HALS_ST_ERR] != 0.0) return false;
    eps0 = pe0;
    eps = pe;
    return true;
}

int nnf_hals_fast_part0(nnf_ctx*, int RP, const hals_args&, int max_blocks_cap, int* nblocks_out, hipStream_t);
int nnf_hals_fast_part1(nnf_ctx*, int RP, const hals_args&, int max_blocks_cap, int* nblocks_out, hipStream_t);
int nnf_hals_fast_part2(nnf_ctx*, int RP, const hals_args&, int max_blocks_cap, int* nblocks_out, hipStream_t);
int nnf_hals_fast_part3(nnf_ctx*, int RP, const hals_args&, int max_blocks_cap, int* nblocks_out, hipStream_t);

// k_hals_wave.hip: one wave per column (lane = row), push form of the sweep; solve mode, <= 4800 columns
bool nnf_hals_wave_fits(nnf_ctx*, int r, int64_t ncols, int max_blocks_cap);
size_t nnf_hals_wave_gram_floats(int r);
size_t nnf_hals_wave_snap_floats(int r, int64_t ncols);
int nnf_hals_wave_run(nnf_ctx*, const float* UtU, const float* UtU2, int64_t ldg, float* Gw, float* snap, unsigned* counter,
                      hals_args a, int* nblocks_out, hipStream_t);

// k_hals_mfma.hip: push form on the matrix cores, many columns, ranks 48..100 (resident columns only)
bool nnf_hals_mfma_supported(int RP);
size_t nnf_hals_mfma_gram_floats(int RP);
size_t nnf_hals_mfma_resid_floats(int RP, int64_t ncols);
int nnf_hals_mfma_run(nnf_ctx*, int RP, const float* UtU, int64_t ldg, float* gram, hals_args a, int max_blocks_cap, int* nblocks_out,
                      hipStream_t);

// k_hals_quad.hip: four lanes per column, for solves with few columns
bool nnf_hals_quad_fits(nnf_ctx*, int r, int64_t ncols, int max_blocks_cap);
size_t nnf_hals_quad_gram_floats(int r);
int nnf_hals_quad_run(nnf_ctx*, const float* UtU, const float* UtU2, int64_t ldg, float* Gq, unsigned* counter, hals_args a,
                      int* nblocks_out, hipStream_t);
