// NTD pieces (nn_fac/ntd.py:514-645): mode-n products of the data tensor with a transposed factor, and the projected-
// gradient update of the (small) core tensor as ONE single-workgroup launch.
//
//   nnf_ttm3_f32    out = T x_mode F^T   (tl.tenalg.mode_dot(T, F.T, mode), the building block of ntd.py:550,581)
//       mode 0: T viewed as I x (J*K), first axis contracted   -> the W^T X kernel (k_stream.hip), out[r][J][K]
//       mode 2: T viewed as (I*J) x K, last axis contracted     -> the X H^T kernel,                out[r][I][J]
//       mode 1: middle axis, per slab i: out[i][r][K] = F^T T_i -> nnf_ttm_mid_kernel below (VALU, factor row in SGPRs)
//     The two big cases stream T once with the MFMA kernels; the new axis comes out FIRST (row-major [r][rest]) so the
//     next contraction of a chain is again a first/last-axis product of a 2-D view, never a transposed copy.
//
//   nnf_ntd_core_pg_f32   ntd.py:588-619 + 639: step = round(prod 1/sigma_max(M_i), 6); up to max_iter projected-gradient
//     steps core -= min(step*grad, core) with grad = core x_0 M0 x_1 M1 x_2 M2 - MtX + sparse, stopped when the update
//     norm falls below delta times the first one; then the Gram-form reconstruction error.  The core has a few hundred
//     to a few thousand entries, so the whole loop runs in one workgroup with the core in LDS, in fp64 (the reference's
//     arithmetic: the 300-step loop is a recurrence, and the error expression cancels to ~1e-8 of its terms).
#include "nnf_internal.h"

// ---------------------------------------------------------------------------------------------------------
// middle-axis product: out[i][b][k] = sum_j Ft[b][j] * T[i][j][k].   grid = (ceil(K/256), I), 256 threads;
// thread = one k of one slab, RC rank rows at a time (the slab's J x 256 tile is re-read from L2 per rank chunk).
// ---------------------------------------------------------------------------------------------------------
template <int RC>
__global__ __launch_bounds__(256) void nnf_ttm_mid_kernel(const float* __restrict__ T, int64_t J, int64_t K,
                                                          const float* __restrict__ Ft, int64_t ldf, int r,
                                                          float* __restrict__ out) {
    const int64_t i = blockIdx.y, k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float* slab = T + i * J * K;
    float* o = out + i * (int64_t)r * K;
    for (int b0 = 0; b0 < r; b0 += RC) {
        float acc[RC];
#pragma unroll
        for (int b = 0; b < RC; ++b) acc[b] = 0.f;
        for (int64_t j = 0; j < J; ++j) {
            const float t = (k < K) ? slab[j * K + k] : 0.f;
#pragma unroll
            for (int b = 0; b < RC; ++b)
                if (b0 + b < r) acc[b] = fmaf(Ft[(int64_t)(b0 + b) * ldf + j], t, acc[b]);   // wave-uniform operand
        }
        if (k < K) {
#pragma unroll
            for (int b = 0; b < RC; ++b)
                if (b0 + b < r) o[(int64_t)(b0 + b) * K + k] = acc[b];
        }
    }
}

extern "C" int nnf_ttm3_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft, int64_t ldf, int r,
                            int mode, float* out, void* stream) {
    if (!ctx || !T || !Ft || !out || I < 1 || J < 1 || K < 1 || r < 1 || mode < 0 || mode > 2) return NNF_ERR_ARG;
    const int64_t dim = mode == 0 ? I : (mode == 1 ? J : K);
    if (ldf < dim) return NNF_ERR_ARG;
    // (ranks above 128: modes 0 and 2 are the W^T X / X H^T kernels, which walk the rank in passes; the middle axis stays <= 128)
    if (r > NNF_MAX_RANK && mode == 1) return NNF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    nnf_ws_cursor cur(ctx);
    if (mode == 0) return nnf_xty_impl(ctx, cur, T, I, J * K, J * K, Ft, r, ldf, out, J * K, st);
    if (mode == 2) return nnf_xht_impl(ctx, cur, T, I * J, K, K, Ft, r, ldf, out, I * J, st);
    // middle axis: per slab i, out_i (r x K) = Ft (r x J) T_i (J x K) -- a batch of products with one rank-sized left operand
    if ((int64_t)8 * J * 4 <= 64 * 1024 && I <= 65535)
        return nnf_small_gemm_launch(Ft, ldf, r, (int)J, T, K, K, out, K, I, J * K, (int64_t)r * K, st);
    if (I > 65535) return NNF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((nnf_ttm_mid_kernel<16>), dim3((unsigned)nnf_cdiv(K, 256), (unsigned)I), dim3(256), 0, st, T, J, K, Ft, ldf,
                       r, out);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// core update.  LDS (fp64): core, MtX, two scratch tensors (S each), M0, M1, M2, a power-iteration vector pair.
// status_f64[0..5] = {iterations done, last update norm, first update norm, step, reconstruction error, 0}
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pg_block_sum(double v, double* red) {
    v = nnf_wave_sum_f64(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];   // every thread, same order
    __syncthreads();
    return s;
}

// largest eigenvalue of the symmetric PSD r x r matrix M (LDS, fp64) by power iteration from the ones vector (the Gram of
// a non-negative factor has a positive Perron vector, so the start is never orthogonal to it); x, y: r doubles in LDS.
__device__ double pg_sigma_max(const float* M, int r, double* x, double* y, double* red) {
    for (int a = threadIdx.x; a < r; a += blockDim.x) x[a] = 1.0;
    __syncthreads();
    double lam = 0.0;
    for (int it = 0; it < 3000; ++it) {
        for (int a = threadIdx.x; a < r; a += blockDim.x) {
            double s = 0.0;
            for (int b = 0; b < r; ++b) s += (double)M[a * r + b] * x[b];
            y[a] = s;
        }
        __syncthreads();
        double n2 = 0.0, xy = 0.0, xx = 0.0;
        for (int a = threadIdx.x; a < r; a += blockDim.x) { n2 += y[a] * y[a]; xy += x[a] * y[a]; xx += x[a] * x[a]; }
        n2 = pg_block_sum(n2, red);
        xy = pg_block_sum(xy, red);
        xx = pg_block_sum(xx, red);
        const double lam_new = xy / xx;   // Rayleigh quotient
        if (n2 == 0.0) return 0.0;
        const double inv = 1.0 / sqrt(n2);
        for (int a = threadIdx.x; a < r; a += blockDim.x) x[a] = y[a] * inv;
        __syncthreads();
        if (it > 8 && fabs(lam_new - lam) <= 1e-15 * fabs(lam_new)) { lam = lam_new; break; }
        lam = lam_new;
    }
    return lam;
}

// dst = src x_mode M  (dst[.., a', ..] = sum_a M[a'][a] src[.., a, ..]),  dims d0 x d1 x d2, M is d_mode x d_mode.
// Mt is the TRANSPOSE of M with rows padded to dmp = roundup(d_mode, 4) floats (zero padding): Mt[a*dmp + a'] = M[a'][a].
// Work item = (fibre along the mode, group of four outputs a'..a'+3): the fibre element is read once per group and
// feeds four fp64 FMAs whose Gram operands come from one 16-byte LDS read at an address shared by neighbouring
// threads.  (One output per thread read the fibre AND a Gram row element for every FMA: 33 us per projected-gradient
// step on a 20^3 core.)  Items are ordered group-major so that consecutive threads walk consecutive fibres (unit stride
// in LDS) for modes 0 and 1; for the last mode a fibre IS contiguous, so there the threads of a fibre's groups are
// neighbours (one broadcast read of the fibre element, consecutive 16-byte Gram reads).
template <typename ST>
__device__ void pg_mode_dot(const ST* src, ST* dst, const float* Mt, int d0, int d1, int d2, int mode) {
    const int S = d0 * d1 * d2;
    const int dm = mode == 0 ? d0 : (mode == 1 ? d1 : d2);
    const int dmp = (dm + 3) & ~3;
    const int stride = mode == 0 ? d1 * d2 : (mode == 1 ? d2 : 1);
    const int nf = S / dm, ng = dmp >> 2;
    for (int it = threadIdx.x; it < nf * ng; it += blockDim.x) {
        const int grp = mode == 2 ? it % ng : it / nf, f = mode == 2 ? it / ng : it - grp * nf;
        // fibre f: the other two indices, in memory order
        const int base = mode == 0 ? f : (mode == 1 ? (f / d2) * (d1 * d2) + (f % d2) : f * d2);
        const float* mt = Mt + 4 * grp;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int t = 0; t < dm; ++t) {
            const double x = (double)src[base + t * stride];
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(mt + t * dmp);
            a0 += (double)g4[0] * x;
            a1 += (double)g4[1] * x;
            a2 += (double)g4[2] * x;
            a3 += (double)g4[3] * x;
        }
        const int o = 4 * grp;
        dst[base + o * stride] = (ST)a0;
        if (o + 1 < dm) dst[base + (o + 1) * stride] = (ST)a1;
        if (o + 2 < dm) dst[base + (o + 2) * stride] = (ST)a2;
        if (o + 3 < dm) dst[base + (o + 3) * stride] = (ST)a3;
    }
    __syncthreads();
}

// ST: storage type of the four core-sized arrays.  double (the reference's arithmetic) while they fit in LDS; float -- with
// every dot product and the update still accumulated in fp64 -- for cores up to ~9000 entries; beyond that (gbuf != 0)
// double again, in the context workspace.
template <typename ST>
__global__ __launch_bounds__(1024) void nnf_ntd_core_pg_kernel(float* __restrict__ core_g, const float* __restrict__ mtx_g,
                                                               const float* __restrict__ M0g, const float* __restrict__ M1g,
                                                               const float* __restrict__ M2g, int d0, int d1, int d2,
                                                               double sparse, double delta, int max_iter, double norm_sq,
                                                               double* __restrict__ status, double* gbuf) {
    // working set: LDS when it fits (the usual few-hundred-entry core), else a slice of the context workspace -- still
    // one workgroup (its waves share the CU's L1, and every phase ends in a barrier), just with L2 latency per access
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int S = d0 * d1 * d2;
    // LDS: [vx 128][vy 128][red 16] doubles, [T0][T1][T2] floats (transposed Grams, fp32 values), then (unless gbuf) the four
    // core-sized arrays of ST, 8-byte aligned
    double* vx = lds_all;
    // the Grams live in LDS transposed and row-padded for the mode products (16-byte aligned: 272 doubles in front, every
    // block a multiple of four floats); the three power iterations read the originals from global memory
    const int p0 = (d0 + 3) & ~3, p1 = (d1 + 3) & ~3, p2 = (d2 + 3) & ~3;
    float* T0 = reinterpret_cast<float*>(lds_all + 272);
    float* T1 = T0 + d0 * p0;
    float* T2 = T1 + d1 * p1;
    const int tfl = d0 * p0 + d1 * p1 + d2 * p2;      // multiple of 4
    ST* core = gbuf ? reinterpret_cast<ST*>(gbuf) : reinterpret_cast<ST*>(T0 + tfl);
    ST* mtx = core + S;
    ST* ta = mtx + S;
    ST* tb = ta + S;
    double* vy = vx + 128;
    double* red = vy + 128;   // 16 doubles
    for (int e = threadIdx.x; e < S; e += blockDim.x) { core[e] = (ST)core_g[e]; mtx[e] = (ST)mtx_g[e]; }
    for (int e = threadIdx.x; e < d0 * p0; e += blockDim.x) { const int a = e / p0, b = e - a * p0; T0[e] = b < d0 ? M0g[b * d0 + a] : 0.f; }
    for (int e = threadIdx.x; e < d1 * p1; e += blockDim.x) { const int a = e / p1, b = e - a * p1; T1[e] = b < d1 ? M1g[b * d1 + a] : 0.f; }
    for (int e = threadIdx.x; e < d2 * p2; e += blockDim.x) { const int a = e / p2, b = e - a * p2; T2[e] = b < d2 ? M2g[b * d2 + a] : 0.f; }
    __syncthreads();
    // ntd.py:592-596
    double step = 1.0;
    step *= 1.0 / pg_sigma_max(M0g, d0, vx, vy, red);
    step *= 1.0 / pg_sigma_max(M1g, d1, vx, vy, red);
    step *= 1.0 / pg_sigma_max(M2g, d2, vx, vy, red);
    step = rint(step * 1e6) / 1e6;
    // ntd.py:609-619
    int cnt = 1;
    double upd0 = 0.0, upd = 1.0;
    while (cnt <= max_iter && upd >= delta * upd0) {
        pg_mode_dot(core, ta, T0, d0, d1, d2, 0);
        pg_mode_dot(ta, tb, T1, d0, d1, d2, 1);
        pg_mode_dot(tb, ta, T2, d0, d1, d2, 2);
        double s2 = 0.0;
        for (int e = threadIdx.x; e < S; e += blockDim.x) {
            const double c = (double)core[e];
            const double grad = -(double)mtx[e] + (double)ta[e] + sparse;
            const double dc = fmin(step * grad, c);
            core[e] = (ST)(c - dc);
            s2 += dc * dc;
        }
        upd = sqrt(pg_block_sum(s2, red));
        if (cnt == 1) upd0 = upd;
        ++cnt;
    }
    // ntd.py:639 (the caller adds the sparsity terms and divides by norm_sq; it recomputes this itself if it normalises
    // the core first)
    pg_mode_dot(core, ta, T0, d0, d1, d2, 0);
    pg_mode_dot(ta, tb, T1, d0, d1, d2, 1);
    pg_mode_dot(tb, ta, T2, d0, d1, d2, 2);
    double ip = 0.0, qf = 0.0;
    for (int e = threadIdx.x; e < S; e += blockDim.x) { ip += (double)mtx[e] * (double)core[e]; qf += (double)ta[e] * (double)core[e]; }
    ip = pg_block_sum(ip, red);
    qf = pg_block_sum(qf, red);
    for (int e = threadIdx.x; e < S; e += blockDim.x) core_g[e] = (float)core[e];
    if (threadIdx.x == 0) {
        status[0] = (double)(cnt - 1);
        status[1] = upd;
        status[2] = upd0;
        status[3] = step;
        status[4] = norm_sq - 2.0 * ip + qf;
        status[5] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same loop on d0 workgroups (one per mode-0 slab of the core) for cores of a few thousand entries, where one CU spends
// ~18 us per step (300 steps = most of an NTD iteration at 300^3 / 20^3).  The three mode products of the gradient commute:
// a workgroup forms  y = slab x_1 M1 x_2 M2  locally (two small products on its d1 x d2 slab), publishes it, and after ONE grid
// barrier combines everybody's y with its row of M0:  (core x M)[a] = sum_a'' M0[a][a''] y[a''].  The squared norm of the previous
// step's update travels with y, so the stopping test of ntd.py:609 is taken right after the barrier -- BEFORE the update of the
// step it decides about is applied: no speculation, no roll-back, one barrier per step.  fp64 throughout, every workgroup
// adds the partials in index order (same doubles everywhere -> same decisions).  Exchange through agent-scope atomics
// (write-through stores / cache-bypassing loads), ping-pong buffers by step parity, bounded spins (status[5] = 1 on a time-out).
// ---------------------------------------------------------------------------------------------------------
struct pg_multi_sync {
    unsigned* counter;                 // arrivals, zeroed before the launch
    unsigned long long* Y;             // [2][d0][S1] doubles (as bits)
    unsigned long long* part;          // [2][d0][2] doubles (as bits): partial sums that travel with a barrier
};
__device__ __forceinline__ void pg_st(unsigned long long* p, double v) {
    __hip_atomic_store(p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pg_ld(const unsigned long long* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// all workgroups arrive; returns false on a time-out (workgroup-uniform).  `episode` counts barriers from 1.
__device__ __forceinline__ bool pg_grid_barrier(unsigned* counter, unsigned episode, unsigned nwg, unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's published words have reached the L2 ...
    __syncthreads();                                        // ... and so have those of every wave of the workgroup
    if (threadIdx.x == 0) {
        // every exchanged word is an agent-scope atomic (write-through store / cache-bypassing load): relaxed ordering + the
        // drain below is enough, and spares the cache write-back / invalidate an acquire-release pair costs per barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = episode * nwg;
        unsigned spins = 0, ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = 0; break; }
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0u;
}

__global__ __launch_bounds__(256) void nnf_ntd_core_pg_multi_kernel(float* __restrict__ core_g, const float* __restrict__ mtx_g,
                                                                   const float* __restrict__ M0g, const float* __restrict__ M1g,
                                                                   const float* __restrict__ M2g, int d0, int d1, int d2,
                                                                   double sparse, double delta, int max_iter, double norm_sq,
                                                                   double* __restrict__ status, pg_multi_sync sy) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int a = blockIdx.x, S1 = d1 * d2;
    const unsigned nwg = gridDim.x;
    double* vx = lds_all;            // 128
    double* vy = vx + 128;           // 128
    double* red = vy + 128;          // 16
    double* m0 = red + 16;           // d0 doubles: row a of M0 (padded to 128)
    const int p1 = (d1 + 3) & ~3, p2 = (d2 + 3) & ~3;
    float* T1 = reinterpret_cast<float*>(m0 + 128);
    float* T2 = T1 + d1 * p1;
    double* c = reinterpret_cast<double*>(T2 + d2 * p2 + ((d1 * p1 + d2 * p2) & 1));   // 8-byte aligned (both blocks are multiples of 4 floats)
    double* mx = c + S1;
    double* y = mx + S1;
    double* ta = y + S1;
    __shared__ unsigned flag;
    for (int e = threadIdx.x; e < S1; e += blockDim.x) { c[e] = (double)core_g[(size_t)a * S1 + e]; mx[e] = (double)mtx_g[(size_t)a * S1 + e]; }
    for (int e = threadIdx.x; e < d0; e += blockDim.x) m0[e] = (double)M0g[a * d0 + e];
    for (int e = threadIdx.x; e < d1 * p1; e += blockDim.x) { const int r_ = e / p1, b = e - r_ * p1; T1[e] = b < d1 ? M1g[b * d1 + r_] : 0.f; }
    for (int e = threadIdx.x; e < d2 * p2; e += blockDim.x) { const int r_ = e / p2, b = e - r_ * p2; T2[e] = b < d2 ? M2g[b * d2 + r_] : 0.f; }
    __syncthreads();
    double step = 1.0;   // ntd.py:592-596, every workgroup the same arithmetic
    step *= 1.0 / pg_sigma_max(M0g, d0, vx, vy, red);
    step *= 1.0 / pg_sigma_max(M1g, d1, vx, vy, red);
    step *= 1.0 / pg_sigma_max(M2g, d2, vx, vy, red);
    step = rint(step * 1e6) / 1e6;

    unsigned episode = 0;
    bool ok = true;
    double s2_mine = 0.0, upd0 = 0.0, upd = 1.0;
    int t = 1;
    // gradient of the current core: ta = (core x_0 M0 x_1 M1 x_2 M2)[a]; brings back the sum of the partials p0 published with it
    auto full_product = [&](double pub0, double pub1, double& sum0, double& sum1) -> bool {
        pg_mode_dot<double>(c, ta, T1, 1, d1, d2, 1);          // (ends with a barrier)
        pg_mode_dot<double>(ta, y, T2, 1, d1, d2, 2);
        ++episode;
        unsigned long long* Yp = sy.Y + (size_t)(episode & 1) * nwg * S1;
        for (int e = threadIdx.x; e < S1; e += blockDim.x) pg_st(Yp + (size_t)a * S1 + e, y[e]);
        unsigned long long* Pp = sy.part + (size_t)(episode & 1) * nwg * 2;
        if (threadIdx.x == 0) { pg_st(Pp + 2 * a, pub0); pg_st(Pp + 2 * a + 1, pub1); }
        if (!pg_grid_barrier(sy.counter, episode, nwg, &flag)) return false;
        // everybody's y against this workgroup's row of M0.  The loads of up to 32 slabs are issued together (an L2 round trip
        // each: one at a time they would be the whole step), the sum runs in slab order.
        for (int e = threadIdx.x; e < S1; e += blockDim.x) {
            double s = 0.0;
            for (unsigned q0_ = 0; q0_ < nwg; q0_ += 32) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = (q0_ + u < nwg) ? pg_ld(Yp + (size_t)(q0_ + u) * S1 + e) : 0.0;
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (q0_ + u < nwg) s += m0[q0_ + u] * v[u];
            }
            ta[e] = s;
        }
        // the travelling partial sums: lanes of the first wave fetch them, fixed-order wave sum, broadcast through LDS
        if (threadIdx.x < 64) {
            double q0 = 0.0, q1 = 0.0;
            for (unsigned q = threadIdx.x; q < nwg; q += 64) { q0 += pg_ld(Pp + 2 * q); q1 += pg_ld(Pp + 2 * q + 1); }
            q0 = nnf_wave_sum_f64(q0);
            q1 = nnf_wave_sum_f64(q1);
            if (threadIdx.x == 0) { red[8] = q0; red[9] = q1; }
        }
        __syncthreads();
        sum0 = red[8];
        sum1 = red[9];
        __syncthreads();
        return true;
    };
    while (true) {
        double n2 = 0.0, dummy = 0.0;
        ok = full_product(s2_mine, 0.0, n2, dummy);
        if (!ok) break;
        if (t >= 2) {
            upd = sqrt(n2);
            if (t == 2) upd0 = upd;
        }
        if (!(t <= max_iter && (t == 1 || upd >= delta * upd0))) break;   // ntd.py:609, decided before the update is applied
        double s2 = 0.0;
        for (int e = threadIdx.x; e < S1; e += blockDim.x) {
            const double grad = -mx[e] + ta[e] + sparse;
            const double dc = fmin(step * grad, c[e]);
            c[e] -= dc;
            s2 += dc * dc;
        }
        s2_mine = pg_block_sum(s2, red);
        ++t;
    }
    // ta = (core x M)[a] for the final core: ntd.py:639
    double ip = 0.0, qf = 0.0;
    if (ok) {
        for (int e = threadIdx.x; e < S1; e += blockDim.x) { ip += mx[e] * c[e]; qf += ta[e] * c[e]; }
        ip = pg_block_sum(ip, red);
        qf = pg_block_sum(qf, red);
        ++episode;
        unsigned long long* Pp = sy.part + (size_t)(episode & 1) * nwg * 2;
        if (threadIdx.x == 0) { pg_st(Pp + 2 * a, ip); pg_st(Pp + 2 * a + 1, qf); }
        ok = pg_grid_barrier(sy.counter, episode, nwg, &flag);
        if (ok && threadIdx.x < 64) {
            double q0 = 0.0, q1 = 0.0;
            for (unsigned q = threadIdx.x; q < nwg; q += 64) { q0 += pg_ld(Pp + 2 * q); q1 += pg_ld(Pp + 2 * q + 1); }
            ip = nnf_wave_sum_f64(q0);
            qf = nnf_wave_sum_f64(q1);
        }
    }
    for (int e = threadIdx.x; e < S1; e += blockDim.x) core_g[(size_t)a * S1 + e] = (float)c[e];
    if (a == 0 && threadIdx.x == 0) {
        status[0] = (double)(t - 1);
        status[1] = upd;
        status[2] = upd0;
        status[3] = step;
        status[4] = norm_sq - 2.0 * ip + qf;
        status[5] = ok ? 0.0 : 1.0;
    }
}

extern "C" int nnf_ntd_core_pg_f32(nnf_ctx* ctx, float* core, const float* MtX, const float* M0, const float* M1, const float* M2,
                                   int d0, int d1, int d2, double sparse, double delta, int max_iter, double norm_sq,
                                   double* status_f64, void* stream) {
    if (!ctx || !core || !MtX || !M0 || !M1 || !M2 || !status_f64 || d0 < 1 || d1 < 1 || d2 < 1 || max_iter < 0) return NNF_ERR_ARG;
    if (d0 > 128 || d1 > 128 || d2 > 128) return NNF_ERR_UNSUPPORTED;
    const int64_t S = (int64_t)d0 * d1 * d2;
    if (S > ((int64_t)1 << 22)) return NNF_ERR_UNSUPPORTED;
    const size_t fixed = (size_t)272 * 8 + ((size_t)d0 * ((d0 + 3) & ~3) + (size_t)d1 * ((d1 + 3) & ~3) + (size_t)d2 * ((d2 + 3) & ~3)) * 4;
    const size_t lim = (size_t)160 * 1024;
    const int threads = S >= 1024 ? 1024 : (S >= 512 ? 512 : 256);
    hipStream_t st = (hipStream_t)stream;
    {   // a few thousand entries and several mode-0 slabs: one workgroup per slab (NNF_NTD_PG_MULTI=0: the one-workgroup form)
        static const bool multi_ok = !(getenv("NNF_NTD_PG_MULTI") && getenv("NNF_NTD_PG_MULTI")[0] == '0');
        const int64_t S1 = (int64_t)d1 * d2;
        const size_t shm = (size_t)(128 + 128 + 16 + 128) * 8 + ((size_t)d1 * ((d1 + 3) & ~3) + (size_t)d2 * ((d2 + 3) & ~3) + 2) * 4 +
                           (size_t)4 * S1 * 8;
        if (multi_ok && S >= 2048 && d0 >= 4 && d0 <= ctx->num_cus && shm <= (size_t)150 * 1024) {
            nnf_ws_cursor cur(ctx);
            unsigned* counter = (unsigned*)cur.take(256);
            unsigned long long* Y = (unsigned long long*)cur.take((size_t)2 * S * 8);
            unsigned long long* part = (unsigned long long*)cur.take((size_t)2 * d0 * 2 * 8);
            if (!counter || !Y || !part) return NNF_ERR_WORKSPACE;
            if (hipMemsetAsync(counter, 0, 4, st) != hipSuccess) return NNF_ERR_LAUNCH;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_ntd_core_pg_multi_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
                return NNF_ERR_LAUNCH;
            hipLaunchKernelGGL(nnf_ntd_core_pg_multi_kernel, dim3(d0), dim3(256), shm, st, core, MtX, M0, M1, M2, d0, d1, d2, sparse,
                               delta, max_iter, norm_sq, status_f64, pg_multi_sync{counter, Y, part});
            NNF_CHECK_LAUNCH();
            return NNF_OK;
        }
    }
    if (fixed + (size_t)4 * S * 8 <= lim) {          // everything in LDS, fp64 storage
        const size_t shm = fixed + (size_t)4 * S * 8;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_ntd_core_pg_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return NNF_ERR_LAUNCH;
        hipLaunchKernelGGL(nnf_ntd_core_pg_kernel<double>, dim3(1), dim3(threads), shm, st, core, MtX, M0, M1, M2, d0, d1, d2, sparse,
                           delta, max_iter, norm_sq, status_f64, (double*)nullptr);
    } else if (fixed + (size_t)4 * S * 4 <= lim) {   // everything in LDS, fp32 storage, fp64 accumulation
        const size_t shm = fixed + (size_t)4 * S * 4;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_ntd_core_pg_kernel<float>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return NNF_ERR_LAUNCH;
        hipLaunchKernelGGL(nnf_ntd_core_pg_kernel<float>, dim3(1), dim3(threads), shm, st, core, MtX, M0, M1, M2, d0, d1, d2, sparse,
                           delta, max_iter, norm_sq, status_f64, (double*)nullptr);
    } else {                                          // core-sized arrays in the context workspace (slow: L2 latency per access)
        nnf_ws_cursor cur(ctx);
        double* gbuf = (double*)cur.take((size_t)4 * S * 8);
        if (!gbuf) return NNF_ERR_WORKSPACE;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_ntd_core_pg_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fixed) != hipSuccess)
            return NNF_ERR_LAUNCH;
        hipLaunchKernelGGL(nnf_ntd_core_pg_kernel<double>, dim3(1), dim3(threads), fixed, st, core, MtX, M0, M1, M2, d0, d1, d2, sparse,
                           delta, max_iter, norm_sq, status_f64, gbuf);
    }
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
