// HALS-NNLS solves with FEW columns (the r x n "V side" of NMF at n = 2000, the 500 x R factors of NTF, the replicated
// 100 x 4000 solve of the row-sharded 1e6 x 4000 problem): ONE WAVE PER COLUMN, lane i <-> row i, "push" form of the sweep.
//
// The sweep of nnls.py:158-170 is r SEQUENTIAL row updates; each of the other layouts (k_hals_fast.hip: a lane per column,
// k_hals_quad.hip: four lanes per column) spends a whole dot product G[k,:].v -- r/2 or r/8 packed FMAs plus a cross-lane
// reduction -- on the critical path of every row, so a solve with few columns is a handful of lone waves crawling through
// ~r * 17 dependent instructions per sweep (3.5 us per sweep at 50 x 2000, 125 waves, 95 % of the chip idle).  Here the
// residual of EVERY row is kept up to date instead (one VGPR: lane i holds the scaled residual of row i)
//
//        acc[i] = ( UtM[i] - sp - sum_j UtU[i][j] v[j] ) / UtU[i][i]                      (all v[j] current)
//
// and a row update is      d = max(acc[k], -v[k]) ;  v[k] += d ;  acc[i] -= G'[i][k] * d  for all i      (G' = D^-1 UtU)
// -- the reference's statement with the dot product read off the residual: v_max (all lanes, lane k's value is the step),
// v_readlane (the step to an SGPR), v_fma (every lane pushes the step into its own residual), v_writelane (lane k keeps
// its step): FOUR instructions per row, three of them dependent, whatever the rank.  Column k of G' comes from an LDS image
// shared by the workgroup's waves, prefetched a block of eight rows ahead.  There are n waves instead of n/16: 2000 columns
// = two waves on every SIMD of 250 CUs.
//
// Rounding: the residual is formed from scratch (b' - G'v) when the kernel starts and pushed forward from then on; each
// push rounds once relative to the residual itself (which shrinks as the solve converges), not to the dot product, so the
// carried residual is as accurate as a freshly evaluated fp32 one.  Results differ from the other layouts in
// the last bits, like those differ from each other.
//
// Stopping rule (nnls.py:156) on the device, no barrier anywhere in the sweep loop:
//   * a wave's fp32 sum of squared steps (DPP) -> fp64 -> its slot of an LDS ring; the LAST wave of the workgroup to arrive
//     for a sweep (LDS counter) adds the slots in wave order and publishes the block sum as two tagged granules (k_hals_common.h);
//   * every wave collects the global sum of sweep s-2 after sweep s (its granule loads went out after sweep s-1, when the
//     other workgroups had published): lag-TWO speculation -- a sweep is ~0.5 us here, shorter than an L2 round trip under
//     load.  A snapshot of the column costs one VGPR, so the two sweeps run ahead are undone from registers.
//   * rows with a zero Gram diagonal (nnls.py:160) and the padding rows hold residual 0 and "-v" = -inf: their step is 0.
#include "k_hals_common.h"

typedef float f32x2w __attribute__((ext_vector_type(2)));

constexpr int WAVE_MAX_NW = 16;      // waves (= columns) per workgroup
constexpr int WAVE_RING = 4;         // LDS ring of per-sweep wave partials (a wave is at most 3 sweeps ahead of another)
constexpr int WAVE_PF = 8;           // granule pairs per lane a collect can hold: nblocks <= 512

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float wave_dpp_add_f32(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, m);
}
// fixed-order fp32 sum over the 64 lanes (same DPP ladder as nnf_wave_sum_f64), wave-uniform result
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = wave_dpp_add_f32<0xB1, 0xf>(v);
    v = wave_dpp_add_f32<0x4E, 0xf>(v);
    v = wave_dpp_add_f32<0x141, 0xf>(v);
    v = wave_dpp_add_f32<0x140, 0xf>(v);
    v = wave_dpp_add_f32<0x142, 0xa>(v);
    v = wave_dpp_add_f32<0x143, 0xc>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// prep: image of G' by COLUMNS.  img[k][lane] (RL = 1) or img[k][lane] = {row lane, row lane + 64} (RL = 2):
//   G'[i][k] = UtU[i][k] / UtU[i][i]   (0 where the diagonal is 0, outside r x r, in the padding rows k >= r)
// read as UtU[k][i] (symmetric) so that a workgroup reads one contiguous Gram row.  Then 1/diag per row, counter, status.
__global__ void nnf_hals_prep_wave_kernel(const float* __restrict__ UtU, const float* __restrict__ UtU2, int64_t ldg, int r, int RL,
                                          float* __restrict__ img, float* __restrict__ dinv, unsigned* counter, double* status) {
    const int k = blockIdx.x;                   // image row = Gram column
    const int W = 64 * RL;
    auto gram = [&](int a, int b) -> float {
        const float g = UtU[(int64_t)a * ldg + b];
        return UtU2 ? g * UtU2[(int64_t)a * ldg + b] : g;
    };
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        const int i = (RL == 2) ? ((c >> 1) + 64 * (c & 1)) : c;    // RL = 2: float2 {row lane, row lane + 64}
        float val = 0.f;
        if (k < r && i < r) {
            const float d = gram(i, i);
            if (d != 0.f) val = gram(k, i) * (float)(1.0 / (double)d);
        }
        img[(size_t)k * W + c] = val;
    }
    if (k < r && threadIdx.x == 0) {
        const float d = gram(k, k);
        dinv[k] = (d != 0.f) ? (float)(1.0 / (double)d) : 0.f;
    }
    if (blockIdx.x == 0) {
        for (int i = r + threadIdx.x; i < 128; i += blockDim.x) dinv[i] = 0.f;
        if (threadIdx.x == 0) {
            *counter = 0u;
            if (status) {
                status[NNF_HALS_ST_EPS] = 1.0;
                status[NNF_HALS_ST_CNT] = 1.0;
                status[NNF_HALS_ST_EPS0] = 0.0;
                status[NNF_HALS_ST_ERR] = 0.0;
            }
        }
    }
}

struct wave_prefetch {
    unsigned long long g0[WAVE_PF], g1[WAVE_PF];
    int s;
};
__device__ __forceinline__ void wave_collect_issue(const hals_sync& sy, int s, int nblocks, int lane, wave_prefetch& pf) {
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    pf.s = s;
#pragma unroll
    for (int i = 0; i < WAVE_PF; ++i) {
        const int b = lane + 64 * i;
        pf.g0[i] = pf.g1[i] = 0ull;
        if (b < nblocks) {
            pf.g0[i] = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pf.g1[i] = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// global sum of sweep s: granules strided over the lanes (prefetched copies first, bounded re-reads for late ones), added in
// index order per lane, DPP wave sum -- the same double in every wave of every workgroup.  false = time-out (wave-uniform).
__device__ __forceinline__ bool wave_collect(const hals_sync& sy, int s, int nblocks, int lane, double& total, const wave_prefetch& pf) {
    const unsigned tag = sy.epoch * 1024u + (unsigned)s;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    double v = 0.0;
    bool late = false;
#pragma unroll
    for (int i = 0; i < WAVE_PF; ++i) {
        const int b = lane + 64 * i;
        if (b < nblocks) {
            unsigned long long g0 = (pf.s == s) ? pf.g0[i] : 0ull, g1 = (pf.s == s) ? pf.g1[i] : 0ull;
            unsigned spins = 0;
            while (!((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag)) {
                if (spins > 0) __builtin_amdgcn_s_sleep(1);
                if (++spins > HALS_SPIN_LIMIT) { late = true; break; }
                g0 = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g1 = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            v += __builtin_bit_cast(double, (g1 << 32) | (g0 & 0xffffffffull));
        }
    }
    total = nnf_wave_sum_f64(v);
    return __ballot(late) == 0ull;
}

// eight row updates k = kb .. kb+7, all in the half H of the rows (H = 0: rows 0..63, lane = row; H = 1: rows 64..127)
template <int RL, int H>
__device__ __forceinline__ void wave_rows8(const float (&g)[8][RL], int kb, float (&acc)[RL], const float (&nvd)[RL], float (&dk)[RL]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int kl = (kb + u) & 63;
        float d;   // max(acc, -v) in one instruction (fmaxf adds a canonicalising max); lane kl's value is the step of row k
        asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(acc[H]), "v"(nvd[H]));
        const float sd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), kl));
#pragma unroll
        for (int h = 0; h < RL; ++h) acc[h] = fmaf(g[u][h], -sd, acc[h]);
        dk[H] = __builtin_bit_cast(float, __builtin_amdgcn_writelane(__builtin_bit_cast(int, sd), kl, __builtin_bit_cast(int, dk[H])));
    }
}

template <int RL>
__global__ __launch_bounds__(64 * WAVE_MAX_NW) void nnf_hals_wave_kernel(hals_args a, int RU) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    // LDS: image RU x (64 RL) floats | ring of wave partials WAVE_RING x NW doubles | arrival counters WAVE_RING
    const int W = 64 * RL;
    const int NW = blockDim.x >> 6;
    double* part = reinterpret_cast<double*>(wlds + (size_t)RU * W);
    unsigned* arrive = reinterpret_cast<unsigned*>(part + WAVE_RING * WAVE_MAX_NW);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblocks = gridDim.x;
    const int64_t col = (int64_t)blockIdx.x * NW + w;
    const bool valid = col < a.ncols;
    for (int e = threadIdx.x; e < RU * W / 4; e += blockDim.x)
        reinterpret_cast<f32x4*>(wlds)[e] = reinterpret_cast<const f32x4*>(a.Gp)[e];
    if (threadIdx.x < WAVE_RING) arrive[threadIdx.x] = 0u;

    // this lane's rows: lane (and lane + 64); start values from a.Vsrc (== a.V for an in-place solve)
    float v[RL], acc[RL], nvd[RL], dk[RL], v1[RL], v2[RL];
    bool dead[RL];
#pragma unroll
    for (int h = 0; h < RL; ++h) {
        const int row = lane + 64 * h;
        const bool in = valid && row < a.r;
        const float di = a.dinv[row];                     // 0: zero diagonal or padding row (128 entries are always there)
        dead[h] = !(in && di != 0.f);
        v[h] = in ? a.Vsrc[(int64_t)row * a.ldvs + col] : 0.f;
        const float bm = in ? a.UtM[(int64_t)row * a.ldm + col] : 0.f;
        acc[h] = dead[h] ? 0.f : (bm - a.sp) * di;
        nvd[h] = dead[h] ? -__builtin_inff() : -v[h];
        v1[h] = v2[h] = v[h];
    }
    __syncthreads();                                      // the only barrier: the image is in LDS
    const float* img = wlds + (RL == 2 ? 2 * lane : lane);
    // residual from scratch: acc = b' - G' v   (columns of G' one by one, v[j] through an SGPR)
    for (int jb = 0; jb < RU; jb += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = jb + u;
            const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (RL == 2 && j >= 64) ? v[RL - 1] : v[0]), j & 63));
#pragma unroll
            for (int h = 0; h < RL; ++h) acc[h] = fmaf(img[(size_t)j * W + h], -vj, acc[h]);
        }
    }

    double eps0 = 0.0, eps = 1.0;
    int done = 0;
    bool ok = true, stopped = false;
    wave_prefetch pf;
    pf.s = 0;
    // first block of image columns (the pipeline then runs across sweeps: the image is the same every sweep)
    float g[2][8][RL];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int h = 0; h < RL; ++h) g[0][u][h] = img[(size_t)u * W + h];

    auto decide = [&](int c, double tot) {       // nnls.py:156 after sweep c; true = sweep c was the last one
        if (c == 1) eps0 = tot;
        eps = tot;
        done = c;
        return !(eps >= a.delta * eps0);
    };

    for (int s = 1; s <= a.max_sweeps; ++s) {
#pragma unroll
        for (int h = 0; h < RL; ++h) dk[h] = 0.f;
        for (int kb = 0; kb < RU; kb += 16) {    // two blocks of eight rows per trip: the two register sets alternate statically
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int k0 = kb + 8 * half;
                if (k0 < RU) {
                    int kn = k0 + 8;             // the block after this one (wraps to the next sweep's first block)
                    if (kn >= RU) kn = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int h = 0; h < RL; ++h) g[half ^ 1][u][h] = img[(size_t)(kn + u) * W + h];
                    if (RL == 2 && k0 >= 64) wave_rows8<RL, RL - 1>(g[half], k0, acc, nvd, dk);
                    else wave_rows8<RL, 0>(g[half], k0, acc, nvd, dk);
                } else {
                    // RU is an odd number of blocks: this half-trip does not exist; keep the register sets in step
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int h = 0; h < RL; ++h) g[half ^ 1][u][h] = g[half][u][h];
                }
            }
        }
        // end of sweep s: apply the steps, this column's sum of squared steps
        float f = 0.f;
#pragma unroll
        for (int h = 0; h < RL; ++h) {
            v2[h] = v1[h];
            v1[h] = v[h];                                 // v1 = V after sweep s-1, v2 = after s-2
            v[h] += dk[h];
            nvd[h] = dead[h] ? -__builtin_inff() : -v[h];
            f = fmaf(dk[h], dk[h], f);
        }
        const double wsum = (double)wave_sum_f32(f);
        // workgroup partial: slot, arrival count; the last wave to arrive adds the slots in wave order and publishes
        const int slot = s & (WAVE_RING - 1);
        if (lane == 0) {
            part[slot * WAVE_MAX_NW + w] = wsum;
            __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the slot is written before the arrival is counted
            const unsigned prev = __hip_atomic_fetch_add(&arrive[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (prev == (unsigned)NW - 1u) {
                double bs = 0.0;
                for (int i = 0; i < NW; ++i) bs += part[slot * WAVE_MAX_NW + i];
                arrive[slot] = 0u;
                const unsigned long long bits = __builtin_bit_cast(unsigned long long, bs);
                const unsigned long long tag = (unsigned long long)(a.sy.epoch * 1024u + (unsigned)s) << 32;
                unsigned long long* gq = reinterpret_cast<unsigned long long*>(a.sy.sslots) + ((size_t)s * nblocks + blockIdx.x) * 2;
                __hip_atomic_store(gq, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gq + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // lag two: the sum of sweep s-2 (granules requested after sweep s-1) is examined now
        const int c = s - 2;
        if (c >= 1) {
            double tot;
            ok = wave_collect(a.sy, c, nblocks, lane, tot, pf);
            if (!ok) break;
            if (decide(c, tot)) { stopped = true; break; }
        }
        if (s >= 2 && s < a.max_sweeps) wave_collect_issue(a.sy, s - 1, nblocks, lane, pf);
    }
    // result: V after sweep `done` once the loop has decided; the budget ran out with one or two sweeps still undecided
    int have = a.max_sweeps;                              // sweeps the registers hold (v), v1 = have-1, v2 = have-2
    if (stopped) {
        // decided at the end of sweep done + 2
#pragma unroll
        for (int h = 0; h < RL; ++h) v[h] = v2[h];
    } else if (ok && a.max_sweeps >= 1) {
        for (int c = (a.max_sweeps >= 2 ? a.max_sweeps - 1 : 1); c <= a.max_sweeps; ++c) {
            double tot;
            wave_prefetch none;
            none.s = 0;
            ok = wave_collect(a.sy, c, nblocks, lane, tot, none);
            if (!ok) break;
            if (decide(c, tot) && c < have) {             // sweep have-1 was the last one: drop the sweep run ahead
#pragma unroll
                for (int h = 0; h < RL; ++h) v[h] = v1[h];
                break;
            }
        }
    } else if (!ok) {
        // time-out: report the last confirmed sweep's column (two sweeps back at most)
#pragma unroll
        for (int h = 0; h < RL; ++h) v[h] = v2[h];
    }
    if (a.max_sweeps >= 1) {
#pragma unroll
        for (int h = 0; h < RL; ++h) {
            const int row = lane + 64 * h;
            if (valid && row < a.r) a.V[(int64_t)row * a.ldv + col] = v[h];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.max_sweeps >= 1) {
            a.status[NNF_HALS_ST_EPS] = eps;
            a.status[NNF_HALS_ST_CNT] = (double)(done + 1);
            a.status[NNF_HALS_ST_EPS0] = eps0;
        }
        if (!ok) a.status[NNF_HALS_ST_ERR] = 1.0;
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
static int wave_ru(int r) { return (r + 7) & ~7; }
static int wave_rl(int r) { return r <= 64 ? 1 : 2; }
static size_t wave_lds(int r) {
    return (size_t)wave_ru(r) * 64 * wave_rl(r) * 4 + (size_t)WAVE_RING * WAVE_MAX_NW * 8 + WAVE_RING * 4 + 16;
}
static int wave_nw(int64_t ncols) {      // waves per workgroup: ~one workgroup per CU, 2 .. 16 columns each
    int nw = 2;
    while (nw < WAVE_MAX_NW && ncols > (int64_t)256 * nw) nw *= 2;
    return nw;
}

size_t nnf_hals_wave_gram_floats(int r) { return (size_t)wave_ru(r) * 64 * wave_rl(r) + 128; }

// all workgroups of the persistent kernel must be co-resident
bool nnf_hals_wave_fits(nnf_ctx* ctx, int r, int64_t ncols, int max_blocks_cap) {
    if (r < 1 || r > 128 || ncols < 1 || ncols > 8192) return false;
    const int nw = wave_nw(ncols);
    const int64_t need = nnf_cdiv(ncols, nw);
    if (need > 64 * WAVE_PF || need > max_blocks_cap) return false;
    static int cached[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};     // [RL-1][log2 nw]
    int lg = 0;
    while ((1 << lg) < nw) ++lg;
    const int rl = wave_rl(r);
    // (occupancy depends on the LDS size, i.e. on r: query with this r's size, cache the worst case per (RL, nw) conservatively)
    int nb = 0;
    const size_t lds = wave_lds(r);
    hipError_t e;
    if (rl == 1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_hals_wave_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_wave_kernel<1>, 64 * nw, lds);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_hals_wave_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_wave_kernel<2>, 64 * nw, lds);
    }
    (void)cached;
    if (e != hipSuccess || nb < 1) return false;
    const int b = nb >= 3 ? nb - 1 : nb;                  // margin: the occupancy API can over-report by one block per CU
    return need <= (int64_t)b * ctx->num_cus;
}

// Gw: workspace of nnf_hals_wave_gram_floats(r) floats.  Solve mode only (a.mode == 0, a.sweep0 == 0).
int nnf_hals_wave_run(nnf_ctx* ctx, const float* UtU, const float* UtU2, int64_t ldg, float* Gw, unsigned* counter, hals_args a,
                      int* nblocks_out, hipStream_t st) {
    const int ru = wave_ru(a.r), rl = wave_rl(a.r);
    float* dinv = Gw + (size_t)ru * 64 * rl;
    hipLaunchKernelGGL(nnf_hals_prep_wave_kernel, dim3(ru), dim3(64), 0, st, UtU, UtU2, ldg, a.r, rl, Gw, dinv, counter, a.status);
    NNF_CHECK_LAUNCH();
    if (a.max_sweeps == 0) {
        if (a.Vsrc != a.V && hipMemcpy2DAsync(a.V, (size_t)a.ldv * 4, a.Vsrc, (size_t)a.ldvs * 4, (size_t)a.ncols * 4, (size_t)a.r,
                                               hipMemcpyDeviceToDevice, st) != hipSuccess)
            return NNF_ERR_LAUNCH;
        return NNF_OK;
    }
    a.Gp = Gw;
    a.dinv = dinv;
    const int nw = wave_nw(a.ncols);
    const int nblocks = (int)nnf_cdiv(a.ncols, nw);
    *nblocks_out = nblocks;
    nnf_probe(ctx, NNF_PROBE_HALS, 0, st);
    if (rl == 1) hipLaunchKernelGGL((nnf_hals_wave_kernel<1>), dim3(nblocks), dim3(64 * nw), wave_lds(a.r), st, a, ru);
    else hipLaunchKernelGGL((nnf_hals_wave_kernel<2>), dim3(nblocks), dim3(64 * nw), wave_lds(a.r), st, a, ru);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_HALS, 1, st);
    return NNF_OK;
}
