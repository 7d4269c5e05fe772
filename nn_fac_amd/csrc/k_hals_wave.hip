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
// its step): FOUR vector instructions per row, three of them dependent, whatever the rank, and no scalar instruction at all
// (the sweep is unrolled: the lane numbers are immediates).  Column k of G' comes from an LDS image shared by the
// workgroup's waves.  There are n waves instead of n/16: 2000 columns = two waves on every SIMD of 250 CUs.
//
// Rounding: the residual is formed from scratch (b' - G'v, summed in row order like the reference's dot product) when the
// kernel starts and every 8 sweeps, and pushed forward in between; each push rounds once relative to the residual itself
// (which shrinks as the solve converges), not to the dot product, so the carried residual is as accurate as a freshly
// evaluated fp32 one.  Results differ from the other layouts in the last bits, like those differ from
// each other.
//
// Stopping rule (nnls.py:156) on the device.  A sweep takes ~0.5 us here -- a third of the round trip through the memory
// side of the chip that a grid-wide exchange costs (the granules cross XCDs: ~1.6 us measured) -- so the compute waves never
// wait for it:
//   * a compute wave ends a sweep with its fp32 sum of squared steps (DPP) as a tagged fp64 in an LDS slot and a snapshot of
//     its column in a global scratch ring (256 coalesced bytes; 16 sweeps deep), then goes on to the next sweep;
//   * FOUR more waves per workgroup, the COMMUNICATION waves, do the exchange; wave u serves the sweeps c = u + 1 (mod 4):
//     it adds the slots of sweep c in wave order once they are all there, publishes the block sum as two tagged granules
//     (k_hals_common.h), then reads the granules of ALL blocks for sweep c - 4 (published a few sweeps ago: normally one
//     round trip), adds them in block order -- the same double in every workgroup -- and leaves the total and the verdict
//     `total >= delta * eps0` in an LDS ring.  Each of the four takes a round trip per sweep it serves; together they keep up
//     with the sweeps.  (One wave with four requests in flight would do as well -- but in-flight registers across a loop
//     with re-read branches are exactly what hipcc's wait-count insertion turns into "wait for everything".)
//   * a compute wave reads the verdicts IN SWEEP ORDER, one LDS word per sweep, before it starts a sweep: it may run ahead of
//     them by at most the depth of its snapshot ring, and at the first "stop" verdict -- sweep c -- it reloads the snapshot of
//     sweep c and leaves.  Sweeps run ahead of a stop are discarded.
// Nothing in a sweep depends on the exchange; every spin is bounded.  Control words travel through LDS as relaxed
// workgroup-scope atomics (`volatile` makes hipcc drain every outstanding memory operation in front of each access).
// Rows with a zero Gram diagonal (nnls.py:160) and the padding rows hold residual 0 and "-v" = -inf: their step is 0.
#include "k_hals_common.h"

constexpr int WAVE_COMM = 4;         // communication waves per workgroup
constexpr int WAVE_MAX_NW = 16 - WAVE_COMM;   // compute waves (= columns) per workgroup (1024 threads in all)
constexpr int WAVE_SNAP = 16;        // snapshot / slot / verdict rings (sweeps a compute wave may run ahead of the verdicts, + 1)
constexpr int WAVE_NP = 6;           // granule pairs per lane of a communication wave: nblocks <= 384
#ifndef WAVE_REFRESH_V
#define WAVE_REFRESH_V 8             // (A/B builds: 4, 2 -- tools/probes/wave_refresh_probe.sh; accuracy / time table in DESIGN.md section 4)
#endif
constexpr int WAVE_REFRESH = WAVE_REFRESH_V;   // the residual is re-formed from scratch (b' - G'v) every so many sweeps (a power of two)
constexpr unsigned WAVE_ERR = 0xffffffffu;

#ifndef WAVE_DBG
#define WAVE_DBG 0      // timing-only ablations (tools/probes/vside_probe.py): 1 = verdicts without totals (run to the budget), 4 = no row updates
#endif
NNF_BUILD_FLAGS(k_hals_wave, "WAVE_DBG=" NNF_STR(WAVE_DBG) " WAVE_REFRESH_V=" NNF_STR(WAVE_REFRESH_V))

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
// UtU[i][k] as stored -- NOT UtU[k][i]: the reference reads rows of whatever it is handed (nnls.py:167) and its own tests pass
// a random, non-symmetric "Gram" (tests/nnls_tests.py:44-45).  Then 1/diag per row, counter, status.
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
            if (d != 0.f) val = gram(i, k) * (float)(1.0 / (double)d);
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

// LDS control block behind the image.  Every word that crosses waves carries the sweep number it belongs to.
struct wave_ctl {
    unsigned long long slot[WAVE_SNAP][WAVE_MAX_NW][2];   // tagged halves {sweep, lo32} {sweep, hi32} of a compute wave's fp64 sum
    unsigned long long tot[WAVE_SNAP][2];                 // the same for the global sum of a sweep
    unsigned long long ver[WAVE_SNAP];                    // verdict {sweep : 32 | 1 = stop, 2 = time-out : 32}, written after tot
    unsigned long long eps0[2];                           // tagged halves (tag 1) of the global sum of sweep 1
    unsigned long long fin;                               // {1 : 32 | stopping sweep : 32} once a compute wave has found it
};

// Words of the control block are exchanged between waves of the workgroup with RELAXED workgroup-scope atomics: plain LDS
// reads / writes that the compiler neither caches in registers nor hoists out of a polling loop.  (`volatile` accesses made
// hipcc drain EVERY outstanding memory operation first -- s_waitcnt vmcnt(0) in front of each poll, i.e. the round trip of the
// snapshot store of the sweep before, 0.65 us per sweep.)
__device__ __forceinline__ unsigned long long wave_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wave_st(unsigned long long* p, unsigned long long x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wave_st_tagged(unsigned long long* p2, unsigned tagv, double x) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, x), tag = (unsigned long long)tagv << 32;
    wave_st(p2, tag | (bits & 0xffffffffull));
    wave_st(p2 + 1, tag | (bits >> 32));
}
// false while either half does not carry `tagv` yet
__device__ __forceinline__ bool wave_ld_tagged(const unsigned long long* p2, unsigned tagv, double& x) {
    const unsigned long long h0 = wave_ld(p2), h1 = wave_ld(p2 + 1);
    x = __builtin_bit_cast(double, (h1 << 32) | (h0 & 0xffffffffull));
    return (unsigned)(h0 >> 32) == tagv && (unsigned)(h1 >> 32) == tagv;
}

// row update K (compile-time: every operand position is an immediate) of the CPW columns a wave holds -- the columns'
// chains are independent, so a wave with two columns fills the latency of one chain with the other
template <int RL, int CPW, int K>
__device__ __forceinline__ void wave_row(const float* img, float (&acc)[CPW][RL], const float (&nvd)[CPW][RL], float (&dk)[CPW][RL]) {
    constexpr int H = K >> 6, KL = K & 63, W = 64 * RL;
    float g[RL];
#pragma unroll
    for (int h = 0; h < RL; ++h) g[h] = img[K * W + h];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        float d;   // max(acc, -v) in one instruction (fmaxf adds a canonicalising max); lane KL's value is the step of row K
        asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(acc[c][H]), "v"(nvd[c][H]));
        const float sd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), KL));
#pragma unroll
        for (int h = 0; h < RL; ++h) acc[c][h] = fmaf(g[h], -sd, acc[c][h]);
        // lane KL keeps its step (v_writelane: SGPR data, immediate lane select; this clang has no builtin for it)
        asm("v_writelane_b32 %0, %1, %2" : "+v"(dk[c][H]) : "s"(sd), "n"(KL));
    }
}
template <int RL, int CPW, int K, int RU>
struct wave_rows {
    static __device__ __forceinline__ void run(const float* img, float (&acc)[CPW][RL], const float (&nvd)[CPW][RL], float (&dk)[CPW][RL]) {
        wave_row<RL, CPW, K>(img, acc, nvd, dk);
        if constexpr (K + 1 < RU) wave_rows<RL, CPW, K + 1, RU>::run(img, acc, nvd, dk);
    }
};

// fixed-order fp64 sum of lanes 0 .. 15 of `v` (other lanes must hold 0), wave-uniform
__device__ __forceinline__ double wave_sum16_f64(double v) {
    v = nnf_dpp_add_f64<0xB1, 0xf>(v);
    v = nnf_dpp_add_f64<0x4E, 0xf>(v);
    v = nnf_dpp_add_f64<0x141, 0xf>(v);
    v = nnf_dpp_add_f64<0x140, 0xf>(v);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 0);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 0);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}

// global sum of sweep s: the granules of all blocks, strided over the lanes (late ones are re-read, bounded), added in block
// order per lane, DPP wave sum -- the same double in every workgroup.  false = time-out or `fin` set (wave-uniform).
__device__ __forceinline__ bool wave_total(const hals_sync& sy, const wave_ctl* ctl, int s, int nblocks, int lane, double& total) {
    const unsigned tag = sy.epoch * 1024u + (unsigned)s;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(sy.sslots) + (size_t)s * nblocks * 2;
    unsigned long long g0[WAVE_NP], g1[WAVE_NP];
    // every request of the wave goes out before the first answer is looked at: no load under a branch (lanes past the last
    // block re-read block 0 and ignore it) -- one round trip for the lot, not one per pair
#pragma unroll
    for (int i = 0; i < WAVE_NP; ++i) {
        const int b = lane + 64 * i < nblocks ? lane + 64 * i : 0;
        g0[i] = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g1[i] = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool fresh = true;
#pragma unroll
    for (int i = 0; i < WAVE_NP; ++i) fresh = fresh && (unsigned)(g0[i] >> 32) == tag && (unsigned)(g1[i] >> 32) == tag;
    bool late = false;
    if (__ballot(!fresh) != 0ull) {                       // somebody's granule is not there yet: re-read (rare, bounded)
#pragma unroll 1
        for (int i = 0; i < WAVE_NP; ++i) {
            const int b = lane + 64 * i < nblocks ? lane + 64 * i : 0;
            unsigned long long a0 = g0[0], a1 = g1[0];
#pragma unroll
            for (int u = 0; u < WAVE_NP; ++u)
                if (u == i) { a0 = g0[u]; a1 = g1[u]; }
            unsigned spins = 0;
            while (!((unsigned)(a0 >> 32) == tag && (unsigned)(a1 >> 32) == tag)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > HALS_SPIN_LIMIT || ((spins & 63u) == 0u && wave_ld(&ctl->fin) != 0ull)) { late = true; break; }
                a0 = __hip_atomic_load(base + 2 * (size_t)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a1 = __hip_atomic_load(base + 2 * (size_t)b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < WAVE_NP; ++u)
                if (u == i) { g0[u] = a0; g1[u] = a1; }
        }
    }
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < WAVE_NP; ++i)
        v += (lane + 64 * i < nblocks) ? __builtin_bit_cast(double, (g1[i] << 32) | (g0[i] & 0xffffffffull)) : 0.0;
    total = nnf_wave_sum_f64(v);
    return __ballot(late) == 0ull;
}

// ---- a communication wave: serves the sweeps c = u + 1, u + 1 + WAVE_COMM, ... ---------------------------------------------
// block sum of sweep c: waits (bounded) until every compute wave has left its tagged sum; false: time-out or the solve is over
__device__ __forceinline__ bool wave_block_sum(wave_ctl* ctl, int c, int NW, int lane, double& bs) {
    const int sl = c & (WAVE_SNAP - 1);
    unsigned spins = 0;
    double v = 0.0;
    for (;;) {
        double x = 0.0;
        const bool have = lane >= NW || wave_ld_tagged(&ctl->slot[sl][lane < NW ? lane : 0][0], (unsigned)c, x);
        if (__ballot(!have) == 0ull) {
            v = lane < NW ? x : 0.0;
            break;
        }
        if (wave_ld(&ctl->fin) != 0ull) return false;     // the compute waves have left: the slots of this sweep will never come
        __builtin_amdgcn_s_sleep(2);
        if (++spins > HALS_SPIN_LIMIT) return false;
    }
    bs = wave_sum16_f64(v);
    return true;
}
__device__ __forceinline__ void wave_publish(const hals_sync& sy, int s, int nblocks, int lane, double mine) {
    if (lane == 0) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, mine);
        const unsigned long long tag = (unsigned long long)(sy.epoch * 1024u + (unsigned)s) << 32;
        unsigned long long* g = reinterpret_cast<unsigned long long*>(sy.sslots) + ((size_t)s * nblocks + blockIdx.x) * 2;
        __hip_atomic_store(g, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// total + verdict of sweep cd into the rings (nnls.py:156: the loop goes on while eps >= delta * eps0).  false: give up.
__device__ __forceinline__ bool wave_judge(const hals_args& a, wave_ctl* ctl, int cd, int nblocks, int lane) {
    double tot = 0.0;
    unsigned flag = 0u;
    if (!(WAVE_DBG & 1)) {
        if (!wave_total(a.sy, ctl, cd, nblocks, lane, tot)) {
            if (wave_ld(&ctl->fin) != 0ull) return false;
            flag = 2u;                                    // time-out
        } else {
            double e0 = tot;
            if (cd == 1) {
                if (lane == 0) wave_st_tagged(&ctl->eps0[0], 1u, tot);
            } else {                                      // eps0 comes from the wave that serves sweep 1 (bounded wait)
                unsigned spins = 0;
                while (!wave_ld_tagged(&ctl->eps0[0], 1u, e0)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > HALS_SPIN_LIMIT || wave_ld(&ctl->fin) != 0ull) return false;
                }
            }
            flag = (tot >= a.delta * e0) ? 0u : 1u;
        }
    }
    if (lane == 0) {
        const int sl = cd & (WAVE_SNAP - 1);
        wave_st_tagged(&ctl->tot[sl][0], (unsigned)cd, tot);
        wave_st(&ctl->ver[sl], ((unsigned long long)(unsigned)cd << 32) | (unsigned long long)flag);   // after tot: LDS keeps a wave's order
    }
    return flag == 0u;
}
__device__ __forceinline__ void wave_comm_loop(const hals_args& a, wave_ctl* ctl, int NW, int u, int lane, int nblocks) {
    const int S = a.max_sweeps;
    int c = u + 1;
    for (; c <= S; c += WAVE_COMM) {
        double bs;
        if (!wave_block_sum(ctl, c, NW, lane, bs)) return;
        wave_publish(a.sy, c, nblocks, lane, bs);
        const int cd = c - WAVE_COMM;                     // published by everybody a few sweeps ago
        if (cd >= 1 && !wave_judge(a, ctl, cd, nblocks, lane)) return;
    }
    // the last sweep this wave served has not been judged yet
    const int cd = c - WAVE_COMM;
    if (cd >= 1 && cd <= S) (void)wave_judge(a, ctl, cd, nblocks, lane);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// a.Gp = UtU (r x r, row stride ldg), a.Gs = the second Gram of a Hadamard pair or NULL: every workgroup builds its own LDS
// image of G' by columns from them (a few Gram entries per thread; a preparation launch of its own was 5-6 us in front of a
// 17 us solve at the NTF shapes).  snap: global scratch [ncols][WAVE_SNAP][64 RL] floats.
// CPW: columns per compute wave (2 when one column per wave would need more workgroups than the chip holds at once).
template <int RL, int RU, int CPW>
__global__ __launch_bounds__(1024) void nnf_hals_wave_kernel(hals_args a, int64_t ldg, float* __restrict__ snap) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    constexpr int W = 64 * RL;
    wave_ctl* ctl = reinterpret_cast<wave_ctl*>(wlds + (size_t)RU * W);
    const int NW = (blockDim.x >> 6) - WAVE_COMM;         // compute waves; waves NW .. NW+3 are the communication waves
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblocks = gridDim.x;
    auto gram = [&](int i, int k) -> float {              // UtU[i][k] as stored (rows: see the preparation kernel's note)
        const float g = a.Gp[(int64_t)i * ldg + k];
        return a.Gs ? g * a.Gs[(int64_t)i * ldg + k] : g;
    };
    for (int e = threadIdx.x; e < RU * W; e += blockDim.x) {
        const int k = e / W, c = e - k * W;
        const int i = (RL == 2) ? ((c >> 1) + 64 * (c & 1)) : c;
        float val = 0.f;
        if (k < a.r && i < a.r) {
            const float d = gram(i, i);
            if (d != 0.f) val = gram(i, k) * (float)(1.0 / (double)d);
        }
        wlds[e] = val;
    }
    for (int e = threadIdx.x; e < (int)(sizeof(wave_ctl) / 8); e += blockDim.x)
        reinterpret_cast<unsigned long long*>(ctl)[e] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {            // (rewritten by this same thread at the end)
        a.status[NNF_HALS_ST_EPS] = 1.0;
        a.status[NNF_HALS_ST_CNT] = 1.0;
        a.status[NNF_HALS_ST_EPS0] = 0.0;
        a.status[NNF_HALS_ST_ERR] = 0.0;
    }
    __syncthreads();                                      // the only barrier: image and control block are in LDS
    if (w >= NW) {
        wave_comm_loop(a, ctl, NW, w - NW, lane, nblocks);
        return;
    }
    const int64_t col0 = ((int64_t)blockIdx.x * NW + w) * CPW;
    // this lane's rows: lane (and lane + 64) of each column; start values from a.Vsrc (== a.V for an in-place solve)
    float v[CPW][RL], acc[CPW][RL], nvd[CPW][RL], dk[CPW][RL], bs[CPW][RL];
    bool dead[CPW][RL];
    float* mysnap[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int64_t col = col0 + c;
        const bool valid = col < a.ncols;
#pragma unroll
        for (int h = 0; h < RL; ++h) {
            const int row = lane + 64 * h;
            const bool in = valid && row < a.r;
            float di = 0.f;                               // 0: zero diagonal or padding row
            if (row < a.r) {
                const float d = gram(row, row);
                di = (d != 0.f) ? (float)(1.0 / (double)d) : 0.f;
            }
            dead[c][h] = !(in && di != 0.f);
            v[c][h] = in ? a.Vsrc[(int64_t)row * a.ldvs + col] : 0.f;
            const float bm = in ? a.UtM[(int64_t)row * a.ldm + col] : 0.f;
            bs[c][h] = dead[c][h] ? 0.f : (bm - a.sp) * di;
            nvd[c][h] = dead[c][h] ? -__builtin_inff() : -v[c][h];
            acc[c][h] = 0.f;
        }
        mysnap[c] = snap + ((size_t)(valid ? col : 0) * WAVE_SNAP) * W + (RL == 2 ? 2 * lane : lane);
    }
    const bool valid0 = col0 < a.ncols;                   // (a wave whose first column is past the end has nothing to sweep)
    const float* img = wlds + (RL == 2 ? 2 * lane : lane);
    // residual from scratch: acc = b' - G' v   (columns of G' one by one in row order -- the order of the reference's dot
    // product --, v[j] through an SGPR).  At the start, and again every WAVE_REFRESH sweeps: the pushes in between carry it
    // forward exactly up to one rounding each, and the refresh keeps those roundings from adding up over a long solve.
    auto refresh = [&]() {
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int h = 0; h < RL; ++h) acc[c][h] = bs[c][h];
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            float g[RL];
#pragma unroll
            for (int h = 0; h < RL; ++h) g[h] = img[j * W + h];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[c][j >> 6]), j & 63));
#pragma unroll
                for (int h = 0; h < RL; ++h) acc[c][h] = fmaf(g[h], -vj, acc[c][h]);
            }
        }
    };
    if (valid0) refresh();
    const int S = a.max_sweeps;
    int judged = 0;                                       // verdicts read so far: sweeps 1 .. judged go on
    unsigned stop = 0u;                                   // the stopping sweep once its verdict has been read; WAVE_ERR
    // Verdicts are read IN SWEEP ORDER.  The sweep loop looks at ONE ring word per sweep (verdicts arrive at the rate of the
    // sweeps, a few sweeps late: that keeps pace) and only when it would otherwise overrun the ring does it wait.
    auto take = [&](unsigned long long x) -> bool {       // x: the ring word of sweep judged + 1; true = it was that sweep's verdict
        if ((int)(unsigned)(x >> 32) != judged + 1) return false;
        ++judged;
        const unsigned fl = (unsigned)x;
        if (fl != 0u) stop = (fl == 1u) ? (unsigned)judged : WAVE_ERR;
        return true;
    };
    auto wait_verdicts = [&](int need) {                  // rare path: until sweep `need` is judged (bounded)
        unsigned spins = 0;
        while (stop == 0u && judged < need) {
            if (take(wave_ld(&ctl->ver[(judged + 1) & (WAVE_SNAP - 1)]))) { spins = 0; continue; }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > HALS_SPIN_LIMIT) stop = WAVE_ERR;
        }
    };
    int ran = 0;                                          // sweeps this wave has run
    for (int s = 1; s <= S; ++s) {
        // the slot / snapshot of sweep s reuses those of sweep s - WAVE_SNAP: that sweep must have been judged (and go on)
        (void)take(wave_ld(&ctl->ver[(judged + 1) & (WAVE_SNAP - 1)]));
        if (__builtin_expect(judged < s - WAVE_SNAP + 1, 0)) wait_verdicts(s - WAVE_SNAP + 1);
        if (stop != 0u) break;
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int h = 0; h < RL; ++h) dk[c][h] = 0.f;
        if (valid0 && s > 1 && ((s - 1) & (WAVE_REFRESH - 1)) == 0) refresh();
        if (valid0 && !(WAVE_DBG & 4)) wave_rows<RL, CPW, 0, RU>::run(img, acc, nvd, dk);
        float f = 0.f;
        const int sl = s & (WAVE_SNAP - 1);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
#pragma unroll
            for (int h = 0; h < RL; ++h) {
                v[c][h] += dk[c][h];
                nvd[c][h] = dead[c][h] ? -__builtin_inff() : -v[c][h];
                f = fmaf(dk[c][h], dk[c][h], f);
            }
            if (col0 + c < a.ncols) {                     // V after sweep s: coalesced, fire and forget (read back by this wave only)
#pragma unroll
                for (int h = 0; h < RL; ++h) mysnap[c][(size_t)sl * W + h] = v[c][h];
            }
        }
        const double wsum = (double)wave_sum_f32(f);
        if (lane == 0) wave_st_tagged(&ctl->slot[sl][w][0], (unsigned)s, wsum);
        ran = s;
    }
    if (stop == 0u && S >= 1) wait_verdicts(S);           // the budget ran out: every sweep must be judged
    const unsigned last = stop != 0u ? stop : (unsigned)S;    // the sweep whose column is the result
    if (w == 0 && lane == 0) wave_st(&ctl->fin, (1ull << 32) | (unsigned long long)last);     // (communication waves may leave)
    // A time-out seen by ANY compute wave is left in the call's workspace word, tagged with the call's epoch: workgroup 0 may
    // still find every granule later (it needs the same late granule, so it ends after this store) and folds the word in.
    const unsigned err_tag = 0xE0000000u | a.sy.epoch;
    if (last == WAVE_ERR && lane == 0) __hip_atomic_store(a.sy.counter, err_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (S >= 1) {
        if (last != WAVE_ERR && (int)last != ran) __builtin_amdgcn_s_waitcnt(0);   // (own snapshot stores have landed before they are read back)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int64_t col = col0 + c;
            if (col < a.ncols) {
                if (last != WAVE_ERR && (int)last != ran) {   // sweeps run ahead of the stop are dropped
                    const int sl = (int)last & (WAVE_SNAP - 1);
#pragma unroll
                    for (int h = 0; h < RL; ++h) v[c][h] = __builtin_nontemporal_load(&mysnap[c][(size_t)sl * W + h]);
                }
#pragma unroll
                for (int h = 0; h < RL; ++h) {
                    const int row = lane + 64 * h;
                    if (row < a.r) a.V[(int64_t)row * a.ldv + col] = v[c][h];
                }
            }
        }
    }
    if (blockIdx.x == 0 && w == 0 && lane == 0) {
        if (S >= 1 && last != WAVE_ERR) {
            double eps = 1.0, eps0 = 0.0;
            (void)wave_ld_tagged(&ctl->tot[(int)last & (WAVE_SNAP - 1)][0], last, eps);
            (void)wave_ld_tagged(&ctl->eps0[0], 1u, eps0);
            if (last == 1u) eps0 = eps;
            if (WAVE_DBG & 1) { eps = 1.0; eps0 = 0.0; }
            a.status[NNF_HALS_ST_EPS] = eps;
            a.status[NNF_HALS_ST_CNT] = (double)(last + 1u);
            a.status[NNF_HALS_ST_EPS0] = eps0;
        }
        if (last == WAVE_ERR || __hip_atomic_load(a.sy.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == err_tag)
            a.status[NNF_HALS_ST_ERR] = 1.0;
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
static int wave_ru(int r) { return (r + 7) & ~7; }
static int wave_rl(int r) { return r <= 64 ? 1 : 2; }
static size_t wave_lds(int r) { return (size_t)wave_ru(r) * 64 * wave_rl(r) * 4 + sizeof(wave_ctl) + 16; }
static int wave_nw(int64_t ncols) {      // compute waves per workgroup: about one workgroup per CU
    int nw = (int)nnf_cdiv(ncols, 256);
    if (nw < 1) nw = 1;
    if (nw > WAVE_MAX_NW) nw = WAVE_MAX_NW;
    return nw;
}

size_t nnf_hals_wave_gram_floats(int r) { return (size_t)wave_ru(r) * 64 * wave_rl(r) + 128; }
size_t nnf_hals_wave_snap_floats(int r, int64_t ncols) { return (size_t)ncols * WAVE_SNAP * 64 * wave_rl(r); }

template <int RL, int RU, int CPW>
static int wave_launch(const hals_args& a, int64_t ldg, float* snap, int nblocks, int nw, size_t lds, hipStream_t st, int* occupancy) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_hals_wave_kernel<RL, RU, CPW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    if (occupancy) {
        static int cached[WAVE_MAX_NW + 1];               // per (RL, RU, CPW) instance and nw: resident workgroups per CU + 1 (0 = not asked yet)
        if (cached[nw] == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nnf_hals_wave_kernel<RL, RU, CPW>, 64 * (nw + WAVE_COMM), lds) != hipSuccess)
                nb = 0;
            cached[nw] = nb + 1;
        }
        *occupancy = cached[nw] - 1;
        return NNF_OK;
    }
    hipLaunchKernelGGL((nnf_hals_wave_kernel<RL, RU, CPW>), dim3(nblocks), dim3(64 * (nw + WAVE_COMM)), lds, st, a, ldg, snap);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}
static int wave_dispatch(int r, int cpw, const hals_args& a, int64_t ldg, float* snap, int nblocks, int nw, hipStream_t st, int* occupancy) {
    const size_t lds = wave_lds(r);
    switch (wave_ru(r)) {
#define WAVE_CASE(N)                                                                                               \
    case N:                                                                                                        \
        return cpw == 1 ? wave_launch<(N <= 64 ? 1 : 2), N, 1>(a, ldg, snap, nblocks, nw, lds, st, occupancy)      \
                        : wave_launch<(N <= 64 ? 1 : 2), N, 2>(a, ldg, snap, nblocks, nw, lds, st, occupancy);
        WAVE_CASE(8) WAVE_CASE(16) WAVE_CASE(24) WAVE_CASE(32) WAVE_CASE(40) WAVE_CASE(48) WAVE_CASE(56) WAVE_CASE(64)
        WAVE_CASE(72) WAVE_CASE(80) WAVE_CASE(88) WAVE_CASE(96) WAVE_CASE(104) WAVE_CASE(112) WAVE_CASE(120) WAVE_CASE(128)
#undef WAVE_CASE
        default: return NNF_ERR_UNSUPPORTED;
    }
}

// Columns per compute wave with which every workgroup of the persistent kernel is co-resident (1, or 2 when one column per
// wave would take more workgroups than the chip holds at once: 4000 columns at rank 100); 0: this layout does not fit.
static int wave_plan(nnf_ctx* ctx, int r, int64_t ncols, int max_blocks_cap) {
    if (r < 1 || r > 128 || ncols < 1) return 0;
    const int nw = wave_nw(ncols);
    const char* pin = getenv("NNF_WAVE_CPW");             // measurement knob (tools/probes/vside_probe.py): start at 2 columns per wave
    for (int cpw = (pin && pin[0] == '2') ? 2 : 1; cpw <= 2; ++cpw) {
        const int64_t need = nnf_cdiv(ncols, (int64_t)nw * cpw);
        if (need > 64 * WAVE_NP || need > max_blocks_cap) continue;
        int nb = 0;
        hals_args dummy{};
        const int rc = wave_dispatch(r, cpw, dummy, 0, nullptr, 0, nw, nullptr, &nb);
        if (getenv("NNF_HALS_DEBUG"))
            fprintf(stderr, "[nnf hals wave] r=%d ncols=%lld nw=%d cpw=%d need=%lld occupancy=%d rc=%d lds=%zu\n", r, (long long)ncols,
                    nw, cpw, (long long)need, nb, rc, wave_lds(r));
        if (rc != NNF_OK || nb < 1) return 0;
        const int b = nb >= 3 ? nb - 1 : nb;              // margin: the occupancy API can over-report by one block per CU
        if (need <= (int64_t)b * ctx->num_cus) return cpw;
    }
    return 0;
}
bool nnf_hals_wave_fits(nnf_ctx* ctx, int r, int64_t ncols, int max_blocks_cap) { return wave_plan(ctx, r, ncols, max_blocks_cap) > 0; }

// Gw: workspace of nnf_hals_wave_gram_floats(r) floats, snap: nnf_hals_wave_snap_floats(r, ncols) floats.
// Solve mode only (a.mode == 0, a.sweep0 == 0).
int nnf_hals_wave_run(nnf_ctx* ctx, const float* UtU, const float* UtU2, int64_t ldg, float* Gw, float* snap, unsigned* counter,
                      hals_args a, int* nblocks_out, hipStream_t st) {
    if (a.max_sweeps == 0) {
        // nothing to sweep: only the status defaults (the preparation kernel writes them) and V_out := V_in
        const int ru = wave_ru(a.r), rl = wave_rl(a.r);
        hipLaunchKernelGGL(nnf_hals_prep_wave_kernel, dim3(ru), dim3(64), 0, st, UtU, UtU2, ldg, a.r, rl, Gw, Gw + (size_t)ru * 64 * rl,
                           counter, a.status);
        NNF_CHECK_LAUNCH();
        if (a.Vsrc != a.V && hipMemcpy2DAsync(a.V, (size_t)a.ldv * 4, a.Vsrc, (size_t)a.ldvs * 4, (size_t)a.ncols * 4, (size_t)a.r,
                                               hipMemcpyDeviceToDevice, st) != hipSuccess)
            return NNF_ERR_LAUNCH;
        return NNF_OK;
    }
    a.Gp = UtU;              // the kernel builds its image from the Gram(s) itself
    a.Gs = UtU2;
    a.dinv = nullptr;
    const int cpw = wave_plan(ctx, a.r, a.ncols, NNF_HALS_MAX_BLOCKS);
    if (cpw < 1) return NNF_ERR_UNSUPPORTED;
    const int nw = wave_nw(a.ncols);
    const int nblocks = (int)nnf_cdiv(a.ncols, (int64_t)nw * cpw);
    *nblocks_out = nblocks;
    nnf_probe(ctx, NNF_PROBE_HALS, 0, st);
    const int rc = wave_dispatch(a.r, cpw, a, ldg, snap, nblocks, nw, st, nullptr);
    nnf_probe(ctx, NNF_PROBE_HALS, 1, st);
    return rc;
}
