// X H^T with the X tile staged through LDS in whole 256-byte row pieces (nmf.py:408, ntd.py / ntf.py mode products along the
// contiguous axis):   out[rk][i] = sum_j V[rk][j] * X[i][j]
//
// nnf_xht_kernel (k_stream.hip) loads the MFMA B fragments of X straight into registers: a wave instruction then reads 16 rows x
// 64 bytes -- half a cache line per row -- and that access shape itself tops out near 3.5 TB/s (DESIGN_HISTORY.md, segment
// MTTKRP ablation; the guide's "fragment-shaped loads": the texture path is busy twice as long for the same bytes).  Here a
// wave instruction reads 4 rows x 256 contiguous bytes, the wave parks the 64-row x 64-column piece in ITS OWN 16 KB of LDS
// (no workgroup barrier involved: LDS operations of one wave complete in order) and reads the fragments back with ds_read_b128:
//     LDS tile, f32x4 slots:   slot(row, piece p) = 16 * row + (p ^ (row & 15))          row < 16 NT, p < 16 (4 columns each)
// The XOR keeps both sides conflict-free: a store instruction covers 4 rows x 16 consecutive pieces (8 contiguous lanes = 8
// different slots of one 128-byte window); a fragment read has lane (ii = l & 15, g = l >> 4) take piece 4t + g of row 16nt + ii,
// and within each of ds_read_b128's four 16-lane groups ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) the sixteen
// (4t + g) ^ ii are all different.
// The rank-side operand, the chunk loop, the leftover-rank FMAs and the epilogue are those of nnf_xht_kernel.
// For ranks <= 32 (at most two staged rank tiles: 64 KB of X tiles + 16 KB of V images, two workgroups per CU).  Measured (tools/
// probes/xht_probe.py): 250000 x 500 rank 30 (config D's partial product) 148 -> 127 us, 100000 x 2000 rank 32 188 -> 159, rank 16
// 180 -> 153 (5.2 TB/s); with 32-row waves where the pitch leaves shared boundary lines (launch_xht_lds) 127 -> 117.  Ranks 33..64 were built too -- 8 waves sharing one image (160 KB, one workgroup per CU): 304-316 us
// against 228-245; 32-column steps with 128-byte pieces, three workgroups per CU: 234-254 -- and dropped: from three rank tiles
// on the product is bound by the fp32 MFMA rate (6.1 row tiles per SIMD at 100000 rows = 167 us at 100 %, 7 on the busiest), not by
// how X arrives.  They, and unaligned X, stay on nnf_xht_kernel.
#include "k_stream_common.h"

template <int MT, int REM, int NT>
__device__ __forceinline__ void nnf_xht_lds_body(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                 const float* __restrict__ V, int64_t ldv, int r, float* __restrict__ out,
                                                 int64_t ldo, int a_vec_ok, int64_t row0, f32x4 (*ldsA)[(MT + (REM > 0 ? 1 : 0)) * 256],
                                                 f32x4* __restrict__ tile) {
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ii = lane & 15, g = lane >> 4;
    const int64_t i0w = row0 + 16 * NT * w;
    int64_t rows = m - i0w;
    if (rows > 16 * NT) rows = 16 * NT;
    const uint32_t bytes = rows > 0 ? (uint32_t)(((rows - 1) * ldx + n) * 4) : 0u;
    const rsrc_t rs = nnf_make_rsrc(X + (rows > 0 ? i0w : 0) * ldx, bytes);
    const int ldx4 = (int)(ldx * 4);
    const int nchunk = (int)((n + 63) >> 6);
    // loader: instruction i covers rows 4i .. 4i+3 of the wave's piece; lane = (row 4i + lr, piece lp)
    const int lr = lane >> 4, lp = lane & 15;
    const int voff = (int)(((int64_t)lr * ldx + 4 * lp) * 4);
    int wslot[4];   // store slot of instruction i (i & 3 = k), without the 64 i of its rows
#pragma unroll
    for (int k = 0; k < 4; ++k) wslot[k] = 16 * lr + (lp ^ (4 * k + lr));
    int rslot[4];   // fragment slot of k-group t, without the 256 nt of its row tile
#pragma unroll
    for (int t = 0; t < 4; ++t) rslot[t] = 16 * ii + ((4 * t + g) ^ ii);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xr[4 * NT];   // the chunk in flight: instruction i
    f32x4 areg[MTA];
    float ev[REM > 0 ? REM : 1][NT];
#pragma unroll
    for (int rr = 0; rr < (REM > 0 ? REM : 1); ++rr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ev[rr][nt] = 0.f;

    stageA_load<MTA>(V, ldv, r, n, 0, a_vec_ok, areg);
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) xr[i] = nnf_bload4<true>(rs, voff, 4 * i * ldx4);
    stageA_store<MTA>(ldsA[0], areg);
    __syncthreads();

    for (int q = 0; q < nchunk; ++q) {
        const f32x4* img = ldsA[q & 1];
        // ragged column tail: the pieces past column n belong to the next row (or to nobody) -- never into a product
        const int64_t nrem = n - (64 * (int64_t)q + 4 * lp);
        if (nrem < 4) {
#pragma unroll
            for (int i = 0; i < 4 * NT; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c >= nrem) xr[i][c] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) tile[64 * i + wslot[i & 3]] = xr[i];
        if (q + 1 < nchunk) {
#pragma unroll
            for (int i = 0; i < 4 * NT; ++i) xr[i] = nnf_bload4<true>(rs, voff, 4 * i * ldx4 + 256 * (q + 1));
        }
        stageA_load<MTA>(V, ldv, r, n, 64 * (int64_t)(q + 1), a_vec_ok, areg);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = img[(mt * 4 + t) * 64 + lane];
            f32x4 xb[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xb[nt] = tile[256 * nt + rslot[t]];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(af[mt][c], xb[nt][c], acc[mt][nt]);
            if constexpr (REM > 0) {
#pragma unroll
                for (int rr = 0; rr < REM; ++rr) {
                    const f32x4 uv = img[(MT * 4 + t) * 64 + 16 * g + rr];   // V[16MT+rr][64q+16t+4g+c], c = 0..3
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float e = ev[rr][nt];
#pragma unroll
                        for (int c = 0; c < 4; ++c) e = fmaf(uv[c], xb[nt][c], e);
                        ev[rr][nt] = e;
                    }
                }
            }
        }
        stageA_store<MTA>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads();
    }

    // epilogue: tile (mt, nt): out[16mt + 4g + reg][i0w + 16nt + ii]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t i = i0w + 16 * nt + ii;
        if (i < m) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int rk = 16 * mt + 4 * g + reg;
                    if (rk < r) out[(int64_t)rk * ldo + i] = acc[mt][nt][reg];
                }
        }
    }
    if constexpr (REM > 0) {
#pragma unroll
        for (int rr = 0; rr < REM; ++rr)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float x = ev[rr][nt];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                const int64_t i = i0w + 16 * nt + ii;
                const int rk = 16 * MT + rr;
                if (g == 0 && rk < r && i < m) out[(int64_t)rk * ldo + i] = x;
            }
    }
}

// The first n_hi workgroups take NTH row tiles per wave, the others NTH-1 (nnf_xht_kernel's split: one round of resident
// workgroups covers the matrix where it can).
template <int MT, int REM, int NTH>
__global__ __launch_bounds__(256, (NTH == 2 ? 3 : 2)) void nnf_xht_lds_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                             const float* __restrict__ V, int64_t ldv, int r,
                                                             float* __restrict__ out, int64_t ldo, int a_vec_ok, int n_hi) {
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);
    __shared__ f32x4 ldsA[2][MTA * 256];
    __shared__ f32x4 tiles[4][NTH * 256];
    const int b = (int)blockIdx.x;
    f32x4* tile = tiles[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    if (b < n_hi)
        nnf_xht_lds_body<MT, REM, NTH>(X, m, n, ldx, V, ldv, r, out, ldo, a_vec_ok, (int64_t)b * (64 * NTH), ldsA, tile);
    else
        nnf_xht_lds_body<MT, REM, NTH - 1>(X, m, n, ldx, V, ldv, r, out, ldo, a_vec_ok,
                                           (int64_t)n_hi * (64 * NTH) + (int64_t)(b - n_hi) * (64 * (NTH - 1)), ldsA, tile);
}

template <int MT, int REM>
static int launch_xht_lds(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V, int r, int64_t ldv,
                          float* out, int64_t ldo, hipStream_t st) {
    if (64 * ldx * 4 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    const int a_vec_ok = ((((uintptr_t)V) & 15) == 0 && (ldv & 3) == 0) ? 1 : 0;
    const int64_t slots = (int64_t)2 * ctx->num_cus;   // resident workgroups
    const int64_t T = nnf_cdiv(m, 16), waves = 4 * slots;
    int nth = 4;
    int64_t n_hi, grid;
    // A row pitch that is not a whole number of 128-byte lines leaves every 256-byte piece sharing its first and last line with
    // the neighbouring chunks' pieces of the same row: the wave comes back for them one chunk later, after everything the XCD's
    // 64 resident workgroups fetched in between -- 4 MB with 64-row waves, the size of the L2 (250000 x 500: 660 MB fetched for
    // 500 MB, 127 us).  32-row waves (48 KB of LDS: three workgroups per CU) halve that distance: 559 MB, 120 us.  Aligned pitches
    // have no shared lines and keep the 64-row waves (100000 x 2000 rank 32: 157 us against 175).
    static const int pin = [] { const char* e = getenv("NNF_XHT_NT"); return e ? atoi(e) : 0; }();       // measurement knob
    // (rows start on a line every 128 / gcd(pitch mod 128, 128) rows: the narrow form from every fourth row on -- with every
    //  second row aligned, 100000 x 2000, the 64-row waves stay ahead, 159 us against 190)
    int64_t off = (ldx * 4) % 128, gcd = 128;
    while (off) { const int64_t t = gcd % off; gcd = off; off = t; }
    const bool shared_lines = pin ? pin == 2 : (128 / gcd >= 4);
    if (shared_lines) {
        nth = 2;
        n_hi = grid = nnf_cdiv(m, 128);
    } else if (T > 4 * waves) {            // several rounds of 256-row workgroups
        n_hi = grid = nnf_cdiv(m, 256);
    } else if (T > 2 * waves) {     // one round: (4,3) or (3,2) tiles per wave
        nth = T > 3 * waves ? 4 : 3;
        n_hi = nnf_cdiv(T - 4 * (nth - 1) * slots, 4);
        grid = slots;
    } else {                        // small: 128-row workgroups
        nth = 3;
        n_hi = 0;
        grid = nnf_cdiv(m, 128);
    }
    if (n_hi * 64 * nth + (grid - n_hi) * 64 * (nth - 1) < m) return NNF_ERR_UNSUPPORTED;   // (the split covers m by construction)
    nnf_probe(ctx, NNF_PROBE_XHT, 0, st);
    if (nth == 4)
        hipLaunchKernelGGL((nnf_xht_lds_kernel<MT, REM, 4>), dim3((int)grid), dim3(256), 0, st, X, m, n, ldx, V, ldv, r, out, ldo,
                           a_vec_ok, (int)n_hi);
    else if (nth == 2)
        hipLaunchKernelGGL((nnf_xht_lds_kernel<MT, REM, 2>), dim3((int)grid), dim3(256), 0, st, X, m, n, ldx, V, ldv, r, out, ldo,
                           a_vec_ok, (int)n_hi);
    else
        hipLaunchKernelGGL((nnf_xht_lds_kernel<MT, REM, 3>), dim3((int)grid), dim3(256), 0, st, X, m, n, ldx, V, ldv, r, out, ldo,
                           a_vec_ok, (int)n_hi);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_XHT, 1, st);
    return NNF_OK;
}

// MT full rank tiles + REM leftover ranks (0, 2, 4) with MT + (REM > 0) <= 2; X 16-byte aligned, ldx % 4 == 0 (the caller checks)
int nnf_xht_lds_launch(nnf_ctx* ctx, int MT, int REM, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V, int r,
                       int64_t ldv, float* out, int64_t ldo, hipStream_t st) {
#define XL(MT_, REM_) \
    if (MT == MT_ && REM == REM_) return launch_xht_lds<MT_, REM_>(ctx, X, m, n, ldx, V, r, ldv, out, ldo, st);
    XL(1, 0) XL(2, 0) XL(1, 2) XL(1, 4)
#undef XL
    return NNF_ERR_UNSUPPORTED;
}
