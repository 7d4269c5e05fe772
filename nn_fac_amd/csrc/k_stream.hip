// Streaming fp32-MFMA contractions over the data matrix X (gfx950 / CDNA4, wave64).
//
//   xty : out[r x n] = Ut[r x m] * X[m x n]          ("W^T X",  nmf.py:433)   split over m, slab reduce
//   xht : out[r x m] = V [r x n] * X[m x n]^T        ("X H^T",  nmf.py:408)
//   gram: G  [r x r] = A [r x K] * A^T               (nmf.py:407,432; ntf.py:442-445)
//   cost: sum f(X, Ut^T V)                           (nmf.py:452,455; beta_divergence.py:45-52) product never materialised
//
// Common design
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/issue/SIMD; MI355X_MICROARCH "Matrix cores").  The 16-granular M tile
//     keeps rank padding small (r=50 -> 64).  Operand lane maps (cdna_hip_programming.md s.3):
//       A[row = l&15][k = l>>4],  B[k = l>>4][col = l&15],  D[row = 4*(l>>4)+reg][col = l&15].
//   * X is read ONCE per kernel, straight from HBM into VGPRs with 16-byte buffer loads whose four components feed
//     four different MFMAs, so no LDS round trip for the streamed operand: for xty/frob a wave instruction covers
//     4 rows x 256 contiguous bytes; the column a lane holds in component c is 4*(l&15)+c, i.e. the four N tiles of a
//     wave are column-interleaved (a pure relabelling, undone in the epilogue's float4 stores).
//   * The small operand (Ut / V tile) is staged once per workgroup into LDS in *fragment order*
//     ([tile][k-group][lane] float4) so every fragment read is one conflict-free linear ds_read_b128.
//   * Hardware bounds checking of the buffer descriptor (num_records) zero-fills rows past the end of a workgroup's
//     row range; ragged k tails are masked with selects (never 0*garbage).
//   * Register double buffering: the loads of the next 64-deep chunk are issued right after the MFMAs that free the
//     registers, so ~16 KB per wave stay in flight (hipcc's in-order vmcnt bookkeeping keeps them counted).
#include "k_stream_common.h"
#include <type_traits>
#ifndef XHT_ABL
#define XHT_ABL 0   // timing-only ablations of nnf_xht_kernel (tools/xht_ablate.sh); 0 = the product
#endif
#ifndef XTY_BIG_WG
#define XTY_BIG_WG 2   // resident workgroups per CU the W^T X kernel of five or six rank tiles is compiled for (A/B: tools/abl_build.sh)
#endif
// W^T X, workgroups per CU by rank tiles: up to four tiles three (<= 168 registers); five and six tiles (ranks 65 ... 98) TWO --
// the kernel fits 256 registers there without a spill where the compiler took up to 348 for one wave per SIMD.  Six tiles + four
// leftover ranks (rank 100) and seven / eight tiles stay at one: at 256 registers rank 100 spills 4 and -- worse -- waits for a
// staging load inside the chunk loop (a full drain of the X prefetch ring per trip, tools/check_loop_drains.py), for 6.54 -> 6.34 ms
// at 10^6 x 4000 (tools/probes/xty_occ_probe.py): not taken.
template <int MT, int REM>
constexpr int nnf_xty_wg_per_cu() { return MT + (REM > 0) <= 4 ? 3 : ((MT <= 6 && !(MT == 6 && REM == 4)) ? XTY_BIG_WG : 1); }
NNF_BUILD_FLAGS(k_stream, "XHT_ABL=" NNF_STR(XHT_ABL) " XTY_BIG_WG=" NNF_STR(XTY_BIG_WG))

// =========================================================================================================
// xty: slab[ks][rk][j] = sum_{i in split ks} Ut[rk][i] * X[i][j]
//   grid: 8*ceil(nsplit/8)*ncb workgroups of 256 threads; workgroup = (row split ks, 256-column block cb);
//   wave w owns columns cb*256 + 64w .. +63 (lane: 4*(l&15)+c), all MT row tiles; k runs over the split's rows.
// =========================================================================================================
// REM > 0: rank = 16*MT + (1..REM) -- the MT full 16-row tiles run on MFMA, the REM leftover rows on the VALU pipe, which
// is otherwise idle here (fp32 MFMA and fp32 VALU have the same peak on gfx950, so padding r=50 to 64 would burn 22 % of
// the MFMA time on zeros).  The leftover rows' operand is the (MT+1)-th tile of the same LDS image, read as a broadcast.
template <int MT, int REM, bool VEC>
__global__ __launch_bounds__(256, (nnf_xty_wg_per_cu<MT, REM>())) void nnf_xty_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                         const float* __restrict__ Ut, int64_t ldu, int r,
                                                         float* __restrict__ slabs, int64_t ldp, int ncb, int nsplit,
                                                         int64_t rows_per_split, int a_vec_ok) {
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);   // tiles staged in LDS
    __shared__ f32x4 ldsA[2][MTA * 256];
    int ks, cb;
    nnf_xcd_map(blockIdx.x, ncb, ks, cb);
    if (ks >= nsplit) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, g = lane >> 4;
    const int64_t i_begin = (int64_t)ks * rows_per_split;
    const int64_t i_end = (i_begin + rows_per_split < m) ? (i_begin + rows_per_split) : m;
    const int nchunk = (int)((i_end - i_begin + 63) >> 6);
    const int64_t jl = (int64_t)cb * 256 + w * 64 + 4 * jj;  // lane's first column

    const rsrc_t rs = nnf_make_rsrc(X + i_begin * ldx, (uint32_t)(((i_end - i_begin - 1) * ldx + n) * 4));
    // lanes whose columns lie outside the matrix read nothing (offset beyond num_records -> 0)
    const int voff = (jl < n) ? (int)(((int64_t)4 * g * ldx + jl) * 4) : (int)0x7ffffff0;
    const int ldx4 = (int)(ldx * 4);

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[mt][cc] = f32x4{0.f, 0.f, 0.f, 0.f};

    // X prefetch ring: two 16-row groups (8 KB per wave) ahead of the MFMAs; with three workgroups per CU and the
    // per-group scheduling fence below that is enough in flight, and it keeps the kernel under 168 VGPRs without spills
    f32x4 xb[2][4];  // [group parity][k-step c]: row i_begin + 16*gi + 4g + c, columns jl..jl+3
    f32x4 areg[MTA];
    f32x4 ev[REM > 0 ? REM : 1];   // leftover rows: partial sums over this lane's rows, columns jl..jl+3
#pragma unroll
    for (int rr = 0; rr < (REM > 0 ? REM : 1); ++rr) ev[rr] = f32x4{0.f, 0.f, 0.f, 0.f};

    stageA_load<MTA>(Ut, ldu, r, i_end, i_begin, a_vec_ok, areg);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[t][c] = nnf_bload4<VEC>(rs, voff, (16 * t + c) * ldx4);
    stageA_store<MTA>(ldsA[0], areg);
    __syncthreads();

    for (int q = 0; q < nchunk; ++q) {
        const f32x4* img = ldsA[q & 1];
        // next chunk's A tile: global loads now, LDS write after the MFMAs (rows past i_end come back as zeros)
        stageA_load<MTA>(Ut, ldu, r, i_end, i_begin + 64 * (int64_t)(q + 1), a_vec_ok, areg);
        const int soff_q = q * 64 * ldx4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = img[(mt * 4 + t) * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[mt][cc] = MFMA16(af[mt][c], xb[t & 1][c][cc], acc[mt][cc]);
            if constexpr (REM > 0) {
#pragma unroll
                for (int rr = 0; rr < REM; ++rr) {
                    const f32x4 uv = img[(MT * 4 + t) * 64 + 16 * g + rr];   // Ut[16MT+rr][row 16t+4g+c], c = 0..3
#pragma unroll
                    for (int c = 0; c < 4; ++c) ev[rr] = __builtin_elementwise_fma(f32x4{uv[c], uv[c], uv[c], uv[c]}, xb[t & 1][c], ev[rr]);
                }
            }
            // refill the registers just consumed with the rows two groups ahead (past the end: zeros)
#pragma unroll
            for (int c = 0; c < 4; ++c) xb[t & 1][c] = nnf_bload4<VEC>(rs, voff, soff_q + (16 * (t + 2) + c) * ldx4);
            // keep every group's loads and leftover-row FMAs inside the group: without the fence hipcc moves all 16 refill
            // loads and the whole VALU part to the end of the chunk, where nothing is left to hide them behind
            __builtin_amdgcn_sched_barrier(0);
        }
        stageA_store<MTA>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads();
    }

    // epilogue: D[row = 4g+reg][col = jj] of tile (mt, cc) is out[16mt+4g+reg][jl+cc] -> one float4 per (mt, reg)
    if (jl < ldp) {
        float* sl = slabs + (int64_t)ks * r * ldp;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rk = 16 * mt + 4 * g + reg;
                if (rk < r) {
                    f32x4 o = {acc[mt][0][reg], acc[mt][1][reg], acc[mt][2][reg], acc[mt][3][reg]};
                    *reinterpret_cast<f32x4*>(sl + (int64_t)rk * ldp + jl) = o;
                }
            }
    }
    if constexpr (REM > 0) {   // sum the four row groups (lanes l, l^16, l^32, l^48), lanes of group 0 store
        float* sl = slabs + (int64_t)ks * r * ldp;
#pragma unroll
        for (int rr = 0; rr < REM; ++rr) {
            f32x4 e = ev[rr];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = e[c];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                e[c] = x;
            }
            const int rk = 16 * MT + rr;
            if (g == 0 && rk < r && jl < ldp) *reinterpret_cast<f32x4*>(sl + (int64_t)rk * ldp + jl) = e;
        }
    }
}

// out[row][col] = sum_s slabs[s][row][col]  (fp64 accumulate, slab order fixed -> bitwise reproducible)
__global__ __launch_bounds__(256) void nnf_reduce_slabs_kernel(const float* __restrict__ slabs, int nslab,
                                                               int64_t slab_stride, int rows, int64_t cols, int64_t lds,
                                                               float* __restrict__ out, int64_t ldo, double* __restrict__ out64) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / cols, col = e - row * cols;
        const float* p = slabs + row * lds + col;
        double s = 0.0;
        // eight slabs in flight, added in slab order (a load per trip waited for alone is a memory round trip per slab)
        for (int k = 0; k < nslab; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u < nslab ? k + u : nslab - 1) * slab_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (k + u < nslab) ? (double)v[u] : 0.0;
        }
        out[row * ldo + col] = (float)s;
        if (out64) out64[e] = s;          // (the Gram before it is rounded to fp32: nnf_gram_f64_f32)
    }
}

// same sums, four columns per thread (16-byte slab loads); needs lds % 4 == 0 and 16-byte aligned slabs
__global__ __launch_bounds__(256) void nnf_reduce_slabs4_kernel(const float* __restrict__ slabs, int nslab,
                                                                int64_t slab_stride, int rows, int64_t cols, int64_t lds,
                                                                float* __restrict__ out, int64_t ldo) {
    const int64_t cq = (cols + 3) >> 2, total = (int64_t)rows * cq;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / cq, col = 4 * (e - row * cq);
        const float* p = slabs + row * lds + col;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int k = 0; k < nslab; k += 4) {   // four slabs in flight, added in slab order
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + u < nslab ? k + u : nslab - 1) * slab_stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = k + u < nslab;
                s0 += in ? (double)v[u][0] : 0.0;
                s1 += in ? (double)v[u][1] : 0.0;
                s2 += in ? (double)v[u][2] : 0.0;
                s3 += in ? (double)v[u][3] : 0.0;
            }
        }
        float* o = out + row * ldo + col;
        o[0] = (float)s0;
        if (col + 1 < cols) o[1] = (float)s1;
        if (col + 2 < cols) o[2] = (float)s2;
        if (col + 3 < cols) o[3] = (float)s3;
    }
}

// Few output elements, many slabs (MTTKRP: 15000 elements x 256 slabs took 64 us with one thread per element): P threads
// per element, each summing every P-th... a contiguous range of slabs, the P partials combined in part order through LDS.
// Fixed order -> bitwise reproducible.  Threads with the same part are consecutive in the element index (coalesced).
template <int P>
__global__ __launch_bounds__(256) void nnf_reduce_slabs_par_kernel(const float* __restrict__ slabs, int nslab,
                                                                   int64_t slab_stride, int rows, int64_t cols, int64_t lds,
                                                                   float* __restrict__ out, int64_t ldo, double* __restrict__ out64) {
    constexpr int EPB = 256 / P;                 // elements per workgroup
    __shared__ double part_sum[P][EPB];
    const int el = threadIdx.x % EPB, part = threadIdx.x / EPB;
    const int64_t total = (int64_t)rows * cols;
    const int per = (nslab + P - 1) / P;
    const int k0 = part * per, k1 = (k0 + per < nslab) ? (k0 + per) : nslab;
    for (int64_t e0 = (int64_t)blockIdx.x * EPB; e0 < total; e0 += (int64_t)gridDim.x * EPB) {
        const int64_t e = e0 + el;
        double s = 0.0;
        if (e < total) {
            const int64_t row = e / cols, col = e - row * cols;
            const float* p = slabs + row * lds + col;
            // this part's slabs eight at a time: all loads of a batch in flight, added in slab order (one load per trip, each
            // waited for alone, was 12-16 dependent memory round trips: 16 us behind W^T X at config B)
            for (int k = k0; k < k1; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u < k1 ? k + u : k1 - 1) * slab_stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (k + u < k1) ? (double)v[u] : 0.0;
            }
        }
        part_sum[part][el] = s;
        __syncthreads();
        if (part == 0 && e < total) {
            double t = part_sum[0][el];
#pragma unroll
            for (int q = 1; q < P; ++q) t += part_sum[q][el];
            const int64_t row = e / cols, col = e - row * cols;
            out[row * ldo + col] = (float)t;
            if (out64) out64[e] = t;
        }
        __syncthreads();
    }
}

int nnf_launch_reduce_slabs(const float* slabs, int nslab, int64_t slab_stride, int rows, int64_t cols, int64_t lds,
                            float* out, int64_t ldo, hipStream_t st, double* out64) {
    {   // enough threads to fill the chip: P parts per element when the output is small
        const int64_t total = (int64_t)rows * cols;
        int P = 1;
        while (P < 16 && total * P < ((int64_t)1 << 19) && 2 * P <= nslab) P *= 2;
        if (P > 1) {
            const int epb = 256 / P;
            int64_t grid = (total + epb - 1) / epb;
            if (grid > 4096) grid = 4096;
#define NNF_RSP(PP)                                                                                                          \
    hipLaunchKernelGGL(nnf_reduce_slabs_par_kernel<PP>, dim3((int)grid), dim3(256), 0, st, slabs, nslab, slab_stride, rows, cols, \
                       lds, out, ldo, out64)
            if (P == 2) NNF_RSP(2);
            else if (P == 4) NNF_RSP(4);
            else if (P == 8) NNF_RSP(8);
            else NNF_RSP(16);
#undef NNF_RSP
            NNF_CHECK_LAUNCH();
            return NNF_OK;
        }
    }
    // (four columns per thread only when that still leaves enough threads to fill the chip: 20 vs 16 us at r x n = 1e5)
    if (out64 == nullptr && (lds & 3) == 0 && (slab_stride & 3) == 0 && (((uintptr_t)slabs) & 15) == 0 && (int64_t)rows * cols >= ((int64_t)1 << 21)) {
        const int64_t total4 = (int64_t)rows * ((cols + 3) >> 2);
        int grid4 = (int)((total4 + 255) / 256);
        if (grid4 > 2048) grid4 = 2048;
        hipLaunchKernelGGL(nnf_reduce_slabs4_kernel, dim3(grid4), dim3(256), 0, st, slabs, nslab, slab_stride, rows, cols, lds, out,
                           ldo);
        NNF_CHECK_LAUNCH();
        return NNF_OK;
    }
    const int64_t total = (int64_t)rows * cols;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(nnf_reduce_slabs_kernel, dim3(grid), dim3(256), 0, st, slabs, nslab, slab_stride, rows, cols, lds,
                       out, ldo, out64);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

template <int MT, int REM, bool VEC>
static int launch_xty(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int r,
                      int64_t ldu, float* out, int64_t ldo, hipStream_t st) {
    const int ncb = (int)nnf_cdiv(n, 256);
    const int64_t ldp = nnf_rup(n, 4);
    int64_t target = nnf_xty_wg_per_cu<MT, REM>() * (int64_t)ctx->num_cus / ncb;   // resident workgroups per CU
    if (target < 1) target = 1;
    int64_t nsplit = target;
    // a workgroup sums its rows in fp32 (MFMA accumulators); the slabs are added in fp64.  Cap the rows per workgroup: at
    // 1e6 x 4000 rank 100 the plan above is 16 splits of 62500 rows, and an entry of U^T X came out with 9.5e-7 relative rms
    // and a -2.2e-7 MEAN error (tools/probes/accum_error_probe.py) -- enough to take the Gram-identity cost of a HALS iteration
    // (which multiplies the mean by ||X||^2) to its 5e-4 bound.  The mean falls with the SQUARE of the chain length (62500 ->
    // 8192 rows: -2.2e-7 -> -3.7e-9, rms 9.5e-7 -> 1.2e-7), and at 8192 rows it was still what sent a 10^6 x 4000 rank-100 run
    // back to the streaming cost kernel after ~20 iterations (tools/probes/identity_terms_probe.py: bias term 3.3e4 of a 5.9e4
    // bound, actual error 1.4e4).  2048 rows: 489 slabs of 1.6 MB there (+10 % traffic on an MFMA-bound pass; fewer if the
    // context workspace is smaller -- the Python engine creates its main context with 1 GiB).
    const int64_t ROWS_CAP = 2048;
    if (nsplit < nnf_cdiv(m, ROWS_CAP)) nsplit = nnf_cdiv(m, ROWS_CAP);
    const int64_t max_split = nnf_cdiv(m, 64);
    if (nsplit > max_split) nsplit = max_split;
    // workspace bound
    const int64_t slab_elems = (int64_t)r * ldp;
    const int64_t ws_max = (int64_t)(cur.remaining() / 4) / slab_elems;
    if (ws_max < 1) return NNF_ERR_WORKSPACE;
    if (nsplit > ws_max) nsplit = ws_max;
    int64_t rows_per_split = nnf_rup(nnf_cdiv(m, nsplit), 64);
    // 32-bit buffer offsets inside one split
    while ((rows_per_split + 128) * ldx * 4 >= (int64_t)0x7fff0000) {
        if (rows_per_split <= 64) return NNF_ERR_UNSUPPORTED;
        rows_per_split = nnf_rup(rows_per_split / 2, 64);
    }
    nsplit = nnf_cdiv(m, rows_per_split);
    if (nsplit > ws_max) return NNF_ERR_WORKSPACE;
    float* slabs = (float*)cur.take((size_t)nsplit * slab_elems * 4);
    if (!slabs) return NNF_ERR_WORKSPACE;
    const int a_vec_ok = ((((uintptr_t)Ut) & 15) == 0 && (ldu & 3) == 0) ? 1 : 0;
    const int grid = 8 * (int)nnf_cdiv(nsplit, 8) * ncb;
    nnf_probe(ctx, NNF_PROBE_XTY, 0, st);   // measurement hook: the main kernel alone (bench.py)
    hipLaunchKernelGGL((nnf_xty_kernel<MT, REM, VEC>), dim3(grid), dim3(256), 0, st, X, m, n, ldx, Ut, ldu, r, slabs, ldp, ncb,
                       (int)nsplit, rows_per_split, a_vec_ok);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_XTY, 1, st);
    return nnf_launch_reduce_slabs(slabs, (int)nsplit, slab_elems, r, n, ldp, out, ldo, st);
}

// =========================================================================================================
// xht: out[rk][i] = sum_j V[rk][j] * X[i][j]
//   workgroup = 256 rows of X (wave w: rows 64w..64w+63 as four 16-row N tiles), k runs over the n columns.
//   B operand lane (ii = l&15, g = l>>4) of tile nt, k-group t: float4 X[i0w+16nt+ii][64q+16t+4g .. +3].
// =========================================================================================================
// NT = 16-row tiles per wave (a workgroup covers 64*NT rows starting at row0).
template <int MT, int REM, bool VEC, int NT>
__device__ __forceinline__ void nnf_xht_body(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                             const float* __restrict__ V, int64_t ldv, int r, float* __restrict__ out,
                                             int64_t ldo, int a_vec_ok, int64_t row0, f32x4 (*ldsA)[(MT + (REM > 0 ? 1 : 0)) * 256],
                                             int q0 = 0, int q1 = -1, int64_t oshift = 0) {
    // [q0, q1): the 64-column chunks this call contracts (default: all of them; a sub-range = a k-split share, see the kernel);
    // row i of the result goes to column i - oshift of `out` (a share's slab starts at the first k-split row)
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ii = lane & 15, g = lane >> 4;
    const int64_t i0w = row0 + 16 * NT * w;
    int64_t rows = m - i0w;
    if (rows > 16 * NT) rows = 16 * NT;
    const uint32_t bytes = rows > 0 ? (uint32_t)(((rows - 1) * ldx + n) * 4) : 0u;
    const rsrc_t rs = nnf_make_rsrc(X + (rows > 0 ? i0w : 0) * ldx, bytes);
    const int voff = (int)(((int64_t)ii * ldx + 4 * g) * 4);
    const int ldx4 = (int)(ldx * 4);
    const int nchunk = q1 >= 0 ? q1 : (int)((n + 63) >> 6);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xb[4][NT];  // [k-group t][row tile nt]
    f32x4 areg[MTA];
    float ev[REM > 0 ? REM : 1][NT];   // leftover rank rows x the 16-row tiles: partial over this lane's k
#pragma unroll
    for (int rr = 0; rr < (REM > 0 ? REM : 1); ++rr)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ev[rr][nt] = 0.f;

    stageA_load<MTA>(V, ldv, r, n, 64 * (int64_t)q0, a_vec_ok, areg);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[t][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldx4 + 256 * q0 + 64 * t);
    stageA_store<MTA>(ldsA[q0 & 1], areg);
    __syncthreads();

    for (int q = q0; q < nchunk; ++q) {
        const f32x4* img = ldsA[q & 1];
        stageA_load<MTA>(V, ldv, r, n, 64 * (int64_t)(q + 1), a_vec_ok, areg);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = img[(mt * 4 + t) * 64 + lane];
            // ragged k tail: never multiply a staged zero by out-of-row data
#if XHT_ABL != 5
            const int64_t nrem = n - (64 * (int64_t)q + 16 * t + 4 * g);
            if (nrem < 4) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c >= nrem) xb[t][nt][c] = 0.f;
            }
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
#if XHT_ABL == 2
                        if (mt == 0) acc[0][nt][c] += af[0][c] * xb[t][nt][c];
#else
                        acc[mt][nt] = MFMA16(af[mt][c], xb[t][nt][c], acc[mt][nt]);
#endif
                    }
            if constexpr (REM > 0 && XHT_ABL != 3) {
#pragma unroll
                for (int rr = 0; rr < REM; ++rr) {
                    const f32x4 uv = img[(MT * 4 + t) * 64 + 16 * g + rr];   // V[16MT+rr][64q+16t+4g+c], c = 0..3
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float e = ev[rr][nt];
#pragma unroll
                        for (int c = 0; c < 4; ++c) e = fmaf(uv[c], xb[t][nt][c], e);
                        ev[rr][nt] = e;
                    }
                }
            }
#if XHT_ABL != 1
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                xb[t][nt] = nnf_bload4<VEC>(rs, voff, nt * 16 * ldx4 + 256 * (q + 1) + 64 * t);
#endif
        }
#if XHT_ABL != 4
        stageA_store<MTA>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads();
#endif
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
                    if (rk < r) out[(int64_t)rk * ldo + (i - oshift)] = acc[mt][nt][reg];
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
                if (g == 0 && rk < r && i < m) out[(int64_t)rk * ldo + (i - oshift)] = x;
            }
    }
}

// A resident wave is the unit the MFMA pipe is shared in, and one round of workgroups covers B's 100000 rows: with
// 64 rows per wave everywhere that is 1563 waves on 1024 SIMDs -- the SIMDs holding two of them decide the time
// (8 row tiles against 6.1 on average).  The first n_hi workgroups take NTH tiles per wave, the others NTH-1, chosen
// on the host so that one full round of resident workgroups covers the matrix (7 tiles on the busiest SIMD).
template <int MT, int REM, bool VEC, int NTH>
__global__ __launch_bounds__(256, (MT + (REM > 0) <= 4 || NTH <= 2 ? 2 : 1)) void nnf_xht_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                         const float* __restrict__ V, int64_t ldv, int r,
                                                         float* __restrict__ out, int64_t ldo, int a_vec_ok, int n_hi,
                                                         float* __restrict__ tail_slabs, int64_t tail_row0, int64_t tail_ld,
                                                         int tail_tiles, int tail_parts, int tail_cpp) {
    constexpr int MTA = MT + (REM > 0 ? 1 : 0);
    __shared__ f32x4 ldsA[2][MTA * 256];
    const int b = (int)blockIdx.x;
    if (b < n_hi)
        nnf_xht_body<MT, REM, VEC, NTH>(X, m, n, ldx, V, ldv, r, out, ldo, a_vec_ok, (int64_t)b * (64 * NTH), ldsA);
    else
        nnf_xht_body<MT, REM, VEC, NTH - 1>(X, m, n, ldx, V, ldv, r, out, ldo, a_vec_ok,
                                            (int64_t)n_hi * (64 * NTH) + (int64_t)(b - n_hi) * (64 * (NTH - 1)), ldsA);
    // k-split tail (launch_xht): the row tiles that do not fill another whole round -- 106 of config B's 6250 -- are shared by ALL
    // workgroups instead of making 27 of them a third longer: workgroup b takes chunk share p = b % parts of the four tiles
    // 4 (b / parts) + wave, into slab p; the shares are added in share order by the usual slab reduction.
    if (tail_parts > 0) {
        const int p = b % tail_parts, tg = b / tail_parts;
        if (4 * tg < tail_tiles) {
            __syncthreads();
            const int nchunk_all = (int)((n + 63) >> 6);
            const int q0 = p * tail_cpp, q1 = (q0 + tail_cpp < nchunk_all) ? q0 + tail_cpp : nchunk_all;
            nnf_xht_body<MT, REM, VEC, 1>(X, m, n, ldx, V, ldv, r, tail_slabs + (int64_t)p * r * tail_ld, tail_ld, a_vec_ok,
                                          tail_row0 + 64 * (int64_t)tg, ldsA, q0 < q1 ? q0 : q1, q1, tail_row0);
        }
    }
}

template <int MT, int REM, bool VEC>
static int launch_xht(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V, int r, int64_t ldv,
                      float* out, int64_t ldo, hipStream_t st) {
    if (64 * ldx * 4 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    const int a_vec_ok = ((((uintptr_t)V) & 15) == 0 && (ldv & 3) == 0) ? 1 : 0;
    const int64_t slots = (int64_t)(MT + (REM > 0) <= 4 ? 2 : 1) * ctx->num_cus;   // resident workgroups
    const int64_t T = nnf_cdiv(m, 16), waves = 4 * slots;
    int nth = 4;
    int64_t n_hi, grid;
    float* tail_slabs = nullptr;
    int64_t tail_row0 = 0, tail_ld = 0;
    int tail_tiles = 0, tail_parts = 0, tail_cpp = 0;
    // six and more rank tiles (ranks 96 ... 128), many rounds: TWO row tiles per wave keep a wave at 248 registers, so that two
    // workgroups share a CU -- the four-tile form needs 404 (256 + 148 accumulation registers) and runs one wave per SIMD.
    // Measured (tools/probes/xht_nt2_probe.py, four -> two tiles): rank 100, 10^6 x 4000 7.16 -> 6.56 ms (0.71 -> 0.775 of the MFMA
    // peak), 500000 rows 3.58 -> 3.47, 250000 1.79 -> 1.73; rank 96 x 600000 2.17 -> 2.05; ranks 112 / 128 x 10^6 -4 % / -2 %;
    // below ~230000 rows (125000: 0.97 -> 1.00) and at five rank tiles (rank 80: 2.75 -> 2.79) the four-tile form stays ahead.
    // NNF_XHT_NT2=0 / 1 forces either form (A/B on one box).
    static const int nt2 = [] { const char* e = getenv("NNF_XHT_NT2"); return e ? atoi(e) : -1; }();
    const bool two_tiles = MT + (REM > 0) > 4 && T > 4 * waves && (nt2 >= 0 ? nt2 != 0 : (MT + (REM > 0) >= 6 && T > 14 * waves));
    if (two_tiles) {
        nth = 2;
        n_hi = grid = nnf_cdiv(m, 128);
    } else if (T > 4 * waves) {            // several rounds: 256-row workgroups
        n_hi = grid = nnf_cdiv(m, 256);
    } else if (T > 2 * waves) {     // one round: (4,3) or (3,2) tiles per wave
        nth = T > 3 * waves ? 4 : 3;
        n_hi = nnf_cdiv(T - 4 * (nth - 1) * slots, 4);
        grid = slots;
        // few tiles beyond a whole round of nth - 1 per wave (config B: 106 beyond 6144): every workgroup stays at nth - 1 and the
        // extra tiles are contracted in k-split shares by all of them -- 6 + 2 % on every SIMD instead of 7 tiles on the busiest.
        // (Not for ranks <= 32: the LDS-staged form is bit for bit the unsplit kernel there, tests.  NNF_XHT_TAIL=0 switches it off.)
        static const int tail_on = [] { const char* e = getenv("NNF_XHT_TAIL"); return e ? atoi(e) : 1; }();
        const int64_t extra = T - 4 * (nth - 1) * slots, nchunk_all = nnf_cdiv(n, 64);
        if (tail_on && MT + (REM > 0) >= 3 && extra > 0 && nchunk_all >= 4) {
            int parts = 32;
            while (parts > 1 && (parts > nchunk_all || 4 * (slots / parts) < extra)) parts >>= 1;
            if (parts >= 4 && 4 * (slots / parts) >= extra && 8 * extra <= T) {
                tail_parts = parts;
                tail_tiles = (int)extra;
                tail_cpp = (int)nnf_cdiv(nchunk_all, parts);
                tail_row0 = 64 * (int64_t)(nth - 1) * slots;
                tail_ld = nnf_rup(m - tail_row0, 4);
                tail_slabs = (float*)cur.take((size_t)parts * r * tail_ld * 4);
                if (tail_slabs) n_hi = 0;
                else tail_parts = 0;
            }
        }
    } else {                        // small: 128-row workgroups
        nth = 3;
        n_hi = 0;
        grid = nnf_cdiv(m, 128);
    }
    if (n_hi * 64 * nth + (grid - n_hi) * 64 * (nth - 1) + 16 * (int64_t)tail_tiles < m) return NNF_ERR_UNSUPPORTED;   // (cannot happen: the split covers m by construction)
    nnf_probe(ctx, NNF_PROBE_XHT, 0, st);
    if (nth == 4)
        hipLaunchKernelGGL((nnf_xht_kernel<MT, REM, VEC, 4>), dim3((int)grid), dim3(256), 0, st, X, m, n, ldx, V, ldv, r, out, ldo,
                           a_vec_ok, (int)n_hi, tail_slabs, tail_row0, tail_ld, tail_tiles, tail_parts, tail_cpp);
    else if (nth == 2) {
        if constexpr (MT + (REM > 0) > 4)
            hipLaunchKernelGGL((nnf_xht_kernel<MT, REM, VEC, 2>), dim3((int)grid), dim3(256), 0, st, X, m, n, ldx, V, ldv, r, out, ldo,
                               a_vec_ok, (int)n_hi, tail_slabs, tail_row0, tail_ld, tail_tiles, tail_parts, tail_cpp);
    } else
        hipLaunchKernelGGL((nnf_xht_kernel<MT, REM, VEC, 3>), dim3((int)grid), dim3(256), 0, st, X, m, n, ldx, V, ldv, r, out, ldo,
                           a_vec_ok, (int)n_hi, tail_slabs, tail_row0, tail_ld, tail_tiles, tail_parts, tail_cpp);
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_XHT, 1, st);
    if (tail_parts > 0)    // the k-split shares of the last rows, added in share order
        return nnf_launch_reduce_slabs(tail_slabs, tail_parts, (int64_t)r * tail_ld, r, m - tail_row0, tail_ld, out + tail_row0, ldo, st);
    return NNF_OK;
}

// =========================================================================================================
// gram: slab[ks] = A[:, split ks] * A[:, split ks]^T.  Both MFMA operands are the same LDS fragment image
// (B[k][col] = A[col][k] is the A-fragment of tile `col/16`).  Wave w owns tile rows {w, w+4}.
// =========================================================================================================
template <int MT>
__global__ __launch_bounds__(256) void nnf_gram_kernel(const float* __restrict__ A, int r, int64_t K, int64_t lda,
                                                       float* __restrict__ slabs, int64_t k_per_split, int a_vec_ok) {
    __shared__ f32x4 ldsA[2][MT * 256];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, g = lane >> 4;
    const int64_t k_begin = (int64_t)blockIdx.x * k_per_split;
    const int64_t k_end = (k_begin + k_per_split < K) ? (k_begin + k_per_split) : K;
    const int nchunk = (int)((k_end - k_begin + 63) >> 6);
    constexpr int NR = (MT + 3) / 4;  // tile rows per wave
    f32x4 acc[NR][MT];
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 areg[MT];
    stageA_load<MT>(A, lda, r, k_end, k_begin, a_vec_ok, areg);
    stageA_store<MT>(ldsA[0], areg);
    __syncthreads();
    for (int q = 0; q < nchunk; ++q) {
        const f32x4* img = ldsA[q & 1];
        stageA_load<MT>(A, lda, r, k_end, k_begin + 64 * (int64_t)(q + 1), a_vec_ok, areg);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 bf[MT];
#pragma unroll
            for (int b = 0; b < MT; ++b) bf[b] = img[(b * 4 + t) * 64 + lane];
#pragma unroll
            for (int a = 0; a < NR; ++a) {
                const int mt = w + 4 * a;
                if (mt < MT) {
                    const f32x4 af = img[(mt * 4 + t) * 64 + lane];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int b = 0; b < MT; ++b) acc[a][b] = MFMA16(af[c], bf[b][c], acc[a][b]);
                }
            }
        }
        stageA_store<MT>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads();
    }
    float* sl = slabs + (int64_t)blockIdx.x * r * r;
#pragma unroll
    for (int a = 0; a < NR; ++a) {
        const int mt = w + 4 * a;
        if (mt < MT) {
#pragma unroll
            for (int b = 0; b < MT; ++b)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * mt + 4 * g + reg, col = 16 * b + jj;
                    if (row < r && col < r) sl[row * r + col] = acc[a][b][reg];
                }
        }
    }
}

// Short factors (the I_mode x R factors of NTF / NTD: K <= 1024, r <= 64): one workgroup of eight waves, wave w owns the
// k range [w*kpw, (w+1)*kpw) and reads its operand fragments straight from global memory -- ALL of a wave's loads are in
// flight at once, so the kernel is one memory round trip + <= 128 MFMAs + one LDS reduction in fixed wave order.  (The
// chunked kernel above walks K in 64-wide LDS-staged chunks: eight dependent round trips for K = 500, 9 us of a 0.5 ms NTF
// iteration three times over.)  Both MFMA operands are the same registers: lane (i = l&15, g = l>>4) holds
// A[16*mt + i][k0 + 4g .. +3]; component c of every lane contracts k0 + 4g + c over g, the four components cover 16 k's.
template <int MT, int KS>
__global__ __launch_bounds__(512) void nnf_gram_small_kernel(const float* __restrict__ A, int r, int K, int64_t lda,
                                                             float* __restrict__ G) {
    __shared__ f32x4 red[4][MT * MT][64];   // 64 KB at MT = 4
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ii = lane & 15, g = lane >> 4;
    const int kw = w * (16 * KS);
    f32x4 fr[KS][MT];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = 16 * mt + ii, k = kw + 16 * s + 4 * g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < r && k < K) {   // K % 4 == 0 (vector path only): a lane's four k's are in or out together
                v = *reinterpret_cast<const f32x4*>(A + (int64_t)row * lda + k);
            }
            fr[s][mt] = v;
        }
    f32x4 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = MFMA16(fr[s][a][c], fr[s][b][c], acc[a][b]);
    // fixed order: wave w + 4 is added to wave w, then the four sums in order 0..3
    if (w >= 4) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) red[w - 4][a * MT + b][lane] = acc[a][b];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const f32x4 x = red[w][a * MT + b][lane];
                f32x4 y = acc[a][b];
                y[0] += x[0]; y[1] += x[1]; y[2] += x[2]; y[3] += x[3];
                red[w][a * MT + b][lane] = y;
            }
    }
    __syncthreads();
    // tile (a, b), lane l, register reg  <->  G[16a + 4(l>>4) + reg][16b + (l&15)]
    for (int e = threadIdx.x; e < MT * MT * 64; e += 512) {
        const int t = e >> 6, l = e & 63;
        f32x4 s = red[0][t][l];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const f32x4 x = red[ww][t][l];
            s[0] += x[0]; s[1] += x[1]; s[2] += x[2]; s[3] += x[3];
        }
        const int a = t / MT, b = t - a * MT, col = 16 * b + (l & 15);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * a + 4 * (l >> 4) + reg;
            if (row < r && col < r) G[row * r + col] = s[reg];
        }
    }
}

template <int MT>
static int launch_gram_small(const float* A, int r, int64_t K, int64_t lda, float* G, hipStream_t st) {
    const int ks = (int)nnf_cdiv(K, 128);   // 16-wide k steps per wave (8 waves)
#define GRAM_SMALL(KS)                                                                                                 \
    hipLaunchKernelGGL((nnf_gram_small_kernel<MT, KS>), dim3(1), dim3(512), 0, st, A, r, (int)K, lda, G)
    switch (ks) {
        case 1: GRAM_SMALL(1); break;
        case 2: GRAM_SMALL(2); break;
        case 3: GRAM_SMALL(3); break;
        case 4: GRAM_SMALL(4); break;
        case 5: GRAM_SMALL(5); break;
        case 6: GRAM_SMALL(6); break;
        case 7: GRAM_SMALL(7); break;
        default: GRAM_SMALL(8); break;
    }
#undef GRAM_SMALL
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

// G64[a][b] = (double)G[a][b]: the fp64 copy of a Gram that was formed without slabs (K <= 1024: one workgroup)
__global__ void nnf_gram_widen_kernel(const float* __restrict__ G, int64_t ldg, int r, double* __restrict__ G64) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < r * r) G64[e] = (double)G[(int64_t)(e / r) * ldg + (e % r)];
}
static int launch_gram_widen(const float* G, int64_t ldg, int r, double* G64, hipStream_t st) {
    if (!G64) return NNF_OK;
    hipLaunchKernelGGL(nnf_gram_widen_kernel, dim3((r * r + 255) / 256), dim3(256), 0, st, G, ldg, r, G64);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

template <int MT>
static int launch_gram(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                       hipStream_t st, double* G64) {
    if constexpr (MT <= 4) {
        if (K <= 1024 && (K & 3) == 0 && ldg == r && ((((uintptr_t)A) & 15) == 0) && (lda & 3) == 0) {
            const int rc = launch_gram_small<MT>(A, r, K, lda, G, st);
            return rc != NNF_OK ? rc : launch_gram_widen(G, ldg, r, G64, st);
        }
    }
    // one split per CU (the slab reduction spreads every output element over up to 16 threads, so its cost grows slowly
    // with the split count): 64 splits left a 50 x 100000 Gram at 22 us and a 100 x 125000 one at 127 us
    // (a workgroup walks its split in 64-wide LDS-staged chunks, one memory round trip each: with one split per CU the Gram of a
    // 50 x 100000 factor was 7 dependent round trips = 15.7 us in front of W^T X; two resident workgroups per CU halve the
    // chain and overlap each other's waits)
    int64_t nsplit = ctx->num_cus > 8 ? 2 * (int64_t)ctx->num_cus : 8;
    const int64_t max_split = nnf_cdiv(K, 64);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    // short factors (the I_mode x R factors of NTF / NTD, K <= 1024): one workgroup, no split, written straight into G -- the
    // whole Gram is a few microseconds of work and the slab reduction would be a second launch of the same length
    if (K <= 1024 && ldg == r) nsplit = 1;
    // keep the fp32 chain inside a split short (<= 512 columns; r*r*4 bytes of slab each): what the chains leave is all the error
    // the fp64 copy of the sums has (nnf_gram_f64_f32).  The same plan with and without the copy: the fp32 Gram does not depend
    // on which entry point formed it.
    if (nsplit > 1 && nsplit < nnf_cdiv(K, 512)) {
        nsplit = nnf_cdiv(K, 512);
        const int64_t ws_max = (int64_t)(cur.remaining() / 4) / ((int64_t)r * r);
        if (nsplit > ws_max) nsplit = ws_max > 0 ? ws_max : 1;
    }
    const int64_t kps = nnf_rup(nnf_cdiv(K, nsplit), 64);
    nsplit = nnf_cdiv(K, kps);
    const int a_vec_ok = ((((uintptr_t)A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    if (nsplit == 1 && ldg == r) {
        hipLaunchKernelGGL((nnf_gram_kernel<MT>), dim3(1), dim3(256), 0, st, A, r, K, lda, G, kps, a_vec_ok);
        NNF_CHECK_LAUNCH();
        return launch_gram_widen(G, ldg, r, G64, st);
    }
    float* slabs = (float*)cur.take((size_t)nsplit * r * r * 4);
    if (!slabs) return NNF_ERR_WORKSPACE;
    hipLaunchKernelGGL((nnf_gram_kernel<MT>), dim3((int)nsplit), dim3(256), 0, st, A, r, K, lda, slabs, kps, a_vec_ok);
    NNF_CHECK_LAUNCH();
    return nnf_launch_reduce_slabs(slabs, (int)nsplit, (int64_t)r * r, r, r, r, G, ldg, st, G64);
}

// Ranks above NNF_MAX_RANK: the Gram in 64 x 64 blocks.  Workgroup (split ks, block pair (bi, bj)) multiplies the k range of
// its split of row block bi by the same range of row block bj -- two LDS images instead of one, otherwise the kernel above.
// Block (bj, bi) is computed by its own workgroup from the same products in the same order, so the result is symmetric bit
// for bit, as the single-image kernel's is.  Slabs [split][r x r], reduced in split order by the usual launch.
__global__ __launch_bounds__(256) void nnf_gram_blocks_kernel(const float* __restrict__ A, int r, int64_t K, int64_t lda,
                                                              float* __restrict__ slabs, int64_t k_per_split, int a_vec_ok, int nb) {
    constexpr int MT = 4;
    __shared__ f32x4 ldsA[2][MT * 256], ldsB[2][MT * 256];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, g = lane >> 4;
    const int bi = (int)blockIdx.y / nb, bj = (int)blockIdx.y - bi * nb;
    const float* Ai = A + (int64_t)64 * bi * lda;
    const float* Bj = A + (int64_t)64 * bj * lda;
    const int ri = r - 64 * bi < 64 ? r - 64 * bi : 64, rj = r - 64 * bj < 64 ? r - 64 * bj : 64;
    const int64_t k_begin = (int64_t)blockIdx.x * k_per_split;
    const int64_t k_end = (k_begin + k_per_split < K) ? (k_begin + k_per_split) : K;
    const int nchunk = (int)((k_end - k_begin + 63) >> 6);
    f32x4 acc[MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 areg[MT], breg[MT];
    stageA_load<MT>(Ai, lda, ri, k_end, k_begin, a_vec_ok, areg);
    stageA_load<MT>(Bj, lda, rj, k_end, k_begin, a_vec_ok, breg);
    stageA_store<MT>(ldsA[0], areg);
    stageA_store<MT>(ldsB[0], breg);
    __syncthreads();
    for (int q = 0; q < nchunk; ++q) {
        const f32x4* imgA = ldsA[q & 1];
        const f32x4* imgB = ldsB[q & 1];
        stageA_load<MT>(Ai, lda, ri, k_end, k_begin + 64 * (int64_t)(q + 1), a_vec_ok, areg);
        stageA_load<MT>(Bj, lda, rj, k_end, k_begin + 64 * (int64_t)(q + 1), a_vec_ok, breg);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 af = imgA[(w * 4 + t) * 64 + lane];       // wave w owns tile row w of the block
            f32x4 bf[MT];
#pragma unroll
            for (int b = 0; b < MT; ++b) bf[b] = imgB[(b * 4 + t) * 64 + lane];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[b] = MFMA16(af[c], bf[b][c], acc[b]);
        }
        stageA_store<MT>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        stageA_store<MT>(const_cast<f32x4*>(ldsB[(q + 1) & 1]), breg);
        __syncthreads();
    }
    float* sl = slabs + (int64_t)blockIdx.x * r * r;
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 64 * bi + 16 * w + 4 * g + reg, col = 64 * bj + 16 * b + jj;
            if (row < r && col < r) sl[(int64_t)row * r + col] = acc[b][reg];
        }
}

static int launch_gram_blocks(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                              hipStream_t st, double* G64) {
    const int nb = (r + 63) / 64;
    // splits: about two workgroups per CU over all block pairs, 64-column chunks, as many slabs as the workspace holds
    int64_t nsplit = nnf_cdiv((int64_t)2 * ctx->num_cus, (int64_t)nb * nb);
    const int64_t max_split = nnf_cdiv(K, 64);
    if (nsplit > max_split) nsplit = max_split;
    const int64_t ws_max = (int64_t)(cur.remaining() / 4) / ((int64_t)r * r);
    if (ws_max < 1) return NNF_ERR_WORKSPACE;
    if (nsplit < nnf_cdiv(K, 512)) nsplit = nnf_cdiv(K, 512);   // (short fp32 chains, as in launch_gram)
    if (nsplit > ws_max) nsplit = ws_max;
    if (nsplit < 1) nsplit = 1;
    const int64_t kps = nnf_rup(nnf_cdiv(K, nsplit), 64);
    nsplit = nnf_cdiv(K, kps);
    const int a_vec_ok = ((((uintptr_t)A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    float* slabs = (float*)cur.take((size_t)nsplit * r * r * 4);
    if (!slabs) return NNF_ERR_WORKSPACE;
    hipLaunchKernelGGL(nnf_gram_blocks_kernel, dim3((int)nsplit, nb * nb), dim3(256), 0, st, A, r, K, lda, slabs, kps, a_vec_ok, nb);
    NNF_CHECK_LAUNCH();
    return nnf_launch_reduce_slabs(slabs, (int)nsplit, (int64_t)r * r, r, r, r, G, ldg, st, G64);
}

// =========================================================================================================
// frob: sum_ij (X[i][j] - sum_k Ut[k][i] V[k][j])^2
//   workgroup = 128 rows (wave w: rows 32w..32w+31 as two 16-row M tiles), sweeping all columns in 64-wide blocks.
//   P tile: A[i = l&15][k] = Ut[k][i] fragments (whole rank, staged once), B[k][col] = V[k][j0+4jj+c] fragments
//   (restaged per column block, fragment order), D[row = 4g+reg][col = jj] of N tile cc <-> column j0+4jj+cc,
//   which is exactly what a lane's float4 load of X[row][j0+4jj..+3] holds.
//   The rank loop is a run-time loop (both operands come from LDS), so one kernel serves every r <= 128.
// =========================================================================================================
// Right-operand fragments of ALL column blocks, laid out exactly as the cost kernel stages them:
//   Vf[(blk*KS + s)*64 + L] = float4 V[4s + (L>>4)][64 blk + 4(L&15) .. +3]   (zero outside r x n)
// V is tiny (r x n) and identical for every workgroup, so the index arithmetic, the ragged-edge masks and (CP cost) the
// Khatri-Rao products V[k][ja]*Vb[k][jb] are done once here instead of once per workgroup and column block.
__global__ __launch_bounds__(256) void nnf_cost_prepv_kernel(const float* __restrict__ V, int64_t ldv, int r, int64_t n, int KS,
                                                             const float* __restrict__ Vb, int64_t ldvb, int64_t nb,
                                                             f32x4* __restrict__ Vf, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int L = (int)(e & 63);
        const int64_t q = e >> 6;
        const int s_ = (int)(q % KS);
        const int64_t blk = q / KS;
        const int k = 4 * s_ + (L >> 4);
        const int64_t j = 64 * blk + 4 * (L & 15);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < r && j < n) {
            if (Vb == nullptr) {
                const float* p = V + (int64_t)k * ldv + j;
                v[0] = p[0];
                if (j + 1 < n) v[1] = p[1];
                if (j + 2 < n) v[2] = p[2];
                if (j + 3 < n) v[3] = p[3];
            } else {  // Khatri-Rao column j = (ja, jb), jb fastest: V[k][ja] * Vb[k][jb]
                const int64_t ja0 = j / nb, jb0 = j - ja0 * nb;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (j + c < n) {
                        int64_t ja = ja0, jb = jb0 + c;
                        while (jb >= nb) { jb -= nb; ++ja; }
                        v[c] = V[(int64_t)k * ldv + ja] * Vb[(int64_t)k * ldvb + jb];
                    }
                }
            }
        }
        Vf[e] = v;
    }
}

// PIN (ranks above 128, walked in chunks of <= 128): the model of the EARLIER rank chunks, m x n with row stride ldr in Pin, is
// added to this chunk's product before the element-wise part -- read one block ahead like X (Pin may be R1: a lane reads its
// own elements of a block before it writes them).
template <int OP, bool VEC, int NV, bool PIN = false>   // NV = float4 pieces of a V image per thread: 4 up to r = 64, 8 up to r = 128
__global__ __launch_bounds__(256, PIN ? 2 : 3) void nnf_cost_kernel(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
                                                          const float* __restrict__ Ut, int64_t ldu,
                                                          const f32x4* __restrict__ Vf, int r,
                                                          float beta, double* __restrict__ partial,
                                                          const float* __restrict__ Ub, int64_t ldub, int64_t nbu,
                                                          float* __restrict__ R1, float* __restrict__ R2, int64_t ldr,
                                                          int u_vec_ok, int vdb, const float* Pin = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = (r + 3) >> 2;                               // k-steps of 4
    float* ldsU = reinterpret_cast<float*>(smem);              // [wave 4][rt 2][KS][64]
    // V image: two buffers (vdb = 1: the next block is written while this one is read, one barrier per block) or ONE (vdb = 0:
    // ranks 77..104, where the second buffer is what keeps a second workgroup off the CU -- 100 KB against 75 KB of the
    // 160 KB; with a single wave per SIMD the rank-100 cost pass of config E ran at 0.34 of the MFMA peak)
    f32x4* ldsV = reinterpret_cast<f32x4*>(smem + (size_t)4 * 2 * KS * 64 * 4);  // [vdb ? 2 : 1][KS][64] float4
    double* red = reinterpret_cast<double*>(smem + (size_t)4 * 2 * KS * 64 * 4 + (size_t)(vdb ? 2 : 1) * KS * 64 * 16);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jj = lane & 15, g = lane >> 4;
    const int64_t i0w = (int64_t)blockIdx.x * 128 + 32 * w;
    int64_t rows = m - i0w;
    if (rows > 32) rows = 32;
    const uint32_t bytes = rows > 0 ? (uint32_t)(((rows - 1) * ldx + n) * 4) : 0u;
    const rsrc_t rs = nnf_make_rsrc(X + (rows > 0 ? i0w : 0) * ldx, bytes);
    const int ldx4 = (int)(ldx * 4);
    const int voff = (int)(((int64_t)4 * g * ldx + 4 * jj) * 4);
    // column blocks [blk0, blk1) of this workgroup: blockIdx.y splits the column range so that the grid has several times
    // more workgroups than resident slots (128-row workgroups alone give 782 for 512 slots at B: a 1.5-round tail)
    const int nblk_all = (int)((n + 63) >> 6);
    const int per = (nblk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int blk0 = (int)blockIdx.y * per;
    const int nblk = (blk0 + per < nblk_all) ? (blk0 + per) : nblk_all;

    // U fragments of this wave's 32 rows: ldsU[w][rt][s][lane] = Ut[4s + (lane>>4)][i0w + 16rt + (lane&15)]
    if (Ub == nullptr && u_vec_ok && rows == 32) {
        // four consecutive rows i of one rank row k are four consecutive floats of the image: 16-byte loads straight
        // into 16-byte LDS stores, all of a wave's loads in flight together (the element-wise loop below costs a
        // division and a dependent round trip per element -- a prologue as long as the MFMA work of a short column range)
        const int nq = 8 * r;                      // float4 pieces: (k, j) -> Ut[k][i0w + 4j .. +3], j = 0..7
        float* dstw = ldsU + (size_t)(w * 2) * KS * 64;
        for (int e0 = 0; e0 < nq; e0 += 4 * 64) {
            f32x4 piece[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 64 * u + lane;
                piece[u] = (e < nq) ? *reinterpret_cast<const f32x4*>(Ut + (int64_t)(e >> 3) * ldu + i0w + 4 * (e & 7))
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 64 * u + lane;
                const int k = e >> 3, j = e & 7;
                if (e < nq)
                    *reinterpret_cast<f32x4*>(dstw + ((j >> 2) * KS + (k >> 2)) * 64 + (k & 3) * 16 + 4 * (j & 3)) = piece[u];
            }
        }
        if (r & 3) {   // rank rows r .. 4KS-1 of the last k-step are zero
            const int k = r + (lane >> 4);
            if (k < 4 * KS) {
                dstw[(0 * KS + (k >> 2)) * 64 + (k & 3) * 16 + (lane & 15)] = 0.f;
                dstw[(1 * KS + (k >> 2)) * 64 + (k & 3) * 16 + (lane & 15)] = 0.f;
            }
        }
    } else {
        // element-wise staging (ragged last workgroup, unaligned U, Khatri-Rao rows of the CP cost): a lane's entries are
        // (rt, s) -> Ut[4s + (lane>>4)][i0w + 16rt + (lane&15)], i.e. only TWO tensor rows per lane (one division each for the
        // Khatri-Rao split), and the loads go out eight at a time from clamped addresses.  (One entry per trip with its
        // loads under the `k < r && i < m` test was 2 KS dependent round trips + divisions in front of the first MFMA --
        // longer than the MFMA work of a CP-cost workgroup: 4 column blocks.)
        const int L = lane;
        int64_t ia0, ib0, ia1, ib1;
        const int64_t i_0 = i0w + (L & 15), i_1 = i_0 + 16;
        const bool ok0 = i_0 < m, ok1 = i_1 < m;
        if (Ub == nullptr) { ia0 = ok0 ? i_0 : 0; ia1 = ok1 ? i_1 : 0; ib0 = ib1 = 0; }
        else {
            const int64_t c0 = ok0 ? i_0 : 0, c1 = ok1 ? i_1 : 0;
            ia0 = c0 / nbu; ib0 = c0 - ia0 * nbu;
            ia1 = c1 / nbu; ib1 = c1 - ia1 * nbu;
        }
        float* dstw = ldsU + (size_t)(w * 2) * KS * 64 + L;
        auto stage = [&](auto kr) {   // kr: with / without the second (Khatri-Rao) factor -- no per-entry branch either way
            constexpr bool KR = decltype(kr)::value;
            for (int j0 = 0; j0 < 2 * KS; j0 += 8) {
                float ua[8], ub[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u < 2 * KS ? j0 + u : 0;
                    const bool rt = j >= KS;
                    const int k = 4 * (j - (rt ? KS : 0)) + (L >> 4);
                    const bool ok = k < r && (rt ? ok1 : ok0);
                    const int64_t kc = ok ? k : 0;
                    ua[u] = Ut[kc * ldu + (ok ? (rt ? ia1 : ia0) : 0)];
                    if constexpr (KR) ub[u] = Ub[kc * ldub + (ok ? (rt ? ib1 : ib0) : 0)];
                    else ub[u] = 1.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u;
                    if (j < 2 * KS) {
                        const bool rt = j >= KS;
                        const int k = 4 * (j - (rt ? KS : 0)) + (L >> 4);
                        const bool ok = k < r && (rt ? ok1 : ok0);
                        dstw[j * 64] = ok ? ua[u] * ub[u] : 0.f;
                    }
                }
            }
        };
        if (Ub != nullptr) stage(std::true_type{});
        else stage(std::false_type{});
    }
    // V fragments of one 64-column block: img[s][lane] = float4 V[4s + (lane>>4)][j0 + 4(lane&15) .. +3].
    // Staged in two halves: global loads into registers before the MFMAs of the current block, LDS writes after them.
    f32x4 vreg[NV];
    auto stageV_load = [&](int blk) {   // straight copies of the pre-arranged fragments (nnf_cost_prepv_kernel)
        const f32x4* src = Vf + (size_t)blk * KS * 64;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = threadIdx.x + 256 * u;
            vreg[u] = (e < KS * 64 && blk < nblk_all) ? src[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stageV_store = [&](f32x4* img) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = threadIdx.x + 256 * u;
            if (e < KS * 64) img[e] = vreg[u];
        }
    };
    stageV_load(blk0);
    stageV_store(ldsV + (size_t)(vdb ? (blk0 & 1) : 0) * KS * 64);
    // X one block ahead.  (A two-block ring was tried: same time, 12 more registers -- and at <= 136 registers three of these
    // waves leave room on a SIMD for a wave of the persistent V-side sweep kernel, which the outer loop overlaps this
    // kernel with.)  The block past the workgroup's column range is "read" through an out-of-range offset: zeros, no
    // memory traffic (the prefetch used to fetch the next rows' data: 1.24 GB instead of 0.82 GB per launch).
    f32x4 xb[2][4];  // [rt][reg]: row i0w + 16rt + 4g + reg, columns j0+4jj..+3
    f32x4 pb[PIN ? 2 : 1][PIN ? 4 : 1];   // the same elements of Pin
    rsrc_t rsp = rs;
    int ldp4 = 0, voffp = 0;
    if constexpr (PIN) {
        rsp = nnf_make_rsrc(Pin + (rows > 0 ? i0w : 0) * ldr, rows > 0 ? (uint32_t)(((rows - 1) * ldr + n) * 4) : 0u);
        ldp4 = (int)(ldr * 4);
        voffp = (int)(((int64_t)4 * g * ldr + 4 * jj) * 4);
    }
    auto xload = [&](int blk) {
        const int vo = (blk < nblk) ? voff : (int)0x7ffffff0;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) xb[rt][reg] = nnf_bload4<VEC>(rs, vo, (16 * rt + reg) * ldx4 + 256 * blk);
        if constexpr (PIN) {
            const int vp = (blk < nblk) ? voffp : (int)0x7ffffff0;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) pb[rt][reg] = nnf_bload4<VEC>(rsp, vp, (16 * rt + reg) * ldp4 + 256 * blk);
        }
    };
    xload(blk0);
    __syncthreads();

    double dsum = 0.0;
    const float* uf = ldsU + (size_t)(w * 2) * KS * 64 + lane;
    for (int blk = blk0; blk < nblk; ++blk) {
        const f32x4* img = ldsV + (size_t)(vdb ? (blk & 1) : 0) * KS * 64;
        stageV_load(blk + 1);   // past the last block every entry is masked to zero (j >= n)
        f32x4 acc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acc[rt][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k loop, two steps per trip, software-pipelined by hand: the fragments of the next step are read from LDS (three
        // hand-issued ds_reads) while the MFMAs of the current one run.  The compiler rotates a C++ version of this back
        // into "read, wait, multiply" (tools/ notes: ~30 % of the MFMA time exposed).  Nothing is in flight at the loop
        // back-edge or at any other control-flow merge, which is what makes hand-issued loads safe (k_hals_quad.hip).
        {
            const unsigned bvb = (unsigned)(uintptr_t)img + (unsigned)lane * 16u;        // + 1024 per k-step
            const unsigned a0b = (unsigned)(uintptr_t)uf;                                 // + 256 per k-step
            const unsigned a1b = a0b + (unsigned)KS * 256u;
            f32x4 bvA, bvB;
            float a0A, a1A, a0B, a1B;
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(bvA), "=&v"(a0A), "=&v"(a1A) : "v"(bvb), "v"(a0b), "v"(a1b));
            int s = 0;
            for (; s + 1 < KS; s += 2) {
                const int sb = s + 1, sa = (s + 2 < KS) ? s + 2 : KS - 1;   // the clamped extra read is never used
                // (the A set rides through the statement as in/out operands so that its MFMAs cannot be scheduled above it,
                //  and the accumulators ride through the wait so that they cannot sink below it)
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b32 %1, %7\n\tds_read_b32 %2, %8"
                             : "=&v"(bvB), "=&v"(a0B), "=&v"(a1B), "+v"(bvA), "+v"(a0A), "+v"(a1A)
                             : "v"(bvb + (unsigned)sb * 1024u), "v"(a0b + (unsigned)sb * 256u), "v"(a1b + (unsigned)sb * 256u));
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    acc[0][cc] = MFMA16(a0A, bvA[cc], acc[0][cc]);
                    acc[1][cc] = MFMA16(a1A, bvA[cc], acc[1][cc]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(bvB), "+v"(a0B), "+v"(a1B), "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                               "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
                asm volatile("ds_read_b128 %0, %6\n\tds_read_b32 %1, %7\n\tds_read_b32 %2, %8"
                             : "=&v"(bvA), "=&v"(a0A), "=&v"(a1A), "+v"(bvB), "+v"(a0B), "+v"(a1B)
                             : "v"(bvb + (unsigned)sa * 1024u), "v"(a0b + (unsigned)sa * 256u), "v"(a1b + (unsigned)sa * 256u));
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    acc[0][cc] = MFMA16(a0B, bvB[cc], acc[0][cc]);
                    acc[1][cc] = MFMA16(a1B, bvB[cc], acc[1][cc]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(bvA), "+v"(a0A), "+v"(a1A), "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                               "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
            }
            if (s < KS) {   // odd number of k-steps: the A set holds step KS-1
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    acc[0][cc] = MFMA16(a0A, bvA[cc], acc[0][cc]);
                    acc[1][cc] = MFMA16(a1A, bvA[cc], acc[1][cc]);
                }
            }
        }
        // residual of this 32 x 64 block; columns past n hold the next row's data -> masked out
        const int64_t jrem = n - (64 * (int64_t)blk + 4 * jj);
        float loc = 0.f;
        if constexpr (PIN) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[rt][cc][reg] += pb[rt][reg][cc];
        }
        if constexpr (OP == NNF_RATIO_KL || OP == NNF_RATIO_GEN || OP == NNF_PROD) {
            // large-rank MU (r > 64): the element-wise operands are written out, the two contractions follow as plain
            // X H^T / W^T X launches on them (k_mu.hip).  One float4 per (row piece): columns j0+4jj .. +3.
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int rowl = 16 * rt + 4 * g + reg;
                    if (rowl >= rows) continue;
                    f32x4 o1, o2;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float p = acc[rt][cc][reg], x = xb[rt][reg][cc];
                        if constexpr (OP == NNF_PROD) {
                            o1[cc] = p;
                            o2[cc] = 0.f;
                        } else if constexpr (OP == NNF_RATIO_KL) {
                            o1[cc] = x * __builtin_amdgcn_rcpf(p);
                            o2[cc] = 0.f;
                        } else {
                            const float lp = __builtin_amdgcn_logf(p);
                            o2[cc] = __builtin_amdgcn_exp2f((beta - 1.f) * lp);
                            o1[cc] = o2[cc] * __builtin_amdgcn_rcpf(p) * x;
                        }
                    }
                    const int64_t off = (i0w + rowl) * ldr + 64 * (int64_t)blk + 4 * jj;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        if (cc < jrem) {
                            R1[off + cc] = o1[cc];
                            if constexpr (OP == NNF_RATIO_GEN) R2[off + cc] = o2[cc];
                        }
                    }
                }
        } else
        if (rows == 32 && 64 * (int64_t)(blk + 1) <= n) {   // interior block (wave-uniform): no edge masks
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) loc += nnf_cost_term<OP>(xb[rt][reg][cc], acc[rt][cc][reg], beta);
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (16 * rt + 4 * g + reg >= rows) continue;   // rows past the end of the matrix
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        if (cc < jrem) loc += nnf_cost_term<OP>(xb[rt][reg][cc], acc[rt][cc][reg], beta);
                    }
                }
        }
        dsum += (double)loc;
        xload(blk + 1);
        if (!vdb) __syncthreads();   // single buffer: every wave is past its last read of this block's image
        stageV_store(ldsV + (size_t)(vdb ? ((blk + 1) & 1) : 0) * KS * 64);
        __syncthreads();
    }
    const double bs = nnf_block_sum_f64(dsum, red);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = bs;
}

// sum of `count` doubles in index order by one workgroup -> out[0]
__global__ __launch_bounds__(256) void nnf_sum_partials_kernel(const double* __restrict__ partial, int64_t count,
                                                               double scale, double* __restrict__ out) {
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
    if (threadIdx.x == 0) out[0] = t * scale;
}


int nnf_launch_sum_f64(const double* partial, int64_t count, double scale, double* out, hipStream_t st) {
    hipLaunchKernelGGL(nnf_sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, count, scale, out);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

#define DISPATCH_MT(FN, REM, VEC, ...)                    \
    switch (MT) {                                         \
        case 1: return FN<1, REM, VEC>(__VA_ARGS__);      \
        case 2: return FN<2, REM, VEC>(__VA_ARGS__);      \
        case 3: return FN<3, REM, VEC>(__VA_ARGS__);      \
        case 4: return FN<4, REM, VEC>(__VA_ARGS__);      \
        case 5: return FN<5, REM, VEC>(__VA_ARGS__);      \
        case 6: return FN<6, REM, VEC>(__VA_ARGS__);      \
        case 7: return FN<7, REM, VEC>(__VA_ARGS__);      \
        default: return FN<8, REM, VEC>(__VA_ARGS__);     \
    }
// rank -> (MFMA tiles, VALU leftover rows): r = 16q + (1..4), q >= 1 keeps q tiles on MFMA and 2 or 4 rows on VALU
static inline void split_rank(int r, int& MT, int& REM) {
    const int q = r / 16, rem = r % 16;
    if (q >= 1 && q <= 7 && rem >= 1 && rem <= 4) { MT = q; REM = rem <= 2 ? 2 : 4; }
    else { MT = (r + 15) / 16; REM = 0; }
}

int nnf_xty_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                 int r, int64_t ldu, float* out, int64_t ldo, hipStream_t st) {
    if (!ctx || !X || !Ut || !out || m < 1 || n < 1 || r < 1 || ldx < n || ldu < m || ldo < n) return NNF_ERR_ARG;
    if (r > NNF_MAX_RANK) {
        // ranks above 128: passes of <= 128 rank rows (the rows of the result are independent of each other); every pass takes
        // the workspace of the one before it (same stream)
        for (int k0 = 0; k0 < r; k0 += NNF_MAX_RANK) {
            const size_t mark = cur.off;
            const int rc = nnf_xty_impl(ctx, cur, X, m, n, ldx, Ut + (int64_t)k0 * ldu, r - k0 < NNF_MAX_RANK ? r - k0 : NNF_MAX_RANK,
                                        ldu, out + (int64_t)k0 * ldo, ldo, st);
            cur.off = mark;
            if (rc != NNF_OK) return rc;
        }
        return NNF_OK;
    }
    int MT, REM;
    split_rank(r, MT, REM);
    if (!x_vec_ok(X, ldx)) {
        MT = (r + 15) / 16;
        DISPATCH_MT(launch_xty, 0, false, ctx, cur, X, m, n, ldx, Ut, r, ldu, out, ldo, st)
    } else if (REM == 2) {
        DISPATCH_MT(launch_xty, 2, true, ctx, cur, X, m, n, ldx, Ut, r, ldu, out, ldo, st)
    } else if (REM == 4) {
        DISPATCH_MT(launch_xty, 4, true, ctx, cur, X, m, n, ldx, Ut, r, ldu, out, ldo, st)
    } else {
        DISPATCH_MT(launch_xty, 0, true, ctx, cur, X, m, n, ldx, Ut, r, ldu, out, ldo, st)
    }
}
extern "C" int nnf_xty_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int r,
                           int64_t ldu, float* out, int64_t ldo, void* stream) {
    if (!ctx) return NNF_ERR_ARG;
    nnf_ws_cursor cur(ctx);
    return nnf_xty_impl(ctx, cur, X, m, n, ldx, Ut, r, ldu, out, ldo, (hipStream_t)stream);
}

int nnf_xht_lds_launch(nnf_ctx* ctx, int MT, int REM, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V, int r,
                       int64_t ldv, float* out, int64_t ldo, hipStream_t st);   // k_xht_lds.hip
int nnf_xht_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V,
                 int r, int64_t ldv, float* out, int64_t ldo, hipStream_t st) {
    if (!ctx || !X || !V || !out || m < 1 || n < 1 || r < 1 || ldx < n || ldv < n || ldo < m) return NNF_ERR_ARG;
    if (r > NNF_MAX_RANK) {   // passes of <= 128 rank rows, as in nnf_xty_impl
        for (int k0 = 0; k0 < r; k0 += NNF_MAX_RANK) {
            const size_t mark = cur.off;
            const int rc = nnf_xht_impl(ctx, cur, X, m, n, ldx, V + (int64_t)k0 * ldv, r - k0 < NNF_MAX_RANK ? r - k0 : NNF_MAX_RANK,
                                        ldv, out + (int64_t)k0 * ldo, ldo, st);
            cur.off = mark;
            if (rc != NNF_OK) return rc;
        }
        return NNF_OK;
    }
    int MT, REM;
    split_rank(r, MT, REM);
    // ranks 51, 52: three tiles + four leftover ranks next to the 4-row-tile body do not fit 256 registers (84 bytes of scratch,
    // drains inside the chunk loop: 292 us against 238 us for the padded four tiles at 100000 x 2000, tools/probes/rank_step_probe.py)
    if (MT == 3 && REM == 4) { MT = 4; REM = 0; }
    // X staged through LDS in 256-byte row pieces (k_xht_lds.hip) where the load path, not the MFMA rate, bounds the product
    const char* pick = getenv("NNF_XHT");       // measurement knob: "direct" keeps the register-fragment kernel
    if (x_vec_ok(X, ldx) && MT + (REM > 0) <= 2 && !(pick && pick[0] == 'd'))
        return nnf_xht_lds_launch(ctx, MT, REM, X, m, n, ldx, V, r, ldv, out, ldo, st);
    if (!x_vec_ok(X, ldx)) {
        MT = (r + 15) / 16;
        DISPATCH_MT(launch_xht, 0, false, ctx, cur, X, m, n, ldx, V, r, ldv, out, ldo, st)
    } else if (REM == 2) {
        DISPATCH_MT(launch_xht, 2, true, ctx, cur, X, m, n, ldx, V, r, ldv, out, ldo, st)
    } else if (REM == 4) {
        DISPATCH_MT(launch_xht, 4, true, ctx, cur, X, m, n, ldx, V, r, ldv, out, ldo, st)
    } else {
        DISPATCH_MT(launch_xht, 0, true, ctx, cur, X, m, n, ldx, V, r, ldv, out, ldo, st)
    }
}
extern "C" int nnf_xht_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* V, int r,
                           int64_t ldv, float* out, int64_t ldo, void* stream) {
    if (!ctx) return NNF_ERR_ARG;
    nnf_ws_cursor cur(ctx);
    return nnf_xht_impl(ctx, cur, X, m, n, ldx, V, r, ldv, out, ldo, (hipStream_t)stream);
}

int nnf_gram_impl(nnf_ctx* ctx, nnf_ws_cursor& cur, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                  hipStream_t st, double* G64) {
    if (!ctx || !A || !G || r < 1 || K < 1 || lda < K || ldg < r) return NNF_ERR_ARG;
    if (r > NNF_MAX_RANK) return launch_gram_blocks(ctx, cur, A, r, K, lda, G, ldg, st, G64);
    switch ((r + 15) / 16) {
        case 1: return launch_gram<1>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        case 2: return launch_gram<2>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        case 3: return launch_gram<3>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        case 4: return launch_gram<4>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        case 5: return launch_gram<5>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        case 6: return launch_gram<6>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        case 7: return launch_gram<7>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
        default: return launch_gram<8>(ctx, cur, A, r, K, lda, G, ldg, st, G64);
    }
}
extern "C" int nnf_gram_f32(nnf_ctx* ctx, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                            void* stream) {
    if (!ctx) return NNF_ERR_ARG;
    nnf_ws_cursor cur(ctx);
    return nnf_gram_impl(ctx, cur, A, r, K, lda, G, ldg, (hipStream_t)stream, nullptr);
}
// The same Gram, and next to it the sums BEFORE they are rounded to fp32 (G64: r x r doubles, contiguous): the split-K slabs are
// added in fp64 anyway, so the copy costs a second store.  For the Gram-identity cost (nnf_nmf_gram_cost_g64_f32): fp32 storage of
// U^T U alone (relative rms 3.4e-8 per entry) bounds that cost's accuracy at ~1e-4 of a late-run cost at 10^6 x 4000 rank 100.
extern "C" int nnf_gram_f64_f32(nnf_ctx* ctx, const float* A, int r, int64_t K, int64_t lda, float* G, int64_t ldg,
                                double* G64, void* stream) {
    if (!ctx || !G64) return NNF_ERR_ARG;
    nnf_ws_cursor cur(ctx);
    return nnf_gram_impl(ctx, cur, A, r, K, lda, G, ldg, (hipStream_t)stream, G64);
}

template <int OP>
static int launch_cost(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                       const float* V, int64_t ldv, int r, float beta, double scale, double* out_f64, hipStream_t st,
                       const float* Vb = nullptr, int64_t ldvb = 0, int64_t nb = 1, const float* Ub = nullptr,
                       int64_t ldub = 0, int64_t nbu = 1, float* R1 = nullptr, float* R2 = nullptr, int64_t ldr = 0,
                       const float* Pin = nullptr, size_t ws_cap = 0) {
    const int grid = (int)nnf_cdiv(m, 128);
    // column splits: aim at ~8 workgroups per resident slot, keep at least 4 column blocks per workgroup
    const int nblk_all = (int)nnf_cdiv(n, 64);
    int csplit = (int)nnf_cdiv((int64_t)8 * 2 * ctx->num_cus, grid);
    if (csplit > nblk_all / 4) csplit = nblk_all / 4;
    // every column split stages the workgroup's 128 x r tile of U again: keep that re-read below ~5 % of the pass over X
    // (config B: 6 splits moved 1.02 GB for 0.82 GB algorithmic, PMC; the launch time is flat over 2..8 splits, so the
    // splits buy nothing there) -- as long as the grid still fills the resident slots twice over
    {
        int cap = (int)((0.05 * (double)n) / (double)(r > 0 ? r : 1));
        if (cap < 1) cap = 1;
        const int need = (int)nnf_cdiv((int64_t)2 * 3 * ctx->num_cus, grid);   // two rounds of 3 workgroups per CU
        if (cap < need) cap = need;
        if (Ub == nullptr && csplit > cap) csplit = cap;
    }
    {   // tuning knob (tools/cost_probe.py): NNF_COST_CSPLIT overrides the number of column splits
        static const int forced = [] { const char* e = getenv("NNF_COST_CSPLIT"); return e ? atoi(e) : 0; }();
        if (forced > 0) csplit = forced < nblk_all ? forced : nblk_all;
    }
    if (csplit < 1) csplit = 1;
    nnf_ws_cursor cur(ctx);
    if (ws_cap) cur.cap = ws_cap;          // (the tail of the workspace holds the model of the earlier rank chunks)
    double* partial = (double*)cur.take((size_t)grid * csplit * 8);
    if (!partial) return NNF_ERR_WORKSPACE;
    const int KS = (r + 3) / 4;
    const int64_t vf_total = (int64_t)nblk_all * KS * 64;
    f32x4* Vf = (f32x4*)cur.take((size_t)vf_total * 16);
    if (!Vf) return NNF_ERR_WORKSPACE;
    {
        int64_t pg = nnf_cdiv(vf_total, 256);
        if (pg > 1024) pg = 1024;
        hipLaunchKernelGGL(nnf_cost_prepv_kernel, dim3((int)pg), dim3(256), 0, st, V, ldv, r, n, KS, Vb, ldvb, nb, Vf, vf_total);
        NNF_CHECK_LAUNCH();
    }
    const int u_vec_ok = ((((uintptr_t)Ut) & 15) == 0 && (ldu & 3) == 0) ? 1 : 0;
    // two V buffers unless dropping one lets another workgroup onto the CU (ranks 53..64: a third, 77..104: a second; see the kernel)
    const size_t shm2 = (size_t)4 * 2 * KS * 64 * 4 + (size_t)2 * KS * 64 * 16 + 64, shm1 = shm2 - (size_t)KS * 64 * 16;
    const size_t lds_cu = 160 * 1024;
    auto wg_per_cu = [&](size_t b) { const size_t w = lds_cu / b; return w > 3 ? (size_t)3 : w; };   // (launch bound: 3)
    const int vdb = wg_per_cu(shm1) > wg_per_cu(shm2) ? 0 : 1;
    const size_t shm = vdb ? shm2 : shm1;
#define NNF_COST_LAUNCH(VV, NN, PP)                                                                                          \
    do {                                                                                                                     \
        if (shm > 48 * 1024)                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnf_cost_kernel<OP, VV, NN, PP>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                 \
        hipLaunchKernelGGL((nnf_cost_kernel<OP, VV, NN, PP>), dim3(grid, csplit), dim3(256), shm, st, X, m, n, ldx, Ut, ldu, Vf, r, \
                           beta, partial, Ub, ldub, nbu, R1, R2, ldr, u_vec_ok, vdb, Pin);                                   \
    } while (0)
    nnf_probe(ctx, NNF_PROBE_COST, 0, st);
    if (Pin != nullptr) {           // a later rank chunk of a rank above 128 (launch_cost_chunked): one instance per load width
        if (ldr < n) return NNF_ERR_ARG;
        if (x_vec_ok(X, ldx) && x_vec_ok(Pin, ldr)) NNF_COST_LAUNCH(true, 8, true);
        else NNF_COST_LAUNCH(false, 8, true);
    } else if (OP == NNF_PROD) {
        if (x_vec_ok(X, ldx)) NNF_COST_LAUNCH(true, 8, false);
        else NNF_COST_LAUNCH(false, 8, false);
    } else if (x_vec_ok(X, ldx)) {
        if (KS <= 16) NNF_COST_LAUNCH(true, 4, false);
        else NNF_COST_LAUNCH(true, 8, false);
    } else {
        if (KS <= 16) NNF_COST_LAUNCH(false, 4, false);
        else NNF_COST_LAUNCH(false, 8, false);
    }
#undef NNF_COST_LAUNCH
    NNF_CHECK_LAUNCH();
    nnf_probe(ctx, NNF_PROBE_COST, 1, st);
    if (OP == NNF_RATIO_KL || OP == NNF_RATIO_GEN || OP == NNF_PROD) return NNF_OK;   // nothing to sum
    hipLaunchKernelGGL(nnf_sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, (int64_t)grid * csplit, scale, out_f64);
    NNF_CHECK_LAUNCH();
    return NNF_OK;
}

static int cost_args_ok(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                        const float* V, int64_t ldv, int r, double* out) {
    if (!ctx || !X || !Ut || !V || !out || m < 1 || n < 1 || r < 1 || ldx < n || ldu < m || ldv < n) return NNF_ERR_ARG;
    if (32 * ldx * 4 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    return NNF_OK;
}

// The cost / ratio pass at any rank.  Up to NNF_MAX_RANK: one launch.  Above: the model U V is built up over rank chunks of
// <= 128 in an m x n buffer P -- chunk 0 writes its product (NNF_PROD), every later chunk adds its own to what it reads
// back, and the LAST chunk does so inside the pass that was asked for (cost terms or ratios on X and the whole model).
// P is the first output of a ratio pass (R1, in place), else the caller's scratch (nnf_ctx_set_scratch) or, if that is
// absent or too small, the tail of the context workspace; NNF_ERR_WORKSPACE when neither holds 4*m*ldp bytes.
template <int OP>
static int launch_cost_any_rank(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                                const float* V, int64_t ldv, int r, float beta, double scale, double* out_f64, hipStream_t st,
                                float* R1 = nullptr, float* R2 = nullptr, int64_t ldr = 0, const float* Ub = nullptr, int64_t ldub = 0,
                                int64_t nbu = 1) {
    if (r <= NNF_MAX_RANK)
        return launch_cost<OP>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, beta, scale, out_f64, st, nullptr, 0, 1, Ub, ldub, nbu, R1, R2, ldr);
    float* P = R1;
    int64_t ldp = ldr;
    size_t cap = 0;
    if (!P) {
        ldp = (n + 3) & ~(int64_t)3;
        const size_t need = (size_t)m * ldp * 4;
        if (ctx->big && ctx->big_bytes >= need) P = (float*)ctx->big;
        else {
            if (need + ((size_t)8 << 20) > ctx->ws_bytes) return NNF_ERR_WORKSPACE;
            cap = (ctx->ws_bytes - need) & ~(size_t)255;
            P = (float*)(ctx->ws + cap);
        }
    }
    if (ldp * 4 * 32 + 4 * (n + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    for (int k0 = 0; k0 < r; k0 += NNF_MAX_RANK) {
        const int rc = r - k0 < NNF_MAX_RANK ? r - k0 : NNF_MAX_RANK;
        const float* Uc = Ut + (int64_t)k0 * ldu;
        const float* Vc = V + (int64_t)k0 * ldv;
        const float* Ubc = Ub ? Ub + (int64_t)k0 * ldub : nullptr;      // (CP cost: the second factor of the Khatri-Rao rows)
        int e;
        if (k0 + rc < r)
            e = launch_cost<NNF_PROD>(ctx, X, m, n, ldx, Uc, ldu, Vc, ldv, rc, beta, scale, nullptr, st, nullptr, 0, 1, Ubc, ldub, nbu, P,
                                      nullptr, ldp, k0 ? P : nullptr, cap);
        else
            e = launch_cost<OP>(ctx, X, m, n, ldx, Uc, ldu, Vc, ldv, rc, beta, scale, out_f64, st, nullptr, 0, 1, Ubc, ldub, nbu, R1, R2,
                                ldp, P, cap);
        if (e != NNF_OK) return e;
    }
    return NNF_OK;
}

extern "C" int nnf_frob_resid_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                                  int64_t ldu, const float* V, int64_t ldv, int r, double* out_f64, void* stream) {
    const int rc = cost_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, out_f64);
    if (rc != NNF_OK) return rc;
    return launch_cost_any_rank<NNF_COST_FROB>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, 2.f, 1.0, out_f64, (hipStream_t)stream);
}

extern "C" int nnf_betadiv_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut,
                               int64_t ldu, const float* V, int64_t ldv, int r, double beta, double* out_f64,
                               void* stream) {
    const int rc = cost_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, out_f64);
    if (rc != NNF_OK) return rc;
    if (!(beta >= 0.0)) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (beta == 2.0)  // 1/2 ||X - UV||^2  (beta_divergence.py:51-52 with beta = 2)
        return launch_cost_any_rank<NNF_COST_FROB>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, 2.f, 0.5, out_f64, st);
    if (beta == 1.0) return launch_cost_any_rank<NNF_COST_KL>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, 1.f, 1.0, out_f64, st);
    if (beta == 0.0) return launch_cost_any_rank<NNF_COST_IS>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, 0.f, 1.0, out_f64, st);
    return launch_cost_any_rank<NNF_COST_GEN>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, (float)beta, 1.0, out_f64, st);
}

// Element-wise operands of mu_betadivmin (mu.py:84-97) for ranks beyond the fused kernels (64 < r <= 128):
//   R1 = X .* (U V)^(beta-2)   and, unless beta == 1,   R2 = (U V)^(beta-1),   both m x n with row stride ldr.
extern "C" int nnf_mu_ratio_f32(nnf_ctx* ctx, const float* X, int64_t m, int64_t n, int64_t ldx, const float* Ut, int64_t ldu,
                                const float* V, int64_t ldv, int r, double beta, float* R1, float* R2, int64_t ldr,
                                void* stream) {
    double dummy;
    const int rc = cost_args_ok(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, &dummy);
    if (rc != NNF_OK) return rc;
    if (!(beta >= 0.0) || !R1 || ldr < n || (beta != 1.0 && !R2)) return NNF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (beta == 1.0)
        return launch_cost_any_rank<NNF_RATIO_KL>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, 1.f, 1.0, nullptr, st, R1, R2, ldr);
    return launch_cost_any_rank<NNF_RATIO_GEN>(ctx, X, m, n, ldx, Ut, ldu, V, ldv, r, (float)beta, 1.0, nullptr, st, R1, R2, ldr);
}

// beta-divergence between a dense 3-way tensor and its CP model [[F0, F1, F2]]: the cost kernel on T seen as an
// I x (J*K) matrix with the right operand V[k][(j,kk)] = F1t[k][j] * F2t[k][kk] generated while it is staged.
// Replaces the cost lines of ntf.py:470-473 (the reference rebuilds the 500 x 250000 reconstruction explicitly).
extern "C" int nnf_cp3_betadiv_f32(nnf_ctx* ctx, const float* T, int64_t I, int64_t J, int64_t K, const float* Ft0,
                                   int64_t ld0, const float* Ft1, int64_t ld1, const float* Ft2, int64_t ld2, int R,
                                   double beta, double* out_f64, void* stream) {
    if (!ctx || !T || !Ft0 || !Ft1 || !Ft2 || !out_f64 || I < 1 || J < 1 || K < 1 || R < 1 || ld0 < I || ld1 < J ||
        ld2 < K || !(beta >= 0.0))
        return NNF_ERR_ARG;
    // T seen as an (I*J) x K matrix: row (i,j) of the left operand is F0[i,:].*F1[j,:] (generated while it is staged),
    // the right operand is F2^T as is.  I*J rows give the kernel its parallelism (one workgroup per 128 rows).
    // (ranks above 128: the model is built up over rank chunks in a tensor-sized buffer, launch_cost_any_rank)
    const int64_t m = I * J;
    if (32 * K * 4 + 4 * (K + 128) >= (int64_t)0x7fff0000) return NNF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define CP3(OP, B, SC) launch_cost_any_rank<OP>(ctx, T, m, K, K, Ft0, ld0, Ft2, ld2, R, B, SC, out_f64, st, nullptr, nullptr, 0, Ft1, ld1, J)
    if (beta == 2.0) return CP3(NNF_COST_FROB, 2.f, 0.5);
    if (beta == 1.0) return CP3(NNF_COST_KL, 1.f, 1.0);
    if (beta == 0.0) return CP3(NNF_COST_IS, 0.f, 1.0);
    return CP3(NNF_COST_GEN, (float)beta, 1.0);
#undef CP3
}
