// Dev tool: component ablation of the W^T X kernel (generated from k_stream.hip by hand; not part of the library).

#include "../nn_fac_amd/csrc/k_stream_common.h"
#include <cstdio>
#include <vector>
template <int MT, int REM, bool VEC, int MODE>
__global__ __launch_bounds__(256, (MT + (REM > 0) <= 4 ? 2 : 1)) void xty_ab(const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx,
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

    f32x4 xb[4][4];  // [k-group t][k-step c]: row i_begin + 64q + 16t + 4g + c, columns jl..jl+3
    f32x4 areg[MTA];
    f32x4 ev[REM > 0 ? REM : 1];   // leftover rows: partial sums over this lane's rows, columns jl..jl+3
#pragma unroll
    for (int rr = 0; rr < (REM > 0 ? REM : 1); ++rr) ev[rr] = f32x4{0.f, 0.f, 0.f, 0.f};

    stageA_load<MTA>(Ut, ldu, r, i_end, i_begin, a_vec_ok, areg);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[t][c] = nnf_bload4<VEC>(rs, voff, (16 * t + c) * ldx4);
    stageA_store<MTA>(ldsA[0], areg);
    __syncthreads();

    for (int q = 0; q < nchunk; ++q) {
        const f32x4* img = ldsA[MODE == 3 ? 0 : (q & 1)];
        // next chunk's A tile: global loads now, LDS write after the MFMAs (rows past i_end come back as zeros)
        if constexpr (MODE != 3) stageA_load<MTA>(Ut, ldu, r, i_end, i_begin + 64 * (int64_t)(q + 1), a_vec_ok, areg);
        const int soff_next = (q + 1) * 64 * ldx4;
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
                    for (int cc = 0; cc < 4; ++cc) { if constexpr (MODE != 2) acc[mt][cc] = MFMA16(af[mt][c], xb[t][c][cc], acc[mt][cc]); else if (mt == 0 && cc == 0) acc[0][0][0] += af[mt][c] * xb[t][c][c]; }
            if constexpr (REM > 0) {
#pragma unroll
                for (int rr = 0; rr < REM; ++rr) {
                    const f32x4 uv = img[(MT * 4 + t) * 64 + 16 * g + rr];   // Ut[16MT+rr][row 16t+4g+c], c = 0..3
#pragma unroll
                    for (int c = 0; c < 4; ++c) ev[rr] = __builtin_elementwise_fma(f32x4{uv[c], uv[c], uv[c], uv[c]}, xb[t][c], ev[rr]);
                }
            }
            // refill the registers just consumed with the same rows of the next chunk (past the end: zeros)
#pragma unroll
            for (int c = 0; c < 4; ++c) { if constexpr (MODE != 1) xb[t][c] = nnf_bload4<VEC>(rs, voff, soff_next + (16 * t + c) * ldx4); else asm volatile("" : "+v"(xb[t][c])); }
        }
        if constexpr (MODE != 3) { stageA_store<MTA>(const_cast<f32x4*>(ldsA[(q + 1) & 1]), areg);
        __syncthreads(); }
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


__global__ __launch_bounds__(256, 2) void mfma_peak(float* out, int iters) {
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x, b = 1.0f + (float)blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] = MFMA16(a, b, acc[i]);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}
static float run_peak(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_peak, dim3(512), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mfma_peak, dim3(512), dim3(256), 0, 0, out, 100);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1000;
}
template <int MT, int REM, int MODE> float run(const float* X, const float* Ut, float* slabs, int64_t m, int64_t n, int r, int wgpc) {
    const int ncb = 8; const int64_t ldp = n; 
    int64_t nsplit = wgpc * 256 / ncb; int64_t rps = nnf_rup(nnf_cdiv(m, nsplit), 64); nsplit = nnf_cdiv(m, rps);
    const int grid = 8 * (int)nnf_cdiv(nsplit, 8) * ncb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((xty_ab<MT, REM, true, MODE>), dim3(grid), dim3(256), 0, 0, X, m, n, n, Ut, m, r, slabs, ldp, ncb, (int)nsplit, rps, 1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((xty_ab<MT, REM, true, MODE>), dim3(grid), dim3(256), 0, 0, X, m, n, n, Ut, m, r, slabs, ldp, ncb, (int)nsplit, rps, 1);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1000;
}
int main() {
    const int64_t m = 100000, n = 2000; const int r = 50;
    float *X, *Ut, *slabs;
    hipMalloc(&X, m * n * 4); hipMalloc(&Ut, (size_t)64 * m * 4); hipMalloc(&slabs, (size_t)64 * r * n * 4 * 2);
    std::vector<float> h(m * n); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    hipMemcpy(X, h.data(), m * n * 4, hipMemcpyHostToDevice); hipMemcpy(Ut, h.data(), (size_t)64 * m * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
    float tp = run_peak(slabs);
    printf("mfma peak loop (4800 MFMA/wave, 2 waves/SIMD): %.1f us = %.1f TF\n", tp, 512.0 * 4 * 4800 * 2048 / tp * 1e-6);
    printf("full 48+2         %.1f us\n", run<3, 2, 0>(X, Ut, slabs, m, n, r, 2));
    printf("no X loads 48+2   %.1f us\n", run<3, 2, 1>(X, Ut, slabs, m, n, r, 2));
    printf("no X loads 48+0   %.1f us\n", run<3, 0, 1>(X, Ut, slabs, m, n, 48, 2));
    printf("no X loads 64+0   %.1f us\n", run<4, 0, 1>(X, Ut, slabs, m, n, 64, 2));
    printf("full 64+0         %.1f us\n", run<4, 0, 0>(X, Ut, slabs, m, n, 64, 2));
    printf("no MFMA 48+2      %.1f us\n", run<3, 2, 2>(X, Ut, slabs, m, n, r, 2));
    }
    return 0;
}
