// Cycles per instruction (s_memtime) of the cross-lane moves the MFMA sweep kernel (k_hals_mfma.hip) is built from, alone and
// next to fp32 MFMAs: v_mov, v_permlane16_swap, v_permlane32_swap, ds_bpermute, DPP row moves; 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/permlane_rate.hip -o /tmp/permlane_rate && /tmp/permlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    float a = threadIdx.x, b = threadIdx.x * 2.f, c = 1.f, d = 3.f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %2, %3" : "=v"(a), "+v"(b), "=v"(c), "+v"(d));
            if (MODE == 1) asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 2) asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 3) asm volatile("ds_bpermute_b32 %0, %1, %2\n\tds_bpermute_b32 %3, %1, %4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "+v"(b), "+v"(c), "=&v"(d) , "+v"(acc0[0]));
            if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %1, %0\n\tv_fma_f32 %2, %3, %3, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MODE == 6)   // 4 MFMAs
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n\tv_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n\tv_mfma_f32_16x16x4_f32 %3, %4, %5, %3"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b));
            if (MODE == 7)   // 4 MFMAs each followed by 2 v_mov
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %6\n\tv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %6\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %6\n\tv_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %6"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b), "v"(c), "v"(d));
            if (MODE == 8)   // 4 MFMAs each followed by 2 v_fma
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\tv_fma_f32 %6, %7, %7, %6\n\tv_fma_f32 %7, %6, %6, %7\n\tv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n\tv_fma_f32 %6, %7, %7, %6\n\tv_fma_f32 %7, %6, %6, %7\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n\tv_fma_f32 %6, %7, %7, %6\n\tv_fma_f32 %7, %6, %6, %7\n\tv_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n\tv_fma_f32 %6, %7, %7, %6\n\tv_fma_f32 %7, %6, %6, %7"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(c), "+v"(d) : "v"(a), "v"(b));
            if (MODE == 9)   // 4 MFMAs each followed by 2 permlane16_swap
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n\tv_permlane16_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %6, %7\n\tv_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n\tv_permlane16_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %6, %7\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n\tv_permlane16_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %6, %7\n\tv_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n\tv_permlane16_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %6, %7"
                             : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(c), "+v"(d) : "v"(a), "v"(b));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + acc0[0] + acc1[1] + acc2[2] + acc3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_rep, int wgs_per_cu, float* out, long long* cyc) {
    const int iters = 200, nb = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x; s /= nb;
    const double n = (double)iters * REP * per_rep;
    printf("%-44s %d waves/SIMD: %7.2f memtime-ticks per instr per wave (%.2f per SIMD), wall %.3f ms -> %.2f ns per instr per SIMD\n", name, wgs_per_cu, s / n,
           s / n / wgs_per_cu, ms, ms * 1e6 / n / wgs_per_cu);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_mov_b32 x2", 2, w, out, cyc);
        run<1>("v_permlane16_swap x2", 2, w, out, cyc);
        run<2>("v_permlane32_swap x2", 2, w, out, cyc);
        run<3>("ds_bpermute_b32 x2 + wait", 2, w, out, cyc);
        run<4>("v_mov_dpp row_shr x2", 2, w, out, cyc);
        run<5>("v_fma_f32 x2", 2, w, out, cyc);
        run<6>("mfma16x16x4f32 x4", 4, w, out, cyc);
        run<7>("(mfma + 2 v_mov) x4  [per mfma]", 4, w, out, cyc);
        run<8>("(mfma + 2 v_fma) x4  [per mfma]", 4, w, out, cyc);
        run<9>("(mfma + swap16 + swap32) x4  [per mfma]", 4, w, out, cyc);
    }
    return 0;
}
