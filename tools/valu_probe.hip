// Dev tool: sustained rate of v_pk_fma_f32 vs v_fma_f32 with a scalar-register source, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void valu_probe(float* out, int iters, float s0, float s1) {
    f32x2 a[8];
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) { a[i] = f32x2{0.f, 0.f}; v[i] = f32x2{(float)threadIdx.x + i, 1.0f + i}; }
    f32x2 sp = {s0, s1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) {        // packed, SGPR-pair source: 4 flops per lane per instruction
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sp), "v"(v[i]));
                } else if constexpr (KIND == 1) {  // two plain FMAs, SGPR source
                    asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %4, %5, %1" : "+v"(a[i].x), "+v"(a[i].y) : "s"(s0), "v"(v[i].x), "s"(s1), "v"(v[i].y));
                } else if constexpr (KIND == 2) {  // packed, VGPR sources
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(v[(i + 1) & 7]), "v"(v[i]));
                } else {                           // v_fmac (VOP2) with SGPR source
                    asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %4, %5" : "+v"(a[i].x), "+v"(a[i].y) : "s"(s0), "v"(v[i].x), "s"(s1), "v"(v[i].y));
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 123.456f) out[0] = s;
}

int main() {
    float* out; hipMalloc(&out, 64);
    const int iters = 4000;
    const char* names[] = {"v_pk_fma_f32 sgpr-pair", "2 x v_fma_f32 sgpr", "v_pk_fma_f32 vgpr", "2 x v_fmac_f32 sgpr"};
    for (int kind = 0; kind < 4; ++kind)
        for (int wg = 1; wg <= 4; ++wg) {   // workgroups (4 waves) per CU = waves per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                switch (kind) {
                    case 0: hipLaunchKernelGGL(valu_probe<0>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); break;
                    case 1: hipLaunchKernelGGL(valu_probe<1>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); break;
                    case 2: hipLaunchKernelGGL(valu_probe<2>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); break;
                    default: hipLaunchKernelGGL(valu_probe<3>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * wg * 256 * iters * 32 * 4;   // 32 packed-equivalents per iteration, 4 flops each
            printf("%-24s waves/SIMD %d : %.3f ms  %.1f TFLOP/s  (%.2f ns per packed-equivalent per wave)\n", names[kind], wg, ms,
                   flops / ms * 1e-9, ms * 1e6 / (iters * 32.0));
        }
    return 0;
}
