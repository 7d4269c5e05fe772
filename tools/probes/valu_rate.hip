// Issue rate of the fp32 FMA forms the sweep kernels could use, per SIMD, with 1, 2 and 4 waves per SIMD (gfx950):
//   0 v_pk_fma_f32 v,v,v        1 v_pk_fma_f32 with an SGPR pair as src0     2 v_fma_f32 v,s,v     3 v_fma_f32 v,v,v
//   4 v_fmac_f32 v,s,v (VOP2)   5 v_pk_fma_f32 with an SGPR pair as src1
// 8 independent accumulator chains per lane, N instructions per wave; prints cycles per instruction per SIMD (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(1024) void k(int iters, float* out, float s0, float s1) {
    float2 a[8], b;
    for (int j = 0; j < 8; ++j) a[j] = make_float2(threadIdx.x * 1e-3f + j, 1.f - j);
    b = make_float2(1.0001f, 0.9999f);
    const float ss0 = __builtin_amdgcn_readfirstlane(s0), ss1 = __builtin_amdgcn_readfirstlane(s1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(b));
                if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "s"(make_float2(ss0, ss1)), "v"(b));
                if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j].x) : "s"(ss0), "v"(b.x));
                if constexpr (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j].x) : "v"(b.y), "v"(b.x));
                if constexpr (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j].x) : "s"(ss0), "v"(b.x));
                if constexpr (KIND == 5) asm volatile("v_pk_fma_f32 %0, %2, %1, %0" : "+v"(a[j]) : "s"(make_float2(ss0, ss1)), "v"(b));
            }
        }
    }
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += a[j].x + a[j].y;
    if (r == 123.456f) out[0] = r;
}
template <int KIND>
void run(const char* name, float* d) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, iters, d, 1.0001f, 0.9999f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, iters, d, 1.0001f, 0.9999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per = ms / 5 * 1e-3 * 2.4e9 / ((double)iters * 64 * wps);
        printf("%-34s %d wave(s)/SIMD: %6.2f cycles per instruction per SIMD\n", name, wps, per);
    }
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_pk_fma_f32 v,v,v", d);
    run<1>("v_pk_fma_f32 s[pair],v,v", d);
    run<5>("v_pk_fma_f32 v,s[pair],v", d);
    run<3>("v_fma_f32 v,v,v", d);
    run<2>("v_fma_f32 s,v,v", d);
    run<4>("v_fmac_f32 s,v (VOP2)", d);
    return 0;
}
