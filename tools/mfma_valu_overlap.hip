// Do fp32 MFMA and (packed) fp32 VALU work from two different waves of one SIMD run side by side on gfx950?
// Each workgroup = 8 waves on one CU... here: grid = 256 CUs x 1 workgroup of 512 threads (8 waves = 2 per SIMD).
// Waves 0..3 (one per SIMD) run kind A, waves 4..7 kind B.  kinds: 0 idle, 1 MFMA 32x32x2 f32 (4 independent accumulators),
// 2 v_pk_fma_f32 (8 independent chains), 3 plain v_fma_f32 (8 chains).  Prints the time of every pairing.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float work(int kind, int iters, float seed) {
    float r = 0.f;
    if (kind == 1) {
        f32x16 a = {0}, b = {0}, c = {0}, d = {0};
        for (int i = 0; i < iters; ++i) {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, b, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, c, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, d, 0, 0, 0);
        }
        r = a[0] + b[1] + c[2] + d[3];
    } else if (kind == 2) {
        f32x2 x[8];
        for (int j = 0; j < 8; ++j) x[j] = f32x2{seed + j, seed - j};
        const f32x2 m = {1.0001f, 0.9999f}, ad = {1e-3f, -1e-3f};
        for (int i = 0; i < iters * 16; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], m, ad);
        for (int j = 0; j < 8; ++j) r += x[j][0] + x[j][1];
    } else if (kind == 3) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = seed + j;
        for (int i = 0; i < iters * 16; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], 1.0001f, 1e-3f);
        for (int j = 0; j < 8; ++j) r += x[j];
    }
    return r;
}
__global__ __launch_bounds__(512) void k(int kindA, int kindB, int iters, float* out) {
    const int w = threadIdx.x >> 6;
    const float r = work(w < 4 ? kindA : kindB, iters, (float)threadIdx.x * 1e-3f);
    if (r == 123.456f) out[0] = r;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* nm[4] = {"idle", "mfma32x32x2", "pk_fma", "fma"};
    const int iters = 2000;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, a, b, iters, d);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, a, b, iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-12s | %-12s : %8.1f us\n", nm[a], nm[b], ms * 200.f);
        }
    return 0;
}
