// Layout probe for v_mfma_f32_32x32x2_f32 and v_permlane32_swap_b32 on gfx950 (used to design the MFMA-assisted sweep).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned* sw) {
    const int l = threadIdx.x;
    // A[i][k]: lane = i + 32k ; B[k][j]: lane = j + 32k
    const float a = (l < 32) ? (float)(l) : 0.f;        // A[i][0] = i, A[i][1] = 0
    const float b = (l < 32) ? 1.f : 0.f;               // B[0][j] = 1
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[v * 64 + l] = acc[v];
    // second product: D[i][j] = j  (A[i][0] = 1, B[0][j] = j)
    f32x16 acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((l < 32) ? 1.f : 0.f, (l < 32) ? (float)l : 0.f, acc2, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[1024 + v * 64 + l] = acc2[v];
    // k index check: A[i][1] = 100 + i, B[1][j] = 1 -> D = 100 + i when lanes 32..63 carry k = 1
    f32x16 acc3 = {0};
    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32((l >= 32) ? (float)(100 + l - 32) : 0.f, (l >= 32) ? 1.f : 0.f, acc3, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[2048 + v * 64 + l] = acc3[v];
    unsigned x = 1000 + l, y = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    sw[l] = r[0];
    sw[64 + l] = r[1];
}
int main() {
    float* d; unsigned* s;
    hipMalloc(&d, 3072 * 4); hipMalloc(&s, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
    float h[3072]; unsigned hs[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
    printf("row index i held by (vgpr v, lane 0) and (v, lane 32):\n");
    for (int v = 0; v < 16; ++v) printf("v%-2d: lane0 -> i=%g  lane32 -> i=%g | col: lane5 -> j=%g lane37 -> j=%g | k1: lane0 %g\n", v, h[v * 64], h[v * 64 + 32],
                                        h[1024 + v * 64 + 5], h[1024 + v * 64 + 37], h[2048 + v * 64]);
    printf("swap: r0 lanes 0,31,32,63 = %u %u %u %u ; r1 = %u %u %u %u\n", hs[0], hs[31], hs[32], hs[63], hs[64], hs[95], hs[96], hs[127]);
    return 0;
}
