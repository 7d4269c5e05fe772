// Dev tool: latency of a batch of scalar-cache loads (2 x s_load_dwordx16 + s_waitcnt lgkmcnt(0)) as a function of the
// table footprint and of the number of resident waves -- the operand path of the HALS sweep kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NL>   // NL x16 loads per batch
__global__ __launch_bounds__(256) void smem_probe(const float* __restrict__ G, int table_bytes, int iters, unsigned long long* out,
                                                  float* sink) {
    const uint64_t base = (uint64_t)G;
    // every wave starts at a different phase of the table, like the sweep kernel's waves do after a while
    int off = (int)(((blockIdx.x * 4 + (threadIdx.x >> 6)) * 1664) % table_bytes) & ~63;
    off = __builtin_amdgcn_readfirstlane(off);
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f32x16 d0, d1, d2, d3;
        if constexpr (NL == 2) {
            asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %3 offset:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(d0), "=s"(d1) : "s"(base), "s"(off));
            acc += d0[0] + d1[15];
        } else {
            asm volatile("s_load_dwordx16 %0, %4, %5\n\ts_load_dwordx16 %1, %4, %5 offset:64\n\ts_load_dwordx16 %2, %4, %5 offset:128\n\t"
                         "s_load_dwordx16 %3, %4, %5 offset:192\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(d0), "=s"(d1), "=s"(d2), "=s"(d3) : "s"(base), "s"(off));
            acc += d0[0] + d1[15] + d2[3] + d3[7];
        }
        off += 64 * NL;
        if (off + 64 * NL > table_bytes) off = 0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    float* G; unsigned long long* out; float* sink;
    hipMalloc(&G, 1 << 20); hipMemset(G, 0, 1 << 20); hipMalloc(&out, 64); hipMalloc(&sink, 64);
    const int iters = 2000;
    const int tables[] = {1024, 4096, 8192, 11264, 13312, 16384, 32768, 53248, 262144};
    const int grids[] = {256, 512, 1024, 1536};   // 256 CUs: 1, 2, 4, 6 workgroups (4 waves each) per CU
    printf("s_memtime ticks per batch (100 MHz counter -> ns x10); wall ns per batch from events\n");
    for (int nl = 2; nl <= 4; nl += 2)
        for (int tb : tables)
            for (int grid : grids) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto launch = [&]() {
                    if (nl == 2) hipLaunchKernelGGL(smem_probe<2>, dim3(grid), dim3(256), 0, 0, G, tb, iters, out, sink);
                    else hipLaunchKernelGGL(smem_probe<4>, dim3(grid), dim3(256), 0, 0, G, tb, iters, out, sink);
                };
                launch(); hipDeviceSynchronize();
                hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("x16 loads/batch %d  table %6d B  waves/CU %2d : %.1f ns per batch\n", nl, tb, grid * 4 / 256, ms * 1e6 / iters);
            }
    return 0;
}
