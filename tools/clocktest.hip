// clocktest.hip -- how fast does a latency-bound kernel (a few waves, dependent integer chain) run alone vs
// next to a busy grid?  build: hipcc --offload-arch=gfx950 -O3 tools/clocktest.hip -o /tmp/clocktest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void chain(uint32_t *out, int iters, uint64_t *clk)
{
    uint32_t x = threadIdx.x * 2654435761u + 12345u;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        x = __umulhi(x | 0x80000001u, 0x9E3779B9u) + (x >> 3) + i;     // dependent: mulhi + shift + 2 adds
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ void busy(uint32_t *out, int iters)
{
    uint32_t x = threadIdx.x + blockIdx.x, y = x * 3, z = x * 5, w = x * 7;
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; y = y * 22695477u + 1u; z ^= x + y; w += z * 3u; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
}

int main()
{
    uint32_t *o1, *o2; uint64_t *clk, h[2];
    hipMalloc(&o1, 1 << 20); hipMalloc(&o2, 64 << 20); hipMalloc(&clk, 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000000;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 1) hipLaunchKernelGGL(busy, dim3(256 * 8), dim3(256), 0, s2, o2, 3000000);
            if (mode == 2) hipLaunchKernelGGL(busy, dim3(32), dim3(256), 0, s2, o2, 3000000);
            hipEventRecord(a, s1);
            hipLaunchKernelGGL(chain, dim3(7), dim3(64), 0, s1, o1, iters, clk);
            hipEventRecord(b, s1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("mode %d (%s) rep %d: %.2f ms, %.1f ns/iter, shader clock %.0f MHz, %.1f cycles/iter\n", mode,
                   mode == 0 ? "alone" : mode == 1 ? "next to 2048 busy blocks" : "next to 32 busy blocks", rep, ms, ms * 1e6 / iters,
                   (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / iters);
        }
    }
    return 0;
}
