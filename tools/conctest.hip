// conctest.hip -- how many kernels from different HIP streams does the GPU run at the same time?  Each kernel is a few
// single-wave workgroups that spin for ~10 ms; total time = ceil(streams / concurrent kernels) * 10 ms.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
__global__ void spin(unsigned long long ticks, unsigned *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned x = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x++;
    if (x == 0xFFFFFFFFu) out[0] = x;
}
int main(int argc, char **argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 65;
    unsigned *d; hipMalloc(&d, 4);
    for (int S : {1, 2, 4, 8, 12, 16, 24, 32}) {
        hipStream_t st[32];
        for (int i = 0; i < S; i++) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < S; i++) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, st[i], 1000000ull /* 10 ms at 100 MHz */, d);
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%2d streams x %d single-wave workgroups spinning 10 ms: %.1f ms -> ~%.1f kernels at a time\n", S, wgs, ms, S * 10.0 / ms);
        for (int i = 0; i < S; i++) hipStreamDestroy(st[i]);
    }
    return 0;
}
