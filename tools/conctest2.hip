// conctest2.hip -- do single-wave workgroups of concurrently running kernels share SIMDs?  Each wave runs a fixed dependent
// VALU chain; we report the wall time per configuration and, per wave, which CU / SIMD it ran on.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <map>
#include <vector>
__global__ void chain(int iters, unsigned *out, unsigned *where, int lds_bytes_unused)
{
    extern __shared__ unsigned dyn[];
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = x * 3u + 1u;
    }
    if (x == 0xFFFFFFFFu) out[0] = x + dyn[0];
    if (threadIdx.x == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[blockIdx.x] = (hwid & 0xFFFFFFu) | ((xcc & 0xFu) << 24);
    }
}
int main(int argc, char **argv)
{
    const int wgs = 65, iters = 40000;
    const int lds = argc > 1 ? atoi(argv[1]) : 0;
    unsigned *d; hipMalloc(&d, 4);
    for (int S : {1, 4, 8, 16}) {
        hipStream_t st[32];
        unsigned *w[32];
        for (int i = 0; i < S; i++) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipMalloc(&w[i], wgs * 4); }
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < S; i++) hipLaunchKernelGGL(chain, dim3(wgs), dim3(64), lds, st[i], iters, d, w[i], lds);
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        // histogram: waves per (xcc, se, cu, simd)
        std::map<unsigned, int> per_simd, per_cu;
        std::vector<unsigned> h(wgs);
        for (int i = 0; i < S; i++) {
            hipMemcpy(h.data(), w[i], wgs * 4, hipMemcpyDeviceToHost);
            for (unsigned v : h) {
                // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]...
                const unsigned simd = (v >> 4) & 3, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 24) & 15;
                const unsigned cukey = (xcc << 16) | (se << 8) | (sh << 4) | cu;
                per_cu[cukey]++; per_simd[(cukey << 2) | simd]++;
            }
        }
        int maxs = 0, maxc = 0;
        for (auto &p : per_simd) if (p.second > maxs) maxs = p.second;
        for (auto &p : per_cu) if (p.second > maxc) maxc = p.second;
        printf("lds %6d  %2d kernels x %d waves: %.2f ms; distinct CUs %zu (max waves on one CU %d), distinct SIMDs %zu (max on one SIMD %d)\n", lds, S, wgs, ms,
               per_cu.size(), maxc, per_simd.size(), maxs);
        for (int i = 0; i < S; i++) { hipStreamDestroy(st[i]); hipFree(w[i]); }
    }
    return 0;
}
