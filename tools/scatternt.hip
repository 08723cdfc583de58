// scatternt.hip -- the rank store of the suffix sort's round 0 (ISA[sa] = rank: 64 Mi random 4-byte stores, every address once) with plain and with
// non-temporal stores; addresses from a bijective bit mix of the index (no host-side permutation).
//   hipcc --offload-arch=gfx950 -O3 tools/scatternt.hip -o tools/_bin/scatternt && tools/_bin/scatternt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix26(uint32_t x)          // a bijection of 26-bit values
{
    const uint32_t M = (1u << 26) - 1u;
    x = (x * 0x9E3779B1u) & M; x ^= x >> 13; x = (x * 0x85EBCA6Bu) & M; x ^= x >> 11; x = (x * 0xC2B2AE35u) & M; x ^= x >> 15;
    return x;
}

template <int MODE, int ITEMS>      // 0 plain, 1 non-temporal
__global__ __launch_bounds__(256) void k_scatter(uint32_t *__restrict__ tab)
{
    const uint32_t base = blockIdx.x * (256 * ITEMS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = base + k * 256, a = mix26(i);
        if (MODE == 1) __builtin_nontemporal_store(i, tab + a);
        else tab[a] = i;
    }
}

__global__ __launch_bounds__(256) void k_check(const uint32_t *__restrict__ tab, uint32_t *bad)
{
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    if (mix26(tab[a]) != a) atomicAdd(bad, 1u);
}

int main()
{
    const uint32_t m = 64u << 20;
    uint32_t *tab, *bad;
    CK(hipMalloc(&tab, (size_t)m * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *what, auto launch) -> int {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        hipLaunchKernelGGL(k_check, dim3(m / 256), dim3(256), 0, 0, tab, bad);
        uint32_t hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("%-44s %.3f ms (%.1f G/s)%s\n", what, ms, m / ms / 1e6, hb ? "  WRONG" : "");
        return 0;
    };
    if (time_it("plain stores, 4 per thread", [&] { hipLaunchKernelGGL((k_scatter<0, 4>), dim3(m / 1024), dim3(256), 0, 0, tab); })) return 1;
    if (time_it("plain stores, 16 per thread", [&] { hipLaunchKernelGGL((k_scatter<0, 16>), dim3(m / 4096), dim3(256), 0, 0, tab); })) return 1;
    if (time_it("non-temporal stores, 4 per thread", [&] { hipLaunchKernelGGL((k_scatter<1, 4>), dim3(m / 1024), dim3(256), 0, 0, tab); })) return 1;
    if (time_it("non-temporal stores, 16 per thread", [&] { hipLaunchKernelGGL((k_scatter<1, 16>), dim3(m / 4096), dim3(256), 0, 0, tab); })) return 1;
    return 0;
}
