// scattertest.hip -- what a random 4-byte scatter of 64 Mi (address, value) pairs costs on MI355X when the pairs arrive
// (a) in random order, (b) grouped into bins of 2^k consecutive addresses (random inside a bin): the second is what a
// binning pass in front of the scatter would buy.   hipcc --offload-arch=gfx950 -O3 tools/scattertest.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ITEMS>
__global__ __launch_bounds__(256) void k_apply(const uint2 *__restrict__ pairs, uint32_t *__restrict__ tab, uint32_t n)
{
    const uint32_t base = blockIdx.x * (256 * ITEMS) + threadIdx.x;
    uint2 p[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { const uint32_t i = base + k * 256; p[k] = i < n ? pairs[i] : make_uint2(0xffffffffu, 0); }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) if (p[k].x != 0xffffffffu) tab[p[k].x] = p[k].y;
}

int main()
{
    const uint32_t m = 64u << 20;
    uint2 *pairs; uint32_t *tab;
    CK(hipMalloc(&pairs, (size_t)m * 8)); CK(hipMalloc(&tab, (size_t)m * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto fn, int reps) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    std::vector<uint2> h(m);
    // a random permutation of the addresses (every address written once, like ISA[sa[i]] = rank)
    std::vector<uint32_t> perm(m);
    for (uint32_t i = 0; i < m; i++) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = m - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; uint32_t j = (uint32_t)(s % (i + 1)); std::swap(perm[i], perm[j]); }
    for (int shift : {0, 14, 16, 18, 20, 22}) {
        for (uint32_t i = 0; i < m; i++) h[i] = make_uint2(perm[i], i);
        if (shift) std::stable_sort(h.begin(), h.end(), [shift](const uint2 &a, const uint2 &b) { return (a.x >> shift) < (b.x >> shift); });
        CK(hipMemcpy(pairs, h.data(), (size_t)m * 8, hipMemcpyHostToDevice));
        float t4 = time([&] { hipLaunchKernelGGL(k_apply<4>, dim3(m / 1024), dim3(256), 0, 0, pairs, tab, m); }, 3);
        float t8 = time([&] { hipLaunchKernelGGL(k_apply<8>, dim3(m / 2048), dim3(256), 0, 0, pairs, tab, m); }, 3);
        if (shift) printf("bins of %8u addresses (%5u KiB): items4 %.3f ms (%.1f G/s)  items8 %.3f ms (%.1f G/s)\n", 1u << shift, (4u << shift) >> 10, t4, m / t4 / 1e6, t8, m / t8 / 1e6);
        else printf("random order                          : items4 %.3f ms (%.1f G/s)  items8 %.3f ms (%.1f G/s)\n", t4, m / t4 / 1e6, t8, m / t8 / 1e6);
    }
    return 0;
}
