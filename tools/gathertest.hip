// gathertest.hip -- random-access rates that bound the suffix-sort rounds (MI355X): 4-byte gathers / scatters through a
// random permutation, table sizes around the Infinity Cache (256 MiB).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_perm(uint32_t *idx, uint32_t n, uint32_t mul, uint32_t add) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) idx[i] = (uint32_t)(((uint64_t)i * mul + add) % n); }
template <int ITEMS>
__global__ __launch_bounds__(256) void k_gather(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ tab, uint32_t *__restrict__ out, uint32_t n)
{
    uint32_t base = blockIdx.x * (256 * ITEMS) + threadIdx.x;
    uint32_t ix[ITEMS], v[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { uint32_t i = base + k * 256; ix[k] = i < n ? idx[i] : 0; }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) v[k] = tab[ix[k]];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { uint32_t i = base + k * 256; if (i < n) out[i] = v[k]; }
}
template <int ITEMS>
__global__ __launch_bounds__(256) void k_scatter(const uint32_t *__restrict__ idx, uint32_t *__restrict__ tab, uint32_t n)
{
    uint32_t base = blockIdx.x * (256 * ITEMS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { uint32_t i = base + k * 256; if (i < n) tab[idx[i]] = i; }
}
__global__ __launch_bounds__(256) void k_gather_u8(const uint32_t *__restrict__ idx, const uint8_t *__restrict__ tab, uint8_t *__restrict__ out, uint32_t n)
{
    uint32_t base = blockIdx.x * (256 * 8) + threadIdx.x;
    uint32_t ix[8]; uint8_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { uint32_t i = base + k * 256; ix[k] = i < n ? idx[i] : 0; }
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = tab[ix[k]];
#pragma unroll
    for (int k = 0; k < 8; k++) { uint32_t i = base + k * 256; if (i < n) out[i] = v[k]; }
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n16) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n16) b[i] = a[i]; }

int main()
{
    const uint32_t m = 64u << 20;            // accesses per launch
    uint32_t *idx, *out, *tab;
    CK(hipMalloc(&idx, (size_t)m * 4)); CK(hipMalloc(&out, (size_t)m * 4)); CK(hipMalloc(&tab, (size_t)256 << 22));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto fn, int reps) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    {
        float ms = time([&] { hipLaunchKernelGGL(k_copy, dim3(m / 4 / 256), dim3(256), 0, 0, (const uint4 *)idx, (uint4 *)out, (size_t)m / 4); }, 5);
        printf("copy 256 MiB: %.3f ms = %.2f TB/s (r+w)\n", ms, 2.0 * m * 4 / ms / 1e9);
    }
    for (uint32_t tn : {16u << 20, 64u << 20, 128u << 20, 256u << 20}) {      // table entries (x4 bytes)
        // pseudo-random permutation-ish indices: i * odd mod tn
        hipLaunchKernelGGL(k_perm, dim3(m / 256), dim3(256), 0, 0, idx, m, 2654435761u, 12345u);
        // idx currently in [0,m); fold into table range
        if (tn != m) { hipLaunchKernelGGL(k_perm, dim3(m / 256), dim3(256), 0, 0, idx, m, 2654435761u, 12345u); }
        std::vector<uint32_t> h(m);
        CK(hipMemcpy(h.data(), idx, (size_t)m * 4, hipMemcpyDeviceToHost));
        uint64_t s = 88172645463325252ull;
        for (uint32_t i = 0; i < m; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % tn); }
        CK(hipMemcpy(idx, h.data(), (size_t)m * 4, hipMemcpyHostToDevice));
        float g4 = time([&] { hipLaunchKernelGGL(k_gather<4>, dim3(m / 1024), dim3(256), 0, 0, idx, tab, out, m); }, 3);
        float g8 = time([&] { hipLaunchKernelGGL(k_gather<8>, dim3(m / 2048), dim3(256), 0, 0, idx, tab, out, m); }, 3);
        float g16 = time([&] { hipLaunchKernelGGL(k_gather<16>, dim3(m / 4096), dim3(256), 0, 0, idx, tab, out, m); }, 3);
        float s4 = time([&] { hipLaunchKernelGGL(k_scatter<4>, dim3(m / 1024), dim3(256), 0, 0, idx, tab, m); }, 3);
        float s8 = time([&] { hipLaunchKernelGGL(k_scatter<8>, dim3(m / 2048), dim3(256), 0, 0, idx, tab, m); }, 3);
        printf("table %4u MiB: gather4B items4 %.3f ms (%.1f G/s) items8 %.3f (%.1f G/s) items16 %.3f (%.1f G/s) | scatter4B items4 %.3f ms (%.1f G/s) items8 %.3f (%.1f G/s)\n", tn >> 18,
               g4, m / g4 / 1e6, g8, m / g8 / 1e6, g16, m / g16 / 1e6, s4, m / s4 / 1e6, s8, m / s8 / 1e6);
        if (tn <= (64u << 20)) {
            float gb = time([&] { hipLaunchKernelGGL(k_gather_u8, dim3(m / 2048), dim3(256), 0, 0, idx, (const uint8_t *)tab, (uint8_t *)out, m); }, 3);
            printf("   byte table %u MiB: gather1B %.3f ms (%.1f G/s)\n", tn >> 20, gb, m / gb / 1e6);
        }
    }
    return 0;
}
