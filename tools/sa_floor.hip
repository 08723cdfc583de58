// sa_floor.hip -- the mandatory memory accesses of the suffix sort as it stands (bwt_fwd.hip), with no sorting at all: what the
// forward BWT of one block would cost if every kernel did nothing but the loads and stores the algorithm cannot avoid.
//   round 0     7 streaming radix passes over n (u64 key, u32 suffix) pairs: histogram read 8 B, scatter read 12 B + write 12 B
//               (written sequentially here: the floor of a pass is a copy), pass 0 reads the text instead of a key array;
//               one random 4-byte ISA store per suffix;
//   round r>=1  per unresolved suffix (counts from jpk_stats.sa_round_active of the bench block, or argv): one random 4-byte ISA
//               load (key2) and one random 4-byte ISA store (new rank), each beside its sequential list traffic
//               (a_sa 4 B + a_grp 4 B in, k2 4 B out; sa 4 B + rank 4 B in / out);
//   BWT bytes   one random 1-byte load T[SA[i] - 1] per suffix, one sequential byte store;
//   image       2 B per byte streaming.
// Random = a true random permutation of [0, n) (suffix numbers in SA order are one).  Prints ms per component and the total.
//   hipcc --offload-arch=gfx950 -O3 tools/sa_floor.hip -o tools/_bin/sa_floor;  tools/_bin/sa_floor [n m1 m2 ...]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int TB = 256, IT = 8;
__global__ __launch_bounds__(TB) void k_read8(const uint64_t *__restrict__ a, uint32_t *__restrict__ sink, uint32_t n)
{
    uint32_t base = blockIdx.x * (TB * IT) + threadIdx.x;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < IT; k++) { uint32_t i = base + k * TB; if (i < n) acc += a[i]; }
    if (acc == 0x1234567812345678ull) sink[0] = 1;
}
__global__ __launch_bounds__(TB) void k_copy12(const uint64_t *__restrict__ ka, const uint32_t *__restrict__ va, uint64_t *__restrict__ kb, uint32_t *__restrict__ vb, uint32_t n)
{
    uint32_t base = blockIdx.x * (TB * IT) + threadIdx.x;
    uint64_t k[IT]; uint32_t v[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; k[j] = i < n ? ka[i] : 0; v[j] = i < n ? va[i] : 0; }
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; if (i < n) { kb[i] = k[j]; vb[i] = v[j]; } }
}
// sequential (sa, grp) in, random ISA load, sequential k2 out
__global__ __launch_bounds__(TB) void k_gather(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ grp, const uint32_t *__restrict__ isa, uint32_t *__restrict__ k2, uint32_t m)
{
    uint32_t base = blockIdx.x * (TB * IT) + threadIdx.x;
    uint32_t s[IT], g[IT], v[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; s[j] = i < m ? sa[i] : 0; g[j] = i < m ? grp[i] : 0; }
#pragma unroll
    for (int j = 0; j < IT; j++) v[j] = isa[s[j]];
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; if (i < m) k2[i] = v[j] + (g[j] & 1u); }
}
// sequential (sa, rank) in, random ISA store, sequential (sa, rank) out
__global__ __launch_bounds__(TB) void k_scatter(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ rk, uint32_t *__restrict__ isa, uint32_t *__restrict__ sa2, uint32_t *__restrict__ rk2, uint32_t m)
{
    uint32_t base = blockIdx.x * (TB * IT) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; if (i < m) { uint32_t s = sa[i], r = rk[i]; isa[s] = r; if (sa2) { sa2[i] = s; rk2[i] = r; } } }
}
__global__ __launch_bounds__(TB) void k_bwt(const uint32_t *__restrict__ sa, const uint8_t *__restrict__ T, uint8_t *__restrict__ out, uint32_t n)
{
    uint32_t base = blockIdx.x * (TB * IT) + threadIdx.x;
    uint32_t s[IT]; uint8_t v[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; s[j] = i < n ? sa[i] : 1; }
#pragma unroll
    for (int j = 0; j < IT; j++) v[j] = T[s[j] ? s[j] - 1 : 0];
#pragma unroll
    for (int j = 0; j < IT; j++) { uint32_t i = base + j * TB; if (i < n) out[i] = v[j]; }
}
__global__ __launch_bounds__(TB) void k_copy1(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n16) { size_t i = (size_t)blockIdx.x * TB + threadIdx.x; if (i < n16) b[i] = a[i]; }

int main(int argc, char **argv)
{
    uint32_t n = 67108800u;
    std::vector<uint32_t> rounds = {47350511u, 8839248u, 3530u};        // enwik8-like first 64 MiB block, 11-byte packed keys (round 4 sa_rounds; 7-byte keys: 62345314 34623805 1997057 6)
    if (argc > 1) { n = (uint32_t)strtoul(argv[1], 0, 10); rounds.clear(); for (int i = 2; i < argc; i++) rounds.push_back((uint32_t)strtoul(argv[i], 0, 10)); }
    uint64_t *ka, *kb; uint32_t *va, *vb, *isa, *perm, *grp, *k2; uint8_t *T, *bw;
    CK(hipMalloc(&ka, (size_t)n * 8)); CK(hipMalloc(&kb, (size_t)n * 8)); CK(hipMalloc(&va, (size_t)n * 4)); CK(hipMalloc(&vb, (size_t)n * 4));
    CK(hipMalloc(&isa, (size_t)n * 4)); CK(hipMalloc(&perm, (size_t)n * 4)); CK(hipMalloc(&grp, (size_t)n * 4)); CK(hipMalloc(&k2, (size_t)n * 4));
    CK(hipMalloc(&T, (size_t)n + 64)); CK(hipMalloc(&bw, (size_t)n + 64));
    CK(hipMemset(ka, 1, (size_t)n * 8)); CK(hipMemset(va, 1, (size_t)n * 4)); CK(hipMemset(grp, 0, (size_t)n * 4)); CK(hipMemset(T, 65, (size_t)n + 64));
    {
        std::vector<uint32_t> h(n);
        for (uint32_t i = 0; i < n; i++) h[i] = i;
        uint64_t s = 88172645463325252ull;
        for (uint32_t i = n - 1; i > 0; i--) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; uint32_t j = (uint32_t)(s % (i + 1)); uint32_t t = h[i]; h[i] = h[j]; h[j] = t; }
        CK(hipMemcpy(perm, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto fn, int reps) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    auto grid = [](uint32_t m) { return dim3((m + TB * IT - 1) / (TB * IT)); };
    const int R = 5;
    double total = 0;
    float hist = time([&] { hipLaunchKernelGGL(k_read8, grid(n), dim3(TB), 0, 0, ka, k2, n); }, R);
    float scat = time([&] { hipLaunchKernelGGL(k_copy12, grid(n), dim3(TB), 0, 0, ka, va, kb, vb, n); }, R);
    printf("n = %u\n", n);
    printf("round 0: radix pass floor   hist read 8 B/pair %.3f ms (%.2f TB/s) + pair copy 24 B/pair %.3f ms (%.2f TB/s)  x 7 (6 hist) = %.3f ms\n", hist,
           8.0 * n / hist / 1e9, scat, 24.0 * n / scat / 1e9, 6 * hist + 7 * scat);
    total += 6 * hist + 7 * scat;
    float s0 = time([&] { hipLaunchKernelGGL(k_scatter, grid(n), dim3(TB), 0, 0, perm, va, isa, (uint32_t *)nullptr, (uint32_t *)nullptr, n); }, R);
    printf("round 0: ISA store per suffix (random 4 B)  %.3f ms (%.1f G/s)\n", s0, n / s0 / 1e6);
    total += s0;
    double rsum = 0;
    uint64_t acc = 0;
    for (size_t r = 0; r < rounds.size(); r++) {
        const uint32_t m = rounds[r];
        if (!m) continue;
        float g = time([&] { hipLaunchKernelGGL(k_gather, grid(m), dim3(TB), 0, 0, perm, grp, isa, k2, m); }, R);
        float s = time([&] { hipLaunchKernelGGL(k_scatter, grid(m), dim3(TB), 0, 0, perm, k2, isa, vb, va, m); }, R);
        printf("round %zu: %9u unresolved: key2 gather %.3f ms (%.1f G/s)  rank scatter %.3f ms (%.1f G/s)\n", r + 1, m, g, m / g / 1e6, s, m / s / 1e6);
        rsum += g + s;
        acc += m;
    }
    total += rsum;
    float bb = time([&] { hipLaunchKernelGGL(k_bwt, grid(n), dim3(TB), 0, 0, perm, T, bw, n); }, R);
    printf("(BWT bytes: a T[SA-1] gather, random 1 B from %u MiB, would be %.3f ms (%.1f G/s): not part of the design since round 3, the byte rides with the suffix)\n", n >> 20, bb, n / bb / 1e6);
    float im = time([&] { hipLaunchKernelGGL(k_copy1, dim3((n / 16 + TB - 1) / TB), dim3(TB), 0, 0, (const uint4 *)bw, (uint4 *)T, (size_t)n / 16); }, R);
    printf("BWT image: 2 B/B streaming %.3f ms\n", im);
    total += im;
    printf("random accesses: %.1f M (ISA %u + rounds 2 x %llu)\n", (n + 2.0 * acc) / 1e6, n, (unsigned long long)acc);
    printf("FLOOR of the current algorithm on this block: %.3f ms  (radix %.3f + random %.3f + image %.3f)\n", total, 6 * hist + 7 * scat, s0 + rsum, im);
    return 0;
}
