// partcopy.hip -- what the WRITE PATTERN of a radix pass costs with no ranking work at all.  A pass of the one-pass radix sort (radix.hip) reads
// 64 Mi (key u64, value u32) pairs in tiles of 4096 and writes every tile as 256 runs (one per digit, 16 pairs on average) behind the runs the
// tiles in front of it wrote for the same digit.  Here the permutation inside a tile is the identity and the run lengths are given (the same in
// every tile), so a workgroup does nothing but load its tile and store it in runs -- the memory system sees the pass's addresses, the CUs none
// of its work:
//   copy      keys and values to the same place (one stream each)                        the plain copy of profiles/r05_sa_floor.txt
//   uniform   256 runs of 16 pairs per tile (keys: 128 B runs, aligned; values: 64 B)
//   skewed    run lengths drawn like text digits (a few long runs, many short ones; unaligned)
//   one       all 4096 pairs of a tile in ONE run (= the copy, through the same code)
// The store instructions have the shape of the sort's: a wave stores 64 consecutive tile slots per instruction (keys as 8 B, values as 4 B per lane).
//   hipcc --offload-arch=gfx950 -O3 tools/partcopy.hip -o tools/_bin/partcopy && tools/_bin/partcopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int TILE = 4096, TB = 256, ITEMS = TILE / TB;

// dig[i] = the run slot i of a tile belongs to, start[d] = first slot of run d, gbase[d] = where digit d's output begins, len[d] = its length per tile
template <int NT>      // 1: non-temporal stores, 2: non-temporal loads as well, 3: non-temporal loads only
__global__ __launch_bounds__(TB) void k_part(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                             const uint8_t *__restrict__ dig, const uint32_t *__restrict__ start, const uint64_t *__restrict__ gbase,
                                             const uint32_t *__restrict__ len, uint32_t kgrp)
{
    __shared__ uint32_t s_start[256], s_len[256];
    __shared__ uint64_t s_base[256];
    s_start[threadIdx.x] = start[threadIdx.x];
    s_len[threadIdx.x] = len[threadIdx.x];
    s_base[threadIdx.x] = gbase[threadIdx.x];
    __syncthreads();
    // kgrp > 1: workgroups b, b + 8, b + 16, .. (one XCD under round-robin placement) take kgrp CONSECUTIVE tiles of every window of 8 kgrp tiles,
    // so that the runs of neighbouring tiles -- which share cache lines -- meet in one L2
    uint32_t tile = blockIdx.x;
    if (kgrp > 1) { const uint32_t w = tile / (8 * kgrp), r = tile % (8 * kgrp); tile = w * 8 * kgrp + (r % 8) * kgrp + r / 8; }
    const size_t base = (size_t)tile * TILE;
    uint64_t k[ITEMS];
    uint32_t v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {                                   // slot = j * 256 + thread: a wave's instruction covers 64 consecutive slots
        if (NT >= 2) { k[j] = __builtin_nontemporal_load(kin + base + j * TB + threadIdx.x); v[j] = __builtin_nontemporal_load(vin + base + j * TB + threadIdx.x); }
        else { k[j] = kin[base + j * TB + threadIdx.x]; v[j] = vin[base + j * TB + threadIdx.x]; }
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t slot = j * TB + threadIdx.x;
        const uint32_t d = dig[slot];
        const size_t o = s_base[d] + (size_t)tile * s_len[d] + (slot - s_start[d]);
        if (NT == 1 || NT == 2) { __builtin_nontemporal_store(k[j], kout + o); __builtin_nontemporal_store(v[j], vout + o); }
        else { kout[o] = k[j]; vout[o] = v[j]; }
    }
}

// the same with key and value in ONE 12-byte record (array of structures): a run of 16 pairs is 192 bytes in one place instead of 128 + 64 in two,
// i.e. half as many partial lines where runs meet
struct Rec { uint32_t a, b, c; };
__global__ __launch_bounds__(TB) void k_part_aos(const Rec *__restrict__ rin, Rec *__restrict__ rout, const uint8_t *__restrict__ dig, const uint32_t *__restrict__ start,
                                                 const uint64_t *__restrict__ gbase, const uint32_t *__restrict__ len, uint32_t kgrp)
{
    __shared__ uint32_t s_start[256], s_len[256];
    __shared__ uint64_t s_base[256];
    s_start[threadIdx.x] = start[threadIdx.x];
    s_len[threadIdx.x] = len[threadIdx.x];
    s_base[threadIdx.x] = gbase[threadIdx.x];
    __syncthreads();
    uint32_t tile = blockIdx.x;
    if (kgrp > 1) { const uint32_t w = tile / (8 * kgrp), r = tile % (8 * kgrp); tile = w * 8 * kgrp + (r % 8) * kgrp + r / 8; }
    const size_t base = (size_t)tile * TILE;
    Rec r[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) r[j] = rin[base + j * TB + threadIdx.x];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t slot = j * TB + threadIdx.x;
        const uint32_t d = dig[slot];
        rout[s_base[d] + (size_t)tile * s_len[d] + (slot - s_start[d])] = r[j];
    }
}

__global__ __launch_bounds__(TB) void k_copy(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout, uint32_t *__restrict__ vout)
{
    const size_t base = (size_t)blockIdx.x * TILE;
    uint64_t k[ITEMS];
    uint32_t v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) { k[j] = kin[base + j * TB + threadIdx.x]; v[j] = vin[base + j * TB + threadIdx.x]; }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) { kout[base + j * TB + threadIdx.x] = k[j]; vout[base + j * TB + threadIdx.x] = v[j]; }
}

// how workgroups are dealt over the XCDs: cnt[blockIdx % 8][XCC_ID]
__global__ void k_census(uint32_t *cnt)
{
    if (threadIdx.x == 0) {
        const uint32_t x = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 15u;      // HW_REG_XCC_ID, bits 3:0
        atomicAdd(&cnt[(blockIdx.x % 8) * 16 + x], 1u);
    }
}

int main(int argc, char **argv)
{
    const uint32_t ntiles = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384;           // 16384 tiles = 64 Mi pairs = one 64 MiB block
    const size_t n = (size_t)ntiles * TILE;
    uint64_t *kin, *kout, *d_gbase;
    uint32_t *vin, *vout, *d_start, *d_len;
    uint8_t *d_dig;
    CK(hipMalloc(&kin, n * 12)); CK(hipMalloc(&kout, n * 12));      // (the 12-byte record form uses these two alone)
    CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
    CK(hipMalloc(&d_dig, TILE)); CK(hipMalloc(&d_start, 1024)); CK(hipMalloc(&d_len, 1024)); CK(hipMalloc(&d_gbase, 2048));
    CK(hipMemset(kin, 1, n * 12)); CK(hipMemset(vin, 2, n * 4)); CK(hipMemset(kout, 0, n * 12)); CK(hipMemset(vout, 0, n * 4));
    {
        uint32_t *d_c, h_c[128];
        CK(hipMalloc(&d_c, 512)); CK(hipMemset(d_c, 0, 512));
        hipLaunchKernelGGL(k_census, dim3(ntiles), dim3(TB), 0, 0, d_c);
        CK(hipMemcpy(h_c, d_c, 512, hipMemcpyDeviceToHost));
        uint32_t on = 0, tot = 0;
        for (int c = 0; c < 8; c++) { uint32_t mx = 0; for (int x = 0; x < 16; x++) { tot += h_c[c * 16 + x]; mx = std::max(mx, h_c[c * 16 + x]); } on += mx; }
        printf("# placement: %u of %u workgroups sit on the XCD most of their class (blockIdx mod 8) sits on\n", on, tot);
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = (double)n * 24.0;
    printf("# %u tiles of %d pairs (u64 key + u32 value): %.0f MB read + the same written per launch\n", ntiles, TILE, bytes / 2e6);
    auto time_it = [&](const char *what, auto launch) -> int {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 12; rep++) {
            CK(hipEventRecord(a, 0));
            launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2) { sum += ms; best = std::min(best, ms); }
        }
        printf("%-100s %7.1f us (best %7.1f)  = %5.2f TB/s\n", what, 100.f * sum, 1000.f * best, bytes / (sum / 10 * 1e-3) / 1e12);
        return 0;
    };
    if (time_it("copy: one stream of keys, one of values", [&] { hipLaunchKernelGGL(k_copy, dim3(ntiles), dim3(TB), 0, 0, kin, vin, kout, vout); })) return 1;
    bool only_first = false;
    auto pattern = [&](const char *what, const std::vector<uint32_t> &len) -> int {
        std::vector<uint32_t> start(256);
        std::vector<uint64_t> gbase(256);
        std::vector<uint8_t> dig(TILE);
        uint32_t s = 0;
        uint64_t g = 0;
        for (int d = 0; d < 256; d++) {
            start[d] = s; gbase[d] = g;
            for (uint32_t i = 0; i < len[d]; i++) dig[s + i] = (uint8_t)d;
            s += len[d]; g += (uint64_t)len[d] * ntiles;
        }
        if (s != TILE) { printf("bad pattern\n"); return 1; }
        CK(hipMemcpy(d_dig, dig.data(), TILE, hipMemcpyHostToDevice)); CK(hipMemcpy(d_start, start.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, len.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(d_gbase, gbase.data(), 2048, hipMemcpyHostToDevice));
        for (uint32_t kgrp : {1u, 2u, 4u, 8u, 32u, ntiles / 8}) {
            if (kgrp != 1 && ntiles % (8 * kgrp)) continue;
            char line[200]; snprintf(line, sizeof line, "%s%s%u", what, kgrp == 1 ? "" : "  | tiles per XCD and window: ", kgrp);
            if (kgrp == 1) snprintf(line, sizeof line, "%s", what);
            if (time_it(line, [&] { hipLaunchKernelGGL(k_part<0>, dim3(ntiles), dim3(TB), 0, 0, kin, vin, kout, vout, d_dig, d_start, d_gbase, d_len, kgrp); })) return 1;
            if (only_first) break;
        }
        {
            char line[200]; snprintf(line, sizeof line, "%s  | non-temporal stores", what);
            if (time_it(line, [&] { hipLaunchKernelGGL(k_part<1>, dim3(ntiles), dim3(TB), 0, 0, kin, vin, kout, vout, d_dig, d_start, d_gbase, d_len, 1u); })) return 1;
            snprintf(line, sizeof line, "%s  | non-temporal LOADS only", what);
            if (time_it(line, [&] { hipLaunchKernelGGL(k_part<3>, dim3(ntiles), dim3(TB), 0, 0, kin, vin, kout, vout, d_dig, d_start, d_gbase, d_len, 1u); })) return 1;
            snprintf(line, sizeof line, "%s  | non-temporal LOADS only, tiles per XCD and window: 8", what);
            if (time_it(line, [&] { hipLaunchKernelGGL(k_part<3>, dim3(ntiles), dim3(TB), 0, 0, kin, vin, kout, vout, d_dig, d_start, d_gbase, d_len, 8u); })) return 1;
            snprintf(line, sizeof line, "%s  | non-temporal stores and loads", what);
            if (time_it(line, [&] { hipLaunchKernelGGL(k_part<2>, dim3(ntiles), dim3(TB), 0, 0, kin, vin, kout, vout, d_dig, d_start, d_gbase, d_len, 1u); })) return 1;
        }
        for (uint32_t kgrp : {1u, 8u}) {
            char line[200]; snprintf(line, sizeof line, "%s  | 12-byte records%s", what, kgrp == 1 ? "" : ", tiles per XCD and window: 8");
            if (time_it(line, [&] { hipLaunchKernelGGL(k_part_aos, dim3(ntiles), dim3(TB), 0, 0, (const Rec *)kin, (Rec *)kout, d_dig, d_start, d_gbase, d_len, kgrp); })) return 1;
        }
        return 0;
    };
    {
        std::vector<uint32_t> len(256, 0); len[0] = TILE;
        if (pattern("one run per tile (the copy through the pattern's code)", len)) return 1;
    }
    for (int runs : {16, 64, 256}) {
        std::vector<uint32_t> len(256, 0);
        for (int d = 0; d < runs; d++) len[d] = TILE / runs;
        char what[128]; snprintf(what, sizeof what, "uniform: %d runs of %d pairs per tile (aligned)", runs, TILE / runs);
        if (pattern(what, len)) return 1;
    }
    {   // 256 runs, lengths 15 / 17 alternating: the same bytes per run on average, nothing aligned
        std::vector<uint32_t> len(256);
        for (int d = 0; d < 256; d++) len[d] = (d & 1) ? 17 : 15;
        if (pattern("256 runs of 15 / 17 pairs (unaligned)", len)) return 1;
    }
    for (int runs : {128, 64}) {   // fewer, longer unaligned runs: what a tile of 8192 / 16384 pairs would write per 4096 of them
        std::vector<uint32_t> len(256, 0);
        const uint32_t m = TILE / runs;
        for (int d = 0; d < runs; d++) len[d] = (d & 1) ? m + 1 : m - 1;
        char what[128]; snprintf(what, sizeof what, "%d runs of %u / %u pairs (unaligned)", runs, m - 1, m + 1);
        if (pattern(what, len)) return 1;
    }
    {   // text-like: geometric weights (digit d ~ 0.985^d), at least one pair per run
        std::vector<uint32_t> len(256, 1);
        double w[256], tot = 0;
        for (int d = 0; d < 256; d++) { w[d] = 1.0; for (int i = 0; i < d; i++) w[d] *= 0.985; tot += w[d]; }
        uint32_t used = 256;
        for (int d = 0; d < 256; d++) { const uint32_t x = (uint32_t)((TILE - 256) * w[d] / tot); len[d] += x; used += x; }
        len[0] += TILE - used;
        char what[128]; snprintf(what, sizeof what, "skewed: 256 runs, %u .. %u pairs (text-like digit weights)", len[255], len[0]);
        if (pattern(what, len)) return 1;
    }
    {   // strongly skewed: half of the tile in 8 runs, the rest in 248
        std::vector<uint32_t> len(256);
        uint32_t used = 0;
        for (int d = 0; d < 256; d++) { len[d] = d < 8 ? 256 : 8; used += len[d]; }
        len[8] += TILE - used;
        if (pattern("strongly skewed: 8 runs of 256 pairs + 248 runs of 8", len)) return 1;
    }
    return 0;
}
