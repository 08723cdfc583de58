// radix.hip -- stable LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits.
//
// Per pass (reduce-then-scan):
//   k_rs_hist     tile histograms from an LDS-staged 256-bin table      reads 8 B/elem
//   exclusive sum over the digit-major [256][ntiles] table (scan.hip)
//   k_rs_scatter  stable in-tile ranks by wave match-any + per-wave LDS counters, scatter
//                                                                        reads 12 B, writes 12 B / elem
// HBM-bound integer work: no MFMA.  Tiles are 4096 elements (256 threads x 16, wave-striped so that
// every load instruction of a wave is one contiguous 512-B / 256-B segment).
#include "common.hpp"
#include "prims.hpp"

using namespace jpk;

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint64_t *__restrict__ keys, size_t n, int shift,
                                                       uint32_t *__restrict__ tilehist, uint32_t ntiles)
{
    __shared__ uint32_t h[RS_WAVES][256];
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * RS_TILE + (size_t)w * (64 * RS_ITEMS) + l;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        size_t i = base + (size_t)it * 64;
        if (i < n) atomicAdd(&h[w][(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += RS_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) s += h[k][d];
        tilehist[(size_t)d * ntiles + blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                          uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                          const uint32_t *__restrict__ tileoff, uint32_t ntiles)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t gbase[256];
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    for (int d = threadIdx.x; d < 256; d += RS_THREADS) gbase[d] = tileoff[(size_t)d * ntiles + blockIdx.x];
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * RS_TILE + (size_t)w * (64 * RS_ITEMS) + l;
    uint64_t key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rnk[RS_ITEMS];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        size_t i = base + (size_t)it * 64;
        bool valid = i < n;
        key[it] = valid ? kin[i] : 0;
        val[it] = valid ? vin[i] : 0;
        uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
        uint64_t m = match_any8(d, valid);
        uint32_t below = (uint32_t)__popcll(m & lt);
        uint32_t c = valid ? cnt[w][d] : 0;
        rnk[it] = c + below;
        // every lane of the match group has read cnt before its leader bumps it: same wave, program order
        if (valid && below == 0) cnt[w][d] = c + (uint32_t)__popcll(m);
    }
    __syncthreads();
    // exclusive scan across waves per digit
    for (int d = threadIdx.x; d < 256; d += RS_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { uint32_t t = cnt[k][d]; cnt[k][d] = s; s += t; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        size_t i = base + (size_t)it * 64;
        if (i < n) {
            uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
            size_t dst = (size_t)gbase[d] + cnt[w][d] + rnk[it];
            kout[dst] = key[it];
            vout[dst] = val[it];
        }
    }
}

}  // namespace

size_t jpk_radix_scratch_words(size_t n)
{
    size_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    size_t table = 256 * ntiles;
    return table + jpk_scan_scratch_words(table) + 64;
}

int jpk_radix_sort_pairs_u64(jpk_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt, size_t n,
                             const int *shifts, int nshifts, uint32_t *scratch)
{
    if (n == 0 || nshifts == 0) return JPK_OK;
    const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t table = (size_t)256 * ntiles;
    uint32_t *hist = scratch;
    uint32_t *scan_scratch = scratch + table;
    uint64_t *ki = keys, *ko = keys_alt;
    uint32_t *vi = vals, *vo = vals_alt;
    for (int p = 0; p < nshifts; p++) {
        JPK_LAUNCH(ctx, PROF_RS_HIST, n, k_rs_hist, dim3(ntiles), dim3(RS_THREADS), ki, n, shifts[p], hist, ntiles);
        JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
        JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, k_rs_scatter, dim3(ntiles), dim3(RS_THREADS), ki, vi, ko, vo, n, shifts[p], hist, ntiles);
        uint64_t *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
    JPK_HIP(hipGetLastError());
    if (ki != keys) {
        JPK_HIP(hipMemcpyAsync(keys, ki, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        JPK_HIP(hipMemcpyAsync(vals, vi, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return JPK_OK;
}
