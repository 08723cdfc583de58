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

// Pass 0 of the suffix sort reads its (key, value) pairs straight from the text: slot j holds suffix i = n-1-j, key = the
// first 7 bytes of the suffix, big-endian in bits 63..8, zero padded past the end (bwt_fwd.hip, round 0); bits 7..0 = T[i-1].  Three aligned dword
// loads per slot (neighbouring lanes share them), a funnel shift and a byte swap; nothing beyond the dword that holds
// T[n-1] is touched.
struct TextSrc {
    const uint32_t *tb;      // T rounded down to a dword boundary
    uint32_t off;            // T - tb (0..3)
    uint32_t n;
    // group sort (several blocks in one text, jpk_fwd_bwt_group_device): the block of every position and where each block ends -- a
    // suffix stops at the end of ITS block, and the key's low byte carries the block number (the last, most significant digit)
    const uint8_t *blk;      // null: one block
    const uint32_t *bend;
};
__device__ __forceinline__ uint64_t text_key7(const TextSrc &t, uint32_t i)
{
    const uint32_t a = i + t.off, wi = a >> 2, sh = (a & 3u) * 8u;
    const uint32_t lastw = (t.n - 1u + t.off) >> 2;
    const uint32_t w0 = t.tb[wi], w1 = t.tb[wi + 1 < lastw ? wi + 1 : lastw], w2 = t.tb[wi + 2 < lastw ? wi + 2 : lastw];
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    uint64_t v = sh ? (lo >> sh) | ((uint64_t)w2 << (64u - sh)) : lo;            // bytes i .. i+7, little endian
    if (t.blk) {                                                                    // (uniform branch)
        const uint32_t b = t.blk[i];
        const uint32_t left = t.bend[b] - i;                                        // bytes left in the suffix's own block, >= 1
        v &= (left < 8u) ? (1ull << (8u * left)) - 1ull : ~0ull;
        return (__builtin_bswap64(v) & ~0xFFull) | b;                               // low byte: the block number, the sort's last digit
    }
    const uint32_t left = t.n - i;                                                  // >= 1
    v &= (left < 8u) ? (1ull << (8u * left)) - 1ull : ~0ull;
    // The low byte of the key is never a sort digit (7 passes, bits 8..63): it carries T[i-1] (0 for suffix 0), the BWT byte of
    // the suffix, so that no kernel has to gather it from the text once the suffix's SA position is known.
    uint32_t prev = (uint32_t)reinterpret_cast<const uint8_t *>(t.tb)[a - (i ? 1u : 0u)];        // (no branch around the load)
    prev = i ? prev : 0u;
    return (__builtin_bswap64(v) & ~0xFFull) | prev;
}

// The sixteen pairs of a thread, all loads issued back to back: no branch around a load (slots past the end re-read the last
// pair and are zeroed afterwards).  With the loads inside `valid ? .. : 0` the compiler put an s_waitcnt vmcnt(0) behind every
// one of them -- a full memory latency per 64 elements instead of per tile.
template <bool TEXT>
__device__ __forceinline__ void rs_load_tile(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, const TextSrc &txt, size_t n, size_t base,
                                             uint64_t (&key)[RS_ITEMS], uint32_t (&val)[RS_ITEMS])
{
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        const size_t i = base + (size_t)it * 64, ic = i < n ? i : n - 1;
        if (TEXT) {
            val[it] = (uint32_t)(n - 1 - ic);
            key[it] = text_key7(txt, val[it]);
        } else {
            key[it] = kin[ic];
            val[it] = vin[ic];
        }
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++)
        if (base + (size_t)it * 64 >= n) { key[it] = 0; val[it] = 0; }
}

template <bool TEXT>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint64_t *__restrict__ keys, TextSrc txt, size_t n, int shift,
                                                       uint32_t *__restrict__ tilehist, uint32_t ntiles)
{
    __shared__ uint32_t h[RS_WAVES][256];
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * RS_TILE + (size_t)w * (64 * RS_ITEMS) + l;
    // all sixteen loads of a thread are issued before the first one is used (no branch around a load: slots past the end re-read
    // the last element and are masked afterwards) -- with a load inside `if (i < n)` the compiler waits for every load in turn
    uint32_t dig[RS_ITEMS];
    const uint64_t lt = lanemask_lt();
    // counting by wave match instead of LDS atomics: the lanes of a wave that hold the same digit are found with eight ballots
    // and ONE of them adds their number to the wave's counter -- plain LDS read-modify-write, one lane per address.  Text digits
    // are skewed (a tenth of the lanes of a wave hit the same bin): the atomic form spent 92 % of its LDS cycles in same-address
    // conflicts and cost about as much as the whole scatter pass.
    if (!TEXT && (shift & 7) == 0 && (size_t)(blockIdx.x + 1) * RS_TILE <= n) {
        // a whole tile of keys and a digit that is a byte of the key (every pass of the suffix sort): the digit is loaded as that
        // byte from a wave-uniform base with the item offset in the instruction -- no clamp, no 64-bit shift, no validity masks
        const uint8_t *bp = reinterpret_cast<const uint8_t *>(keys + (size_t)blockIdx.x * RS_TILE + (size_t)w * (64 * RS_ITEMS)) + (shift >> 3);
        const uint32_t lo = (uint32_t)l * 8u;
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) dig[it] = bp[lo + (uint32_t)it * 512u];
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) {
            const uint64_t m = match_any8(dig[it], true);
            if ((m & lt) == 0ull) h[w][dig[it]] += (uint32_t)__popcll(m);
        }
    } else {
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) {
            const size_t i = base + (size_t)it * 64, ic = i < n ? i : n - 1;
            if (TEXT) {                                    // digit of byte (56 - shift) / 8 of the suffix: one text byte
                const uint32_t i0 = (uint32_t)(n - 1 - ic), pos = i0 + (uint32_t)((56 - shift) >> 3);
                dig[it] = reinterpret_cast<const uint8_t *>(txt.tb)[(pos < txt.n ? pos : txt.n - 1u) + txt.off];
                const uint32_t lim = txt.blk ? txt.bend[txt.blk[i0]] : txt.n;          // the suffix ends with its block
                if (pos >= lim) dig[it] = 0u;
            } else dig[it] = (uint32_t)(keys[ic] >> shift) & 255u;
        }
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) {
            const bool valid = base + (size_t)it * 64 < n;
            const uint64_t m = match_any8(dig[it], valid);
            if (valid && (m & lt) == 0ull) h[w][dig[it]] += (uint32_t)__popcll(m);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += RS_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) s += h[k][d];
        tilehist[(size_t)d * ntiles + blockIdx.x] = s;
    }
}

template <bool TEXT>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, TextSrc txt,
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
    rs_load_tile<TEXT>(kin, vin, txt, n, base, key, val);
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        size_t i = base + (size_t)it * 64;
        bool valid = i < n;
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

// The same pass with the tile's pairs re-ordered in LDS before they leave: a thread's sixteen (digit, rank) results place its
// pairs at their position in the TILE-sorted order, and the stores then walk that order -- consecutive lanes write consecutive
// addresses inside one digit's run (whole 64-byte lines for every run of >= 8 pairs) instead of the 2-3 pairs per digit that one
// load instruction's 64 lanes happen to share.  One 32 KB staging buffer is used twice (keys, then values).
// FULL: the tile has all RS_TILE pairs (every tile but the last): no bounds logic at all, and the loads go through a wave-uniform
// base pointer with 32-bit lane offsets (the general form computes a 64-bit address per load).
template <bool TEXT, bool FULL>
__device__ __forceinline__ void rs_scatter_staged_tile(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, const TextSrc &txt,
                                                       uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                       const uint32_t *__restrict__ tileoff, uint32_t ntiles, uint32_t tile, uint32_t (*cnt)[256],
                                                       uint32_t *gbase, uint64_t *stage, uint32_t *sm)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l = threadIdx.x & 63;
    const size_t tbase = (size_t)tile * RS_TILE;
    const size_t base = tbase + (size_t)w * (64 * RS_ITEMS) + l;
    const uint32_t tcount = FULL ? (uint32_t)RS_TILE : (uint32_t)(n - tbase);
    uint64_t key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rnk[RS_ITEMS];
    const uint64_t lt = lanemask_lt();
    if (FULL && !TEXT) {
        const uint64_t *kw = kin + tbase + (size_t)w * (64 * RS_ITEMS);      // wave-uniform
        const uint32_t *vw = vin + tbase + (size_t)w * (64 * RS_ITEMS);
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) { key[it] = kw[(uint32_t)(it * 64 + l)]; val[it] = vw[(uint32_t)(it * 64 + l)]; }
    } else rs_load_tile<TEXT>(kin, vin, txt, n, base, key, val);
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        const bool valid = FULL || base + (size_t)it * 64 < n;
        const uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
        const uint64_t m = match_any8(d, valid);
        const uint32_t below = (uint32_t)__popcll(m & lt);
        const uint32_t c = valid ? cnt[w][d] : 0;
        rnk[it] = c + below;
        if (valid && below == 0) cnt[w][d] = c + (uint32_t)__popcll(m);
    }
    __syncthreads();
    {   // thread d: exclusive scan across waves of digit d, then across digits; cnt[k][d] becomes the position, inside the
        // tile-sorted order, of wave k's first pair with digit d
        const int d = threadIdx.x;
        uint32_t c4[RS_WAVES], s = 0;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { c4[k] = s; s += cnt[k][d]; }
        const uint32_t inc = block_incl_scan<OpSum>(s, sm, nullptr);
        const uint32_t ts = inc - s;
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) cnt[k][d] = ts + c4[k];
        gbase[d] = tileoff[(size_t)d * ntiles + tile] - ts;
    }
    __syncthreads();
    uint32_t pos[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        const uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
        pos[it] = cnt[w][d] + rnk[it];
        if (FULL || base + (size_t)it * 64 < n) stage[pos[it]] = key[it];
    }
    __syncthreads();
    uint32_t dstv[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        const uint32_t p = (uint32_t)it * RS_THREADS + threadIdx.x;
        dstv[it] = 0xFFFFFFFFu;
        if (FULL || p < tcount) {
            const uint64_t k = stage[p];
            const uint32_t d = (uint32_t)(k >> shift) & 255u;
            dstv[it] = gbase[d] + p;
            kout[dstv[it]] = k;
        }
    }
    __syncthreads();
    uint32_t *stage32 = reinterpret_cast<uint32_t *>(stage);
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++)
        if (FULL || base + (size_t)it * 64 < n) stage32[pos[it]] = val[it];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        const uint32_t p = (uint32_t)it * RS_THREADS + threadIdx.x;
        if (FULL || p < tcount) vout[dstv[it]] = stage32[p];
    }
}

// two launches per pass: the full tiles (no bounds logic) and, if n is not a multiple of the tile, the last tile alone
template <bool TEXT, bool FULL>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter_staged(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, TextSrc txt,
                                                                 uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                                 const uint32_t *__restrict__ tileoff, uint32_t ntiles, uint32_t tile0)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t gbase[256];          // global offset of the tile's run of digit d, minus the run's start inside the tile
    __shared__ uint64_t stage[RS_TILE];
    __shared__ uint32_t sm[RS_THREADS / 64 + 1];
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    rs_scatter_staged_tile<TEXT, FULL>(kin, vin, txt, kout, vout, n, shift, tileoff, ntiles, tile0 + blockIdx.x, cnt, gbase, stage, sm);
}

template <bool TEXT>
void launch_rs_scatter_staged(jpk_ctx *ctx, const uint64_t *kin, const uint32_t *vin, const TextSrc &txt, uint64_t *kout, uint32_t *vout, size_t n, int shift,
                              const uint32_t *tileoff, uint32_t ntiles)
{
    const uint32_t nfull = (uint32_t)(n / RS_TILE);
    if (nfull) JPK_LAUNCH(ctx, PROF_RS_SCATTER, (size_t)nfull * RS_TILE, (k_rs_scatter_staged<TEXT, true>), dim3(nfull), dim3(RS_THREADS), kin, vin, txt, kout, vout, n, shift,
                          tileoff, ntiles, 0u);
    if (nfull < ntiles) JPK_LAUNCH(ctx, PROF_RS_SCATTER, n - (size_t)nfull * RS_TILE, (k_rs_scatter_staged<TEXT, false>), dim3(1), dim3(RS_THREADS), kin, vin, txt, kout, vout, n,
                                   shift, tileoff, ntiles, nfull);
}

// JPK_RS_STAGED=0 keeps the direct scatter (register -> global) for comparison
bool rs_staged()
{
    static const bool on = [] { const char *e = getenv("JPK_RS_STAGED"); return e ? atoi(e) != 0 : true; }();
    return on;
}

}  // namespace

size_t jpk_radix_scratch_words(size_t n)
{
    size_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    size_t table = 256 * ntiles;
    return table + jpk_scan_scratch_words(table) + 64;
}

// the sorted pairs end up in (*keys_out, *vals_out): the caller's buffers after an even number of passes, the alt buffers after an
// odd number -- no copy back
int jpk_radix_sort_pairs_u64_nocopy(jpk_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt, size_t n,
                                    const int *shifts, int nshifts, uint32_t *scratch, uint64_t **keys_out, uint32_t **vals_out)
{
    *keys_out = keys;
    *vals_out = vals;
    if (n == 0 || nshifts == 0) return JPK_OK;
    const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t table = (size_t)256 * ntiles;
    uint32_t *hist = scratch;
    uint32_t *scan_scratch = scratch + table;
    uint64_t *ki = keys, *ko = keys_alt;
    uint32_t *vi = vals, *vo = vals_alt;
    for (int p = 0; p < nshifts; p++) {
        const TextSrc none = {nullptr, 0u, 0u, nullptr, nullptr};
        JPK_LAUNCH(ctx, PROF_RS_HIST, n, (k_rs_hist<false>), dim3(ntiles), dim3(RS_THREADS), ki, none, n, shifts[p], hist, ntiles);
        JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
        if (rs_staged()) launch_rs_scatter_staged<false>(ctx, ki, vi, none, ko, vo, n, shifts[p], hist, ntiles);
        else JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, (k_rs_scatter<false>), dim3(ntiles), dim3(RS_THREADS), ki, vi, none, ko, vo, n, shifts[p], hist, ntiles);
        uint64_t *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
    JPK_HIP(hipGetLastError());
    *keys_out = ki;
    *vals_out = vi;
    return JPK_OK;
}

int jpk_radix_sort_pairs_u64(jpk_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt, size_t n,
                             const int *shifts, int nshifts, uint32_t *scratch)
{
    uint64_t *ki;
    uint32_t *vi;
    JPK_TRY(jpk_radix_sort_pairs_u64_nocopy(ctx, keys, vals, keys_alt, vals_alt, n, shifts, nshifts, scratch, &ki, &vi));
    if (ki != keys) {
        JPK_HIP(hipMemcpyAsync(keys, ki, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        JPK_HIP(hipMemcpyAsync(vals, vi, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return JPK_OK;
}

// Round 0 of the suffix sort (bwt_fwd.hip): all n suffixes of T by their first 7 bytes, 7 LSD passes; the first pass builds the
// keys from the text on the fly, so no key array is written or read for it.  Result: (keysB, valsB) -- 7 passes, the first one
// lands in B.
int jpk_radix_sort_suffix_keys7(jpk_ctx *ctx, const uint8_t *T, uint32_t n32, uint64_t *keysA, uint32_t *valsA, uint64_t *keysB, uint32_t *valsB,
                                uint32_t *scratch, uint64_t **keys_out, uint32_t **vals_out, const uint8_t *blk, const uint32_t *bend)
{
    const size_t n = n32;
    *keys_out = keysB;
    *vals_out = valsB;
    if (n == 0) return JPK_OK;
    const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t table = (size_t)256 * ntiles;
    uint32_t *hist = scratch;
    uint32_t *scan_scratch = scratch + table;
    TextSrc txt;
    txt.off = (uint32_t)((uintptr_t)T & 3u);
    txt.tb = reinterpret_cast<const uint32_t *>(T - txt.off);
    txt.n = n32;
    txt.blk = blk;
    txt.bend = bend;
    const TextSrc none = {nullptr, 0u, 0u, nullptr, nullptr};
    uint64_t *ki = keysB, *ko = keysA;        // after pass 0 the pairs are in B
    uint32_t *vi = valsB, *vo = valsA;
    const int npass = blk ? 8 : 7;                // group sort: one more pass, on the block number in the key's low byte
    for (int p = 0; p < npass; p++) {
        const int shift = p < 7 ? 8 * (p + 1) : 0;
        if (p == 0) {
            JPK_LAUNCH(ctx, PROF_RS_HIST, n, (k_rs_hist<true>), dim3(ntiles), dim3(RS_THREADS), (const uint64_t *)nullptr, txt, n, shift, hist, ntiles);
            JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
            if (rs_staged()) launch_rs_scatter_staged<true>(ctx, nullptr, nullptr, txt, keysB, valsB, n, shift, hist, ntiles);
            else JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, (k_rs_scatter<true>), dim3(ntiles), dim3(RS_THREADS), (const uint64_t *)nullptr, (const uint32_t *)nullptr, txt,
                       keysB, valsB, n, shift, hist, ntiles);
            continue;
        }
        JPK_LAUNCH(ctx, PROF_RS_HIST, n, (k_rs_hist<false>), dim3(ntiles), dim3(RS_THREADS), ki, none, n, shift, hist, ntiles);
        JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
        if (rs_staged()) launch_rs_scatter_staged<false>(ctx, ki, vi, none, ko, vo, n, shift, hist, ntiles);
        else JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, (k_rs_scatter<false>), dim3(ntiles), dim3(RS_THREADS), ki, vi, none, ko, vo, n, shift, hist, ntiles);
        uint64_t *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
    JPK_HIP(hipGetLastError());
    *keys_out = ki;
    *vals_out = vi;
    return JPK_OK;
}
