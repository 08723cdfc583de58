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
static_assert(RS_TILE == 4096, "bwt_fwd.hip k_pack_keys writes the first pass's tile histogram for tiles of 4096 slots");

// Pass 0 of the suffix sort (SLOTS): the values are implicit -- slot j holds suffix n-1-j, its key comes from the packed key array
// (bwt_fwd.hip k_pack_keys) -- so no value array is read.
// The sixteen pairs of a thread, all loads issued back to back: no branch around a load (slots past the end re-read the last
// pair and are zeroed afterwards).  With the loads inside `valid ? .. : 0` the compiler put an s_waitcnt vmcnt(0) behind every
// one of them -- a full memory latency per 64 elements instead of per tile.
template <bool SLOTS>
__device__ __forceinline__ void rs_load_tile(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, size_t n, size_t base,
                                             uint64_t (&key)[RS_ITEMS], uint32_t (&val)[RS_ITEMS], int tag_shift = 26)
{
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++) {
        const size_t i = base + (size_t)it * 64, ic = i < n ? i : n - 1;
        key[it] = kin[ic];
        val[it] = SLOTS ? (uint32_t)(n - 1 - ic) : vin[ic];
    }
    if (SLOTS && vin) {                                 // (uniform) slot tags: see jpk_radix_sort_slot_keys
        const uint8_t *tag = reinterpret_cast<const uint8_t *>(vin);
        uint32_t tg[RS_ITEMS];
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) { const size_t i = base + (size_t)it * 64; tg[it] = tag[i < n ? i : n - 1]; }
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) val[it] |= tg[it] << tag_shift;
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; it++)
        if (base + (size_t)it * 64 >= n) { key[it] = 0; val[it] = 0; }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint64_t *__restrict__ keys, size_t n, int shift,
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
    // conflicts and cost about as much as the whole scatter pass.  (Round 4 tried both per row -- atomics where the row's digits are
    // spread out, as the packed keys' are, the match where they are not: 12 % fewer vector instructions per block and the same
    // bench line, profiles/r04_hist_rows.txt -- the kernel waits for its loads either way.)
    if ((shift & 7) == 0 && (size_t)(blockIdx.x + 1) * RS_TILE <= n) {
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
            dig[it] = (uint32_t)(keys[ic] >> shift) & 255u;
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

template <bool SLOTS>
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
    rs_load_tile<SLOTS>(kin, vin, n, base, key, val);
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
// One-pass variant (round 4, "Onesweep"): no histogram pass and no scan in front of the scatter.  The GLOBAL digit offsets of every
// pass of the suffix sort are known before the first one (the digits are text bytes: one byte histogram of the text, k_os_*), and a
// tile learns how many pairs with its digit the tiles before it hold by decoupled look-back: every tile publishes, per digit, first
// its own count (flag 1) and then the inclusive prefix (flag 2) in one 32-bit word (flag << 30 | count, counts < 2^30), thread d
// walks back from tile - 1 adding counts until it meets a prefix.  Tiles take their numbers from a ticket counter, so a tile only
// ever waits for tiles that started before it.  The words are read and written with agent-scope atomics (the XCDs' L2s are not
// coherent with each other).  Two status arrays alternate between the passes; a tile zeroes its row of the other one.
struct OsArgs {
    uint32_t *status;          // [ntiles][256] of this pass
    uint32_t *status_next;     // ... of the next pass: zeroed row by row
    const uint32_t *gdig;      // [256] exclusive global offsets of this pass's digit
    uint32_t *ticket;          // tile numbers of this pass
    int tag_shift;             // pass 0 with slot tags: the tag goes into the value's bits from here up
    uint32_t slot_n;           // pass 0 (SLOTS): slot j holds suffix slot_n - 1 - j (the sort itself may cover pad slots behind slot_n - 1, whose
                               // keys are the largest: the stable sort leaves them at the end)
};
constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_PFX = 2u << 30, OS_MASK = (1u << 30) - 1u;

template <bool SLOTS, bool FULL, bool LOOKBACK = false>
__device__ __forceinline__ void rs_scatter_staged_tile(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                       uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                       const uint32_t *__restrict__ tileoff, uint32_t ntiles, uint32_t tile, uint32_t (*cnt)[256],
                                                       uint32_t *gbase, uint64_t *stage, uint32_t *sm, const OsArgs *os = nullptr)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), l = threadIdx.x & 63;
    const size_t tbase = (size_t)tile * RS_TILE;
    const size_t base = tbase + (size_t)w * (64 * RS_ITEMS) + l;
    const uint32_t tcount = FULL ? (uint32_t)RS_TILE : (uint32_t)(n - tbase);
    uint64_t key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rnk[RS_ITEMS];
    const uint64_t lt = lanemask_lt();
    // (one-pass form) the digit's global offset is requested with the tile's pairs, in front of everything: loaded behind the look-back
    // it made the wave wait for vmcnt(0) there -- i.e. for the acknowledgement of the two status stores in front of it, a memory round
    // trip on the tile's critical path (round 6)
    uint32_t gdig_d = 0;
    if (LOOKBACK) gdig_d = os->gdig[threadIdx.x];
    if (FULL) {
        const uint64_t *kw = kin + tbase + (size_t)w * (64 * RS_ITEMS);      // wave-uniform
        const uint32_t *vw = vin + tbase + (size_t)w * (64 * RS_ITEMS);
        const uint32_t v0 = (uint32_t)((os ? (size_t)os->slot_n : n) - 1 - (tbase + (size_t)w * (64 * RS_ITEMS)));   // SLOTS: slot j holds suffix n-1-j (pad slots wrap: never read)
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++) {
            key[it] = kw[(uint32_t)(it * 64 + l)];
            val[it] = SLOTS ? v0 - (uint32_t)(it * 64 + l) : vw[(uint32_t)(it * 64 + l)];
        }
        if (SLOTS && vin) {                             // (uniform) slot tags: see jpk_radix_sort_slot_keys
            const uint8_t *tw = reinterpret_cast<const uint8_t *>(vin) + tbase + (size_t)w * (64 * RS_ITEMS);
            uint32_t tg[RS_ITEMS];
#pragma unroll
            for (int it = 0; it < RS_ITEMS; it++) tg[it] = tw[(uint32_t)(it * 64 + l)];
#pragma unroll
            for (int it = 0; it < RS_ITEMS; it++) val[it] |= tg[it] << (os ? os->tag_shift : 26);
        }
    } else rs_load_tile<SLOTS>(kin, vin, n, base, key, val, os ? os->tag_shift : 26);
    // (one-pass form, round 6) THE AGGREGATE LEAVES EARLY: the tile's digit counts by LDS atomics as soon as its keys are there, published
    // as the tile's aggregate BEFORE the match phase (rounds 4-5: after it and the scan), so that the tiles behind this one find a published
    // word a match phase earlier instead of polling an empty one -- 483 -> 446 us per pass, forward BWT 7.86 -> 7.64 ms (wide text 7.50 ->
    // 7.17), profiles/r06_scatter_ab.txt.  gbase holds the counts until the scan below; the packed keys' digits are spread over all 256
    // bins: few same-address conflicts.
    if (LOOKBACK) {
#pragma unroll
        for (int it = 0; it < RS_ITEMS; it++)
            if (FULL || base + (size_t)it * 64 < n) atomicAdd(&gbase[(uint32_t)(key[it] >> shift) & 255u], 1u);
        __syncthreads();
        const uint32_t s0 = gbase[threadIdx.x];
        __hip_atomic_store(os->status + (size_t)tile * 256u + threadIdx.x, (tile == 0 ? OS_FLAG_PFX : OS_FLAG_AGG) | s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
        uint32_t goff;
        if (LOOKBACK) {
            uint32_t *row = os->status + (size_t)tile * 256u;
            uint32_t excl = 0;
            if (tile != 0) {                                                // (tile 0's word is a prefix already; every aggregate left before the match phase)
                // The walk is latency: every word comes from the memory side (~1.5 us), and with ~1000 tiles resident a tile finds
                // aggregates, not prefixes, for a long way back.  Four rows are requested at once and consumed in order; the walk
                // stops at the first prefix, an unpublished word is polled alone.  (Forward BWT of a 64 MiB block with 1 / 2 / 3 / 4 / 8 / 16 /
                // 32 rows at once: 10.05 / 9.88 / 10.00 / 9.86 / 10.03 / 10.35 / 10.49 ms.)
                constexpr int LB = 4;
                uint32_t back = 1;                                           // rows behind `tile` of the next word to consume
                bool done = false;
                while (!done) {
                    uint32_t v[LB];
#pragma unroll
                    for (int k = 0; k < LB; k++) {
                        const uint32_t b = back + (uint32_t)k;
                        v[k] = b <= tile ? __hip_atomic_load(row + d - (size_t)b * 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < LB; k++) {
                        if (done) break;
                        uint32_t w = v[k];
                        if ((w >> 30) == 0u) {                                // not published yet (rows past tile 0 are never reached: tile 0 is a prefix)
                            const uint32_t *q = row + d - (size_t)(back + (uint32_t)k) * 256u;
                            while (((w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u) __builtin_amdgcn_s_sleep(1);
                        }
                        excl += w & OS_MASK;
                        if ((w >> 30) == 2u) done = true;
                    }
                    back += LB;
                }
                __hip_atomic_store(row + d, OS_FLAG_PFX | (excl + s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            os->status_next[(size_t)tile * 256u + d] = 0u;
            goff = gdig_d + excl;
        } else goff = tileoff[(size_t)d * ntiles + tile];
        gbase[d] = goff - ts;
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
template <bool SLOTS, bool FULL>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter_staged(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                                 uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift,
                                                                 const uint32_t *__restrict__ tileoff, uint32_t ntiles, uint32_t tile0)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t gbase[256];          // global offset of the tile's run of digit d, minus the run's start inside the tile
    __shared__ uint64_t stage[RS_TILE];
    __shared__ uint32_t sm[RS_THREADS / 64 + 1];
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    rs_scatter_staged_tile<SLOTS, FULL>(kin, vin, kout, vout, n, shift, tileoff, ntiles, tile0 + blockIdx.x, cnt, gbase, stage, sm);
}

// (two launches per pass, like the two-pass form: the full tiles -- no bounds logic, a third of the registers -- and then the last,
// partial tile alone; it takes the next ticket and finds every prefix published.  The suffix sort pads its slots to whole tiles since
// round 6: only the first launch exists there.)
// Round 6, what else was built around the look-back and NOT kept (profiles/r06_scatter_ab.txt):
//   * twice the occupancy -- the same tile on 512 threads x 8 pairs, 64 registers, eight waves per SIMD, the per-wave counters inside the
//     staging buffer (34 KB of LDS): 496 us per pass against 483, bench line 5 767-5 891 against 5 831-6 040 MB/s.  The waves do not wait
//     for each other's latencies;
//   * a two-level look-back -- groups of sixteen tiles, tile rows that only ever hold the tile's own counts, the group's last tile
//     publishing the group's aggregate and then its prefix, every tile reading <= 15 rows of its own group and the group rows in front of it
//     (one or two round trips, 14 KB of status words per tile instead of a walk of four rows per round trip): 474 us per pass with the
//     early aggregate against 446 without the second level, forward BWT 7.77 against 7.64 ms.  The walk is short once the aggregates are
//     out early; what a tile waits for is that its predecessors HAVE counted their keys.
// (With the early aggregate the kernel takes 130 registers: three waves per SIMD, three tiles per CU.  Forced back to 128 and four --
// amdgpu_waves_per_eu(4, 4) -- it measured 7.65 against 7.55 ms per forward BWT and 6 223 against 6 349 MB/s on the bench line; with
// two tiles per CU (JPK_OS_LDS=16384) 7.72 ms: profiles/r06_scatter_ab.txt.  Three it is.)
template <bool SLOTS, bool FULL>
__global__ __launch_bounds__(RS_THREADS) void k_os_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                          uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, size_t n, int shift, uint32_t ntiles, OsArgs os)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t gbase[256];
    __shared__ uint64_t stage[RS_TILE];
    __shared__ uint32_t sm[RS_THREADS / 64 + 1];
    __shared__ uint32_t s_tile;
    // (The ticket is one device-scope atomic on ONE address per tile: alone it costs 11.6 ns -- 16384 workgroups that do nothing else take 190 us,
    // 33 us with eight counters, tools/tickettest.hip -- but the pass does not wait for it: with the tile taken from blockIdx.x, as a timing
    // experiment on an otherwise idle GPU, a pass took 458 us against 441 with tickets, profiles/r06_scatter_ab.txt.  Tickets are what makes the
    // look-back safe -- a tile only ever waits for tiles that have STARTED -- and they stay.)
    if (threadIdx.x == 0) s_tile = atomicAdd(os.ticket, 1u);
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    gbase[threadIdx.x] = 0;                       // (the early aggregate's counters)
    __syncthreads();
    const uint32_t tile = s_tile;
    rs_scatter_staged_tile<SLOTS, FULL, true>(kin, vin, kout, vout, n, shift, nullptr, ntiles, tile, cnt, gbase, stage, sm, &os);
}
template <bool SLOTS>
void launch_os_scatter(jpk_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, size_t n, int shift, uint32_t ntiles,
                       const OsArgs &os)
{
    const uint32_t nfull = (uint32_t)(n / RS_TILE);
    // JPK_OS_LDS: extra dynamic LDS per workgroup (bytes), i.e. fewer tiles resident per CU (experiments: 16384 = two per CU)
    static const size_t extra_lds = [] { const char *e = getenv("JPK_OS_LDS"); const long v = e ? atol(e) : 0; return (size_t)(v < 0 ? 0 : (v > 120000 ? 120000 : v)); }();
    if (nfull) JPK_LAUNCH_LDS(ctx, PROF_RS_SCATTER, (size_t)nfull * RS_TILE, extra_lds, (k_os_scatter<SLOTS, true>), dim3(nfull), dim3(RS_THREADS), kin, vin, kout, vout, n, shift, ntiles, os);
    if (nfull < ntiles) JPK_LAUNCH(ctx, PROF_RS_SCATTER, n - (size_t)nfull * RS_TILE, (k_os_scatter<SLOTS, false>), dim3(1), dim3(RS_THREADS), kin, vin, kout, vout, n, shift,
                                   ntiles, os);
}

// digit histograms of all passes (seven; eight in a group sort: the last one on the key's low byte) from one read of the packed keys,
// then their exclusive offsets (one-pass form only).  (Counting them inside k_pack_keys, where the keys sit in LDS anyway, was slower:
// 0.51 ms for the fused kernel against 0.28 + 0.20.)
constexpr int OS_PASSES = 8;
__global__ __launch_bounds__(RS_THREADS) void k_os_digits(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ H, int npass)
{
    __shared__ uint32_t h[OS_PASSES][256];
    for (int i = threadIdx.x; i < OS_PASSES * 256; i += RS_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * RS_THREADS) {
        const uint64_t k = keys[i];
#pragma unroll
        for (int p = 0; p < 7; p++) atomicAdd(&h[p][(uint32_t)(k >> (8 * (p + 1))) & 255u], 1u);
        if (npass > 7) atomicAdd(&h[7][(uint32_t)k & 255u], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npass * 256; i += RS_THREADS)
        if ((&h[0][0])[i]) atomicAdd(&H[i], (&h[0][0])[i]);
}
__global__ __launch_bounds__(256) void k_os_prefix(const uint32_t *__restrict__ H, uint32_t *__restrict__ gdig, int npass)
{
    __shared__ uint32_t sm[256 / 64 + 1];
    for (int p = 0; p < npass; p++) {
        const uint32_t c = H[p * 256 + threadIdx.x];
        const uint32_t inc = block_incl_scan<OpSum>(c, sm, nullptr);
        gdig[p * 256 + threadIdx.x] = inc - c;
        __syncthreads();
    }
}

template <bool SLOTS>
void launch_rs_scatter_staged(jpk_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, size_t n, int shift,
                              const uint32_t *tileoff, uint32_t ntiles)
{
    const uint32_t nfull = (uint32_t)(n / RS_TILE);
    if (nfull) JPK_LAUNCH(ctx, PROF_RS_SCATTER, (size_t)nfull * RS_TILE, (k_rs_scatter_staged<SLOTS, true>), dim3(nfull), dim3(RS_THREADS), kin, vin, kout, vout, n, shift,
                          tileoff, ntiles, 0u);
    if (nfull < ntiles) JPK_LAUNCH(ctx, PROF_RS_SCATTER, n - (size_t)nfull * RS_TILE, (k_rs_scatter_staged<SLOTS, false>), dim3(1), dim3(RS_THREADS), kin, vin, kout, vout, n,
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
    // the tile table (= status array A of the one-pass sort) + the scan's scratch, then status array B, the digit histograms and
    // offsets of seven passes and the tickets
    return table + jpk_scan_scratch_words(table) + 64 + table + OS_PASSES * 256 + OS_PASSES * 256 + 64;
}

// The one-pass (decoupled look-back) form of the suffix sort's round 0; JPK_ONESWEEP=0 selects the two-pass form (histogram + scan + scatter
// per pass).  With round 3's 7-byte text keys the two were the same (profiles/r04_onesweep_ab.txt: a one-pass pass took 455 us against 319 +
// 143 + scans -- the look-back words come from the memory side and a tile waits for them while it holds a quarter of a CU's LDS).  The
// packed keys (bwt_fwd.hip k_pack_keys) changed the balance: their digits are spread over all 256 bins, the two-pass scatter writes
// shorter runs (377 us) and the histogram costs what it cost: forward BWT of a 64 MiB block 10.3 -> 9.95 ms, bench line +3.5 % in five
// alternations out of five, groups of 1 MiB blocks +8 %, of 8 MiB blocks +2..10 % (profiles/r04_onesweep_packed.txt).  Default since then.
static bool rs_onesweep()
{
    static const bool on = [] { const char *e = getenv("JPK_ONESWEEP"); return e ? atoi(e) != 0 : true; }();
    return on;
}
bool jpk_radix_onesweep() { return rs_onesweep(); }

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
        JPK_LAUNCH(ctx, PROF_RS_HIST, n, k_rs_hist, dim3(ntiles), dim3(RS_THREADS), ki, n, shifts[p], hist, ntiles);
        JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
        if (rs_staged()) launch_rs_scatter_staged<false>(ctx, ki, vi, ko, vo, n, shifts[p], hist, ntiles);
        else JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, (k_rs_scatter<false>), dim3(ntiles), dim3(RS_THREADS), ki, vi, ko, vo, n, shifts[p], hist, ntiles);
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

// Round 0 of the suffix sort (bwt_fwd.hip): slot j of keysA holds the packed key of suffix n-1-j (k_pack_keys); 7 LSD passes over
// its 56 key bits -- 8 in a group sort, the last one on the block number in the low byte.  The first pass has no value array to read
// (slot j = suffix n-1-j) and no histogram to count (k_pack_keys left the digit-major tile table of key bits 15..8 at the start of
// `scratch`; its tiles are RS_TILE slots like ours) and lands in B; keysA is overwritten by the second.  Result: (*keys_out, *vals_out).
// `slot_tag` (one-pass form only; may be null): a byte per slot that rides in the bits of the slot's value from `tag_shift` up through the
// sort (the suffix sort's variable-length keys carry the number of symbols a key holds there); n <= 2^tag_shift, tag < 2^(32 - tag_shift).
// `slot_n` (one-pass form only; 0 = n32): the sort covers n32 >= slot_n slots, the last n32 - slot_n of them pads with the largest key (the
// caller rounds n32 up to whole tiles so that no pass needs a second launch for a partial tile); slot j < slot_n holds suffix slot_n - 1 - j.
int jpk_radix_sort_slot_keys(jpk_ctx *ctx, uint32_t n32, uint64_t *keysA, uint32_t *valsA, uint64_t *keysB, uint32_t *valsB,
                             uint32_t *scratch, uint64_t **keys_out, uint32_t **vals_out, bool group, const uint8_t *slot_tag, int tag_shift, uint32_t slot_n)
{
    if (slot_n == 0) slot_n = n32;
    if (slot_n > n32 || (slot_n != n32 && (!rs_onesweep() || n32 % RS_TILE))) return JPK_E_ARG;
    if (slot_tag && (!rs_onesweep() || tag_shift < 1 || tag_shift > 31 || (uint64_t)slot_n > (1ull << tag_shift))) return JPK_E_ARG;
    const size_t n = n32;
    *keys_out = keysB;
    *vals_out = valsB;
    if (n == 0) return JPK_OK;
    const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t table = (size_t)256 * ntiles;
    uint32_t *hist = scratch;
    uint32_t *scan_scratch = scratch + table;
    uint64_t *ki = keysB, *ko = keysA;        // after pass 0 the pairs are in B
    uint32_t *vi = valsB, *vo = valsA;
    const int npass = group ? 8 : 7;              // group sort: one more pass, on the block number in the key's low byte
    if (rs_onesweep()) {
        // one-pass sort: digit histograms of the keys -> digit offsets of all passes; then one scatter launch per pass, nothing else
        uint32_t *statusA = scratch, *statusB = scratch + table + jpk_scan_scratch_words(table) + 64;
        uint32_t *H = statusB + table, *gdig = H + OS_PASSES * 256, *tickets = gdig + OS_PASSES * 256;
        JPK_HIP(hipMemsetAsync(statusA, 0, table * 4, ctx->stream));
        JPK_HIP(hipMemsetAsync(statusB, 0, (table + OS_PASSES * 256 + OS_PASSES * 256 + 64) * 4, ctx->stream));
        {   // (every workgroup ends with up to 8 x 256 atomics on the same counters: against LDS atomics per key -- JPK_DIGITS_GRID, default 2048: 256 / 512 / 768 / 2048 / 4096 / 8192 / 16384 workgroups take 519 / 299 / 227 / 167 / 174 / 206 / 336 us)
            static const uint32_t dg = [] { const char *e = getenv("JPK_DIGITS_GRID"); const long v = e ? atol(e) : 2048; return (uint32_t)(v < 1 ? 1 : v); }();
            JPK_LAUNCH(ctx, PROF_RS_HIST, n, k_os_digits, dim3(ntiles < dg ? (ntiles ? ntiles : 1) : dg), dim3(RS_THREADS), keysA, n32, H, npass);
        }
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_os_prefix, dim3(1), dim3(256), H, gdig, npass);
        for (int p = 0; p < npass; p++) {
            const int shift = p < 7 ? 8 * (p + 1) : 0;
            OsArgs os;
            os.status = (p & 1) ? statusB : statusA;
            os.status_next = (p & 1) ? statusA : statusB;
            os.gdig = gdig + p * 256;
            os.ticket = tickets + p;
            os.tag_shift = tag_shift;
            os.slot_n = slot_n;
            if (p == 0) {
                launch_os_scatter<true>(ctx, keysA, reinterpret_cast<const uint32_t *>(slot_tag), keysB, valsB, n, shift, ntiles, os);
                continue;
            }
            launch_os_scatter<false>(ctx, ki, vi, ko, vo, n, shift, ntiles, os);
            uint64_t *tk = ki; ki = ko; ko = tk;
            uint32_t *tv = vi; vi = vo; vo = tv;
        }
        JPK_HIP(hipGetLastError());
        *keys_out = ki;
        *vals_out = vi;
        return JPK_OK;
    }
    for (int p = 0; p < npass; p++) {
        const int shift = p < 7 ? 8 * (p + 1) : 0;
        if (p == 0) {
            // (the first pass's tile histogram was counted by k_pack_keys, whose tiles are this sort's: hist = scratch, digit-major)
            JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
            if (rs_staged()) launch_rs_scatter_staged<true>(ctx, keysA, nullptr, keysB, valsB, n, shift, hist, ntiles);
            else JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, (k_rs_scatter<true>), dim3(ntiles), dim3(RS_THREADS), (const uint64_t *)keysA, (const uint32_t *)nullptr,
                       keysB, valsB, n, shift, hist, ntiles);
            continue;
        }
        JPK_LAUNCH(ctx, PROF_RS_HIST, n, k_rs_hist, dim3(ntiles), dim3(RS_THREADS), (const uint64_t *)ki, n, shift, hist, ntiles);
        JPK_TRY(jpk_exclusive_sum_u32(ctx, hist, hist, table, scan_scratch, nullptr));
        if (rs_staged()) launch_rs_scatter_staged<false>(ctx, ki, vi, ko, vo, n, shift, hist, ntiles);
        else JPK_LAUNCH(ctx, PROF_RS_SCATTER, n, (k_rs_scatter<false>), dim3(ntiles), dim3(RS_THREADS), (const uint64_t *)ki, (const uint32_t *)vi, ko, vo, n, shift, hist, ntiles);
        uint64_t *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
    JPK_HIP(hipGetLastError());
    *keys_out = ki;
    *vals_out = vi;
    return JPK_OK;
}
