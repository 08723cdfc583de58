// bwt_fwd.hip -- forward BWT on gfx950: GPU suffix-array construction (replaces divsufsort, divsufsort.cpp:1721)
// followed by the BWT image and the 120 sampled ranks of BlockSort::Bwt::ForwardBwt (bwt.cpp:22-65).
//
// Suffix array = prefix doubling (Larsson-Sadakane ranks) with compaction of resolved suffixes:
//   round 0   key = the first symbols of the suffix in an order-preserving code of the block's alphabet, big-endian in 56 bits, zero padded:
//             a VARIABLE-LENGTH prefix code built from the block's sampled histogram (k_key_plan, k_pack_keys_var: about 56 / H0 symbols per
//             key -- 12 for English-like text over 28 byte values, 10 over enwik8's 205 -- and every group of tied suffixes carries its own
//             depth, GD) or, for flat histograms / blocks above 2^28 bytes / the comparators, a fixed-width code (D = 7 bytes for more than
//             128 byte values, 11 for 17..32, up to 56: k_pack_keys).  One LSD radix sort of all n suffixes, 7 passes (radix.hip), fed in
//             descending text position so that a short suffix -- a proper prefix of anything it ties with on the padded bits -- comes
//             first: plain suffix order even when the text contains the smallest symbol.
//   round r   unresolved suffixes only.  The active list keeps groups of equal rank contiguous and in SA order, so a group is sorted by
//             key2 = rank[sa + h] + 1 (0 past the end) independently, h = the symbols its members are known to share (the group's depth;
//             D, 2D, 4D, ... with the fixed-width code):
//               * k_gather_win   key2 of every active suffix (the round's only random READ), head flags per 1024-slot window
//               * groups of <= 1024 suffixes: k_seg_round -- one workgroup owns the groups that start in its window,
//                 stages (sa, key2, group id) in LDS, LDS radix sort, re-ranks;
//               * larger groups: segmented LSD radix sort IN PLACE on (key2, sa), tiles = the pieces a group cuts out of
//                 the windows it crosses; per-group digit-major tables laid out in window order, so one flat exclusive
//                 scan gives every piece its offsets inside its own group;
//               * new ranks go to ISA (the round's only random WRITE); a suffix that has become a group of one is
//                 finished: its BWT byte T[sa - 1] is emitted at its final SA position and it leaves the list;
//               * compaction of the survivors (count / scan of tile totals / scatter).
//   pair round  (k_pair_*, instead of a doubling round when the list stops shrinking) long repeats are resolved by induction from their
//             successors -- what divsufsort's induced sorting does -- instead of log2(LCP) doubling rounds; see the comment at k_pair_dist.
//   One ISA buffer: all reads of a round (k_gather_win) complete before its first write (kernel boundary), which is the
//   condition under which parallel Larsson-Sadakane is exact.
// Host synchronisations (since round 5, JPK_SA_WAIT_ROUND = 1): the number of active suffixes lives in device memory (SaState) and every
// kernel reads it there, but the host waits for the 20-byte copy of the previous round's counts in front of EVERY round (hipEventSynchronize:
// a few microseconds while the other blocks in flight keep the GPU busy) and enqueues exactly the grids, the large-group passes and the
// pair rounds that round needs; round 6 adds one 4-byte read back in front of round 0's pack kernel (which code k_key_final chose).
// JPK_SA_WAIT_ROUND=3 keeps rounds 1 and 2 enqueued blind, one round behind the host's knowledge (rounds 2-4).
// Every array stays in HBM: T n, ISA 4n, BWT-in-SA-order n, radix ping-pong 24n (re-used by the rounds), active list 8n.
#include "common.hpp"
#include "prims.hpp"

using namespace jpk;

namespace {

constexpr int TB = 256;
constexpr int WAVES = TB / 64;
constexpr uint32_t DONE = 0x80000000u;
constexpr uint32_t NONE = 0xFFFFFFFFu;
// Bit 30 of a group rank in the active list (ranks are SA positions < n <= JPK_MAX_BLOCKSIZE < 2^30): the suffix starts inside a run
// of >= D equal bytes (D = round 0's key depth).  Such suffixes do not double their way through the run (log2(run / D) rounds, each over every member of the
// run: an all-zero 64 MiB block took 25 rounds): round 1 sorts their group by (does the run end in a smaller or a larger byte, run
// length) -- the complete order among suffixes that start with the same byte repeated, see k_gather_win -- and from round 2 on they
// compare at the END of their run (distance = remaining run length, uniform inside the group by then) instead of at distance h.
constexpr uint32_t RUNF = 0x40000000u;
static_assert((uint64_t)JPK_MAX_BLOCKSIZE < (1ull << 30), "bit 30 of a rank is free");
static_assert(JPK_FWD_BWT_LIMIT == (1u << 30) && (uint64_t)JPK_MAX_BLOCKSIZE < JPK_FWD_BWT_LIMIT, "jpk_fwd_bwt_device refuses what would need bit 30");

constexpr int CT = 4096;                   // slots per tile of the streaming kernels (count / scatter), 16 per thread
constexpr int CT_ITEMS = CT / TB;          // 16: slot(w, k, l) = tile * CT + w * 1024 + k * 64 + l  -> ballot = one 64-bit word
constexpr int SEG_TILE = 1024;             // a workgroup owns the groups that START in its SEG_TILE window
constexpr int SEG_SPAN = 2 * SEG_TILE;     // ... and therefore sees at most this many elements
constexpr int SEG_ITEMS = SEG_SPAN / TB;   // 8
constexpr int SEG_DBITS = 9;               // digit width of the LDS sort: (<= 31-bit rank, 10-bit local group) = at most 5 passes
constexpr int SEG_DIGITS = 1 << SEG_DBITS;
constexpr int WIN_ITEMS = SEG_TILE / TB;   // 4: slot(w, k, l) = window * 1024 + w * 256 + k * 64 + l
static_assert(SEG_DIGITS == 2 * TB, "two digits per thread in the digit scan");

// device-resident bookkeeping of one suffix sort (lives in the arena; the host reads it asynchronously)
struct SaState {
    uint32_t m[2];                         // unresolved suffixes: round r reads m[r & 1] and writes m[(r + 1) & 1]
    uint32_t npieces;                      // pieces of large groups in the current round
    uint32_t lc;                           // members of large groups in the current round
    uint32_t nrun;                         // unresolved suffixes after round 0 that start inside a run of >= depth equal bytes
    uint32_t pair_steps;                   // k_pair_repair: positions walked so far in this pair round (its work is capped at 8 n)
    uint32_t round_m[JPK_SA_MAX_ROUNDS];   // per round: unresolved suffixes when it starts
    uint32_t round_lc[JPK_SA_MAX_ROUNDS];  // per round: of those, members of groups > SEG_TILE
    // round 0's key (k_key_plan): the text's bytes renumbered 0..sigma-1 in byte order, `bits` bits each, `depth` of them in 56 bits
    uint32_t bits, depth;
    uint32_t vmode;                        // 0: that fixed-width code; 1 / 2 / 3: the variable-length code of order 0 / 1 / 2 (below; k_key_final)
    uint64_t rep;                          // the key field of "code 1 repeated depth times": code * rep = a run of that code
    uint32_t present[256];                 // byte value occurs in the text
    uint8_t lut[256];                      // byte -> code
    // variable-length keys (vmode, round 5): an order-preserving PREFIX code of the block's bytes -- weight-balanced on a sampled
    // histogram -- instead of the fixed-width one: a key holds as many symbols as fit its 56 bits (about 56 / H0: ten for enwik8's 205
    // byte values where the fixed code holds seven) and every group of tied suffixes carries its own depth (GD, see build_sa)
    uint32_t tag_shift, tag_max;           // a key's depth rides in the sorted value's bits from tag_shift up: at most tag_max (26 and 63 up to 2^26 bytes)
    uint32_t cnt[256];                     // sampled byte counts (k_sym_present: every sixteenth 16-byte vector)
    uint32_t vcode[256];                   // code of byte b, right-justified in vlen[b] bits
    uint8_t vlen[256];
    uint8_t vrun_d[256];                   // symbols of a run of byte b that one key holds: 56 / vlen[b]
    uint16_t vtop[256];                    // the byte whose code (of at most 8 bits) starts these 8 bits, 0xFFFF: none
    uint64_t vrunkey[256];                 // the 56-bit key of a run of byte b
    // vmode 2 (order-1 code): every symbol but a key's first is coded in the context of the byte in front of it (256 alphabetic codes, one
    // per context, from sampled pair counts -- k_pair_counts / k_ctx_plan); what k_key_final decides on:
    uint32_t v0_ok, sigma, v0_wl, v0_wtot; // the order-0 code is usable (no code above 27 bits); its weighted length and weight
    uint32_t o1_w, o1_wl, o1_maxlen;       // sampled pairs, their weighted length under the context codes, the longest context code
    // vmode 3 (order-2 code): a key's symbols from the third on are coded in the context of the TWO bytes in front of them when that pair
    // is one of the `nclass` <= 1024 most frequent ones (k_ctx_select; its own row of the code table), behind the one byte otherwise
    uint32_t nclass, o2_w, o2_wl, o2_maxlen;
};
static_assert(offsetof(SaState, vmode) == offsetof(SaState, round_m) + sizeof(uint32_t) * (2 * JPK_SA_MAX_ROUNDS + 2), "the statistics copy takes round_m, round_lc, bits, depth, vmode in one piece");

// one piece of a large group: the part of the group that lies inside one 1024-slot window of the active list
struct Piece {
    uint32_t begin, count;                 // slots [begin, begin + count) of the active list
    uint32_t gs, ge;                       // the group: slots [gs, ge)
    uint32_t fp, nt, tl, pad;              // index of the group's first piece, pieces in the group, this piece's ordinal
};

__device__ __forceinline__ uint64_t mask_below(int l) { return (1ull << l) - 1ull; }              // lanes < l
__device__ __forceinline__ uint64_t mask_upto(int l) { return (l >= 63) ? ~0ull : ((2ull << l) - 1ull); }   // lanes <= l
__device__ __forceinline__ uint32_t top_bit(uint64_t v) { return 63u - (uint32_t)__clzll((long long)v); }   // v != 0

// ---- single-workgroup scans over small per-tile / per-window arrays (1024 threads) -----------------------------------
// (256 threads x 32 items since round 6 -- rounds 1-5: 1024 x 8.  A workgroup of 1024 needs sixteen free wave slots on ONE CU at the same
// moment; among the blocks in flight of the timed loop these kernels waited 100-500 us for that -- k_win_scan1 211 us on average for 3.5 us
// of work, profiles/r05_kernel_stats_bench_loop.txt -- and every kernel behind them on the block's stream with them.)
constexpr int WG1 = 256;
constexpr int WG1_ITEMS = 32;
// out[i] = scan of in[0..i] (inclusive) or in[0..i-1] (exclusive) starting from `init`; REV walks the array backwards
// (suffix scan).  Returns the reduction of everything (all threads).  in == out is allowed.
template <class Op, bool EXCL, bool REV>
__device__ __forceinline__ uint32_t wg_scan(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t init, uint32_t *sm)
{
    uint32_t carry = init;
    for (uint32_t c0 = 0; c0 < n; c0 += WG1 * WG1_ITEMS) {
        const uint32_t i0 = c0 + threadIdx.x * WG1_ITEMS;
        uint32_t v[WG1_ITEMS];
        uint32_t acc = Op::id();
#pragma unroll
        for (int k = 0; k < WG1_ITEMS; k++) {
            const uint32_t i = i0 + k;
            v[k] = (i < n) ? in[REV ? n - 1 - i : i] : Op::id();
            acc = Op::f(acc, v[k]);
        }
        uint32_t tot;
        const uint32_t inc = block_incl_scan<Op>(acc, sm, &tot);
        uint32_t prev = __shfl_up(inc, 1, 64);
        if (lane_id() == 0) prev = (threadIdx.x == 0) ? Op::id() : sm[(threadIdx.x >> 6) - 1];
        uint32_t run = Op::f(carry, prev);
#pragma unroll
        for (int k = 0; k < WG1_ITEMS; k++) {
            const uint32_t i = i0 + k;
            uint32_t o;
            if (EXCL) { o = run; run = Op::f(run, v[k]); }
            else { run = Op::f(run, v[k]); o = run; }
            if (i < n) out[REV ? n - 1 - i : i] = o;
        }
        carry = Op::f(carry, tot);
        __syncthreads();                    // sm is reused by the next chunk; the stores above are visible to the workgroup
    }
    return carry;
}

// ---- round 0's keys ---------------------------------------------------------------------------------------------------
// The key of suffix i is its first `depth` bytes, each renumbered to its rank among the byte values that OCCUR in the text (an
// order-preserving code of `bits` = ceil(log2 sigma) bits), big-endian in the 56 key bits, zero padded past the end of the text: text
// over 28 letters packs 11 bytes where the plain form held 7, DNA 28, and the doubling rounds start at that distance -- on the
// enwik8-like block round 1 starts with 70 % of the suffixes instead of 93 %, round 2 with 12 % instead of 52 %.  More than 128 byte
// values (binary data, real enwik8: 205): bits = 8, depth = 7, the keys of round 3.  Everything downstream only relies on "equal key =
// equal first `depth` bytes" and "code 0 is the smallest"; a byte past the end packs as 0 like the smallest code, and the stable sort
// fed in descending position puts the shorter suffix first, as before.
__global__ __launch_bounds__(TB) void k_sym_present(const uint8_t *__restrict__ T, uint32_t n, SaState *__restrict__ st)
{
    __shared__ uint32_t f[256];
    __shared__ uint32_t c[256];           // sampled counts: every sixteenth vector (the variable-length code is built from them)
    f[threadIdx.x] = 0u;
    c[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t mis0 = (uint32_t)((16u - ((uintptr_t)T & 15u)) & 15u), mis = mis0 < n ? mis0 : n;
    const uint4 *V = reinterpret_cast<const uint4 *>(T + mis);
    const uint32_t nv = (n - mis) / 16u, tail0 = mis + nv * 16u;
    if (blockIdx.x == 0) {                                            // the unaligned head and the tail, a few bytes
        if (threadIdx.x < mis) f[T[threadIdx.x]] = 1u;
        if (tail0 + threadIdx.x < n) f[T[tail0 + threadIdx.x]] = 1u;
    }
    // Four vectors per thread in flight, and the occurring bytes collected in registers (four 64-bit words per thread, merged into the
    // LDS flags once at the end): an LDS store per byte ran into bank conflicts -- 207 different addresses over 32 banks -- and cost
    // 110 us for 64 MiB over enwik8's alphabet, 54 us for the 28-letter text.
    const uint32_t step = gridDim.x * TB;
    uint64_t seen[4] = {0ull, 0ull, 0ull, 0ull};
    for (uint32_t v0 = blockIdx.x * TB + threadIdx.x; v0 < nv; v0 += 4u * step) {
        uint4 xs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t v = v0 + (uint32_t)u * step; xs[u] = v < nv ? V[v] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t v = v0 + (uint32_t)u * step;
            if (v >= nv) break;
            const uint32_t ws[4] = {xs[u].x, xs[u].y, xs[u].z, xs[u].w};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int b = 0; b < 4; b++) {                                       // the thread's own 256-bit set, in registers
                    const uint32_t y = (ws[q] >> (8 * b)) & 255u;
                    const uint64_t bit = 1ull << (y & 63u);
                    const uint32_t hi = y >> 6;
                    seen[0] |= hi == 0u ? bit : 0ull;
                    seen[1] |= hi == 1u ? bit : 0ull;
                    seen[2] |= hi == 2u ? bit : 0ull;
                    seen[3] |= hi == 3u ? bit : 0ull;
                }
            if ((v & 15u) == 0u) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int b = 0; b < 4; b++) atomicAdd(&c[(ws[q] >> (8 * b)) & 255u], 1u);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {                                                   // wave-wide OR, then one lane per set bit stores a flag
        uint64_t m = seen[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m |= (uint64_t)__shfl_xor((unsigned long long)m, o, 64);
        if ((m >> (threadIdx.x & 63u)) & 1ull) f[64 * k + (threadIdx.x & 63u)] = 1u;
    }
    __syncthreads();
    if (f[threadIdx.x]) st->present[threadIdx.x] = 1u;
    if (c[threadIdx.x]) atomicAdd(&st->cnt[threadIdx.x], c[threadIdx.x]);
}

// the weight-balanced splitting of the occurring bytes (cpre = exclusive prefix of their weights, in byte order, cpre[sigma] = total):
// symbol idx walks from the root to its own leaf -- at every node the range [l, r) is cut where the weight is halved, left = 0, right = 1.
// An alphabetic (order-preserving) prefix code with an average length below H + 2.
__device__ __forceinline__ void wb_walk(const uint32_t *cpre, uint32_t idx, uint32_t sigma, uint32_t &code, uint32_t &len)
{
    uint32_t l = 0, r = sigma;
    code = 0; len = 0;
    while (r - l > 1u) {
        const uint64_t tgt2 = (uint64_t)cpre[l] + cpre[r];            // twice the weight at which [l, r) is halved
        uint32_t lo = l + 1u, hi = r - 1u;                            // the cut m lies in [l + 1, r - 1]: first index with 2 cpre[m] >= tgt2
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (2ull * cpre[mid] >= tgt2) hi = mid; else lo = mid + 1u;
        }
        uint32_t m = lo;
        if (m - 1u > l) {
            const uint64_t a = 2ull * cpre[m], b = 2ull * cpre[m - 1u];
            const uint64_t da = a > tgt2 ? a - tgt2 : tgt2 - a, db = b > tgt2 ? b - tgt2 : tgt2 - b;
            if (db < da) m--;
        }
        if (idx < m) { r = m; code <<= 1; } else { l = m; code = (code << 1) | 1u; }
        len++;
        if (len > 30u) break;
    }
    if (len == 0u) len = 1u;                                          // one byte value: the code is "0"
}

// Sampled pair counts for the order-1 code (vmode 2): ctab[c * 256 + s] += occurrences of byte s behind byte c inside the sampled 16-byte
// vectors (every `stride`th: about a million pairs whatever the block).  Workgroup (x, y) counts the pairs whose context byte has low
// nibble y -- 16 KB of LDS counters; text's letters spread over all sixteen -- and adds what it found to the table.
__global__ __launch_bounds__(256) void k_pair_counts(const uint8_t *__restrict__ T, uint32_t n, uint32_t stride, uint32_t *__restrict__ ctab)
{
    __shared__ uint32_t c[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256; i += 256) c[i] = 0u;
    __syncthreads();
    const uint32_t mis0 = (uint32_t)((16u - ((uintptr_t)T & 15u)) & 15u), mis = mis0 < n ? mis0 : n;
    const uint4 *V = reinterpret_cast<const uint4 *>(T + mis);
    const uint32_t nv = (n - mis) / 16u, ns = (nv + stride - 1u) / stride, y = blockIdx.y;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < ns; k += gridDim.x * 256u) {
        const uint4 x = V[(size_t)k * stride];
        const uint32_t ws[4] = {x.x, x.y, x.z, x.w};
        uint32_t prev = ws[0] & 255u;
#pragma unroll
        for (int j = 1; j < 16; j++) {
            const uint32_t b = (ws[j >> 2] >> (8 * (j & 3))) & 255u;
            if ((prev & 15u) == y) atomicAdd(&c[(prev >> 4) * 256u + b], 1u);
            prev = b;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 256; i += 256)
        if (c[i]) atomicAdd(&ctab[((((uint32_t)i >> 8) << 4) | y) * 256u + ((uint32_t)i & 255u)], c[i]);
}

// Order-2 contexts: the (at most) JPK_O2_CLASSES most frequent byte pairs of the sample get a row of their own in the code table -- the
// largest count threshold that admits no more than that many pairs, found by bisection over the 65 536 pair counts (one workgroup of
// 256, 256 counts per thread in registers).  ctxmap[c2 << 8 | c1] = the row that codes a symbol behind the bytes c2 c1: 256 + the pair's
// rank among the chosen ones, or c1 -- the order-1 row -- for every other pair.
constexpr uint32_t JPK_O2_CLASSES = 1024;
static_assert(JPK_O2_CLASSES <= 1024 && 256 + JPK_O2_CLASSES <= 65535, "k_triple_counts keeps 64 rows of counters per group of sixteen; ctxmap holds rows in 16 bits");
__global__ __launch_bounds__(256) void k_ctx_select(const uint32_t *__restrict__ ctab, uint16_t *__restrict__ ctxmap, SaState *__restrict__ st)
{
    // (one workgroup of 256 threads, 256 counts each, since round 6: the 1024-thread form waited 490 us on average in the timed loop for
    // sixteen free wave slots on one CU -- 38 us alone)
    constexpr int NT = 256, NR = 65536 / NT / 2;                          // 128 registers of two counts
    __shared__ uint32_t sm[NT / 64 + 1];
    uint32_t c[NR];                                                       // two counts per register, saturated at 65 535 (the order at the threshold is what matters)
#pragma unroll
    for (int k = 0; k < NR / 2; k++) {
        const uint4 v = reinterpret_cast<const uint4 *>(ctab)[threadIdx.x * (NR / 2) + k];
        c[2 * k] = (v.x < 65535u ? v.x : 65535u) | ((v.y < 65535u ? v.y : 65535u) << 16);
        c[2 * k + 1] = (v.z < 65535u ? v.z : 65535u) | ((v.w < 65535u ? v.w : 65535u) << 16);
    }
    uint32_t lo = 1u, hi = 65536u;
    while (lo < hi) {                                                     // smallest threshold with at most JPK_O2_CLASSES pairs at or above it
        const uint32_t mid = (lo + hi) >> 1;
        uint32_t mine = 0, tot;
#pragma unroll
        for (int k = 0; k < NR; k++) mine += ((c[k] & 0xFFFFu) >= mid ? 1u : 0u) + ((c[k] >> 16) >= mid ? 1u : 0u);
        block_incl_scan<OpSum>(mine, sm, &tot);
        if (tot <= JPK_O2_CLASSES) hi = mid; else lo = mid + 1u;
    }
    uint32_t mine = 0, tot;
#pragma unroll
    for (int k = 0; k < NR; k++) mine += ((c[k] & 0xFFFFu) >= lo ? 1u : 0u) + ((c[k] >> 16) >= lo ? 1u : 0u);
    uint32_t rank = block_incl_scan<OpSum>(mine, sm, &tot) - mine;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const uint32_t i = threadIdx.x * (2u * NR) + 2u * k;
        const uint32_t r0 = (c[k] & 0xFFFFu) >= lo ? 256u + rank++ : (i & 255u);
        const uint32_t r1 = (c[k] >> 16) >= lo ? 256u + rank++ : ((i + 1u) & 255u);
        reinterpret_cast<uint32_t *>(ctxmap)[i >> 1] = r0 | (r1 << 16);
    }
    if (threadIdx.x == 0) st->nclass = tot;
}
// ... and the sampled counts of the bytes behind the chosen pairs, into their rows: workgroup (x, y) counts for the rows 256 + r with
// r mod 16 = y (at most 64 of them: 64 KB of LDS counters, one workgroup per CU -- the kernel is short and reads 1 MiB)
__global__ __launch_bounds__(256) void k_triple_counts(const uint8_t *__restrict__ T, uint32_t n, uint32_t stride, const uint16_t *__restrict__ ctxmap,
                                                      uint32_t *__restrict__ ctab)
{
    __shared__ uint32_t c[64 * 256];
    for (int i = threadIdx.x; i < 64 * 256; i += 256) c[i] = 0u;
    __syncthreads();
    const uint32_t mis0 = (uint32_t)((16u - ((uintptr_t)T & 15u)) & 15u), mis = mis0 < n ? mis0 : n;
    const uint4 *V = reinterpret_cast<const uint4 *>(T + mis);
    const uint32_t nv = (n - mis) / 16u, ns = (nv + stride - 1u) / stride, y = blockIdx.y;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < ns; k += gridDim.x * 256u) {
        const uint4 x = V[(size_t)k * stride];
        const uint32_t ws[4] = {x.x, x.y, x.z, x.w};
        uint32_t p2 = ws[0] & 255u, p1 = (ws[0] >> 8) & 255u;
#pragma unroll
        for (int j = 2; j < 16; j++) {
            const uint32_t b = (ws[j >> 2] >> (8 * (j & 3))) & 255u;
            const uint32_t row = ctxmap[(p2 << 8) | p1];
            if (row >= 256u && ((row - 256u) & 15u) == y) atomicAdd(&c[((row - 256u) >> 4) * 256u + b], 1u);
            p2 = p1;
            p1 = b;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 256; i += 256)
        if (c[i]) atomicAdd(&ctab[(256u + ((((uint32_t)i >> 8) << 4) | y)) * 256u + ((uint32_t)i & 255u)], c[i]);
}

// one workgroup of 256: code of every byte value, bits per code, bytes per key.  force_bits: 0 = from the alphabet, 8 = plain bytes.
// want_var: build the variable-length order-0 code as well -- the weight-balanced splitting of the occurring bytes on the sampled
// histogram (wb_walk): an average length below H0 + 2 (5.3 bits on an enwik8-like alphabet whose H0 is 5.05; Hu-Tucker's optimum is
// 5.2).  A code longer than 27 bits (it cannot happen with sampled weights + 1) leaves v0_ok = 0: the fixed-width code.  Which code the
// keys use is k_key_final's decision.
__global__ __launch_bounds__(256) void k_key_plan(SaState *__restrict__ st, int force_bits, int want_var)
{
    __shared__ uint32_t sm[256 / 64 + 1];
    __shared__ uint32_t cpre[257];             // exclusive prefix of the weights of the occurring bytes, in byte order
    const uint32_t here = st->present[threadIdx.x] ? 1u : 0u;
    uint32_t sigma;
    const uint32_t inc = block_incl_scan<OpSum>(here, sm, &sigma);
    const uint32_t idx = inc - here;
    st->lut[threadIdx.x] = (uint8_t)idx;
    if (threadIdx.x == 0) {
        uint32_t bits = 1;
        while ((1u << bits) < sigma) bits++;
        if (force_bits > 0 && (uint32_t)force_bits > bits) bits = (uint32_t)force_bits;
        const uint32_t depth = 56u / bits;
        uint64_t rep = 0;
        for (uint32_t k = 0; k < depth; k++) rep |= 1ull << (56u - bits * (k + 1u));
        st->bits = bits;
        st->depth = depth;
        st->rep = rep;
        st->sigma = sigma;
    }
    if (!want_var || force_bits > 0) return;          // (uniform)
    const uint32_t w = here ? st->cnt[threadIdx.x] + 1u : 0u;
    uint32_t wtot;
    const uint32_t winc = block_incl_scan<OpSum>(w, sm, &wtot);
    if (here) cpre[idx] = winc - w;
    if (threadIdx.x == 0) cpre[sigma] = wtot;
    __syncthreads();
    uint32_t code = 0, len = 0;
    if (here) wb_walk(cpre, idx, sigma, code, len);
    uint32_t maxlen;
    block_incl_scan<OpMax>(len, sm, &maxlen);
    uint32_t wl;                                                          // sum of weight x length: the average code length
    // (32-bit: weights are sampled, at most n / 16 + 256 in all, lengths <= 30 -- below 2^32 for blocks of 2^26 bytes)
    block_incl_scan<OpSum>(w * len, sm, &wl);
    st->vcode[threadIdx.x] = code;
    st->vlen[threadIdx.x] = (uint8_t)len;
    if (threadIdx.x == 0) {
        st->v0_ok = maxlen <= 27u ? 1u : 0u;                              // (the pack kernel keeps code | length << 27 in one word)
        st->v0_wl = wl;
        st->v0_wtot = wtot;
    }
}

// Context codes: workgroup c turns row c of the sampled counts into the code of every byte BEHIND byte c (rows 0..255) or behind the
// c - 256-th chosen pair of bytes (k_ctx_select) -- the same weight-balanced
// splitting over all occurring bytes, weight = 8 x sampled count + a floor (a pair the sample missed still needs a code; the floor
// grows with the context's count so that no code exceeds ~22 bits) -- in place: ctab[c * 256 + s] = code | length << 27.
__global__ __launch_bounds__(256) void k_ctx_plan(SaState *__restrict__ st, uint32_t *__restrict__ ctab)
{
    __shared__ uint32_t sm[256 / 64 + 1];
    __shared__ uint32_t cpre[257];
    const uint32_t c = blockIdx.x;                                        // the row: a byte (order 1) or 256 + a chosen pair (order 2)
    if (c < 256u ? !st->present[c] : c - 256u >= st->nclass) return;      // (uniform) never a context
    const uint32_t here = st->present[threadIdx.x] ? 1u : 0u;
    const uint32_t raw = here ? ctab[c * 256u + threadIdx.x] : 0u;
    uint32_t sigma, nc;
    const uint32_t inc = block_incl_scan<OpSum>(here, sm, &sigma);
    const uint32_t idx = inc - here;
    block_incl_scan<OpSum>(raw, sm, &nc);
    const uint32_t w = here ? raw * 8u + 1u + ((8u * nc) >> 20) : 0u;
    uint32_t wtot;
    const uint32_t winc = block_incl_scan<OpSum>(w, sm, &wtot);
    if (here) cpre[idx] = winc - w;
    if (threadIdx.x == 0) cpre[sigma] = wtot;
    __syncthreads();
    uint32_t code = 0, len = 0;
    if (here) wb_walk(cpre, idx, sigma, code, len);
    uint32_t maxlen, wl;
    block_incl_scan<OpMax>(len, sm, &maxlen);
    block_incl_scan<OpSum>(raw * len, sm, &wl);
    ctab[c * 256u + threadIdx.x] = here ? (code | (len << 27)) : 0u;      // (every thread has read its count: the scans' barriers lie in between)
    if (threadIdx.x == 0) {
        atomicAdd(c < 256u ? &st->o1_w : &st->o2_w, nc);
        atomicAdd(c < 256u ? &st->o1_wl : &st->o2_wl, wl);
        atomicMax(c < 256u ? &st->o1_maxlen : &st->o2_maxlen, maxlen);
    }
}

// one workgroup of 256 decides which code round 0's keys use and leaves the tables of the choice in the state:
//   vmode 0  the fixed-width code (k_pack_keys), when no variable-length code buys at least 3/4 of a symbol per key over it (near-uniform
//            alphabets: random bytes, DNA, -- a balanced code of a flat histogram IS the fixed code, a slightly skewed one can even be longer)
//   vmode 1  the order-0 code
//   vmode 2  the order-1 code, when an average key holds at least half a symbol more with it: 1 + (56 - len0) / len1 against 56 / len0
//   vmode 3  the order-2 code, when it holds half a symbol more again: 2 + (56 - 2 len0) / len2
// and, for vmode 1 / 2: the 56-bit key of a run of byte b (b, then b behind b, ...), the whole symbols in it, the table that finds the
// first byte of a key from its first 8 bits, the depth tag's place in the sorted value.
__global__ __launch_bounds__(256) void k_key_final(SaState *__restrict__ st, const uint32_t *__restrict__ ctab, const uint16_t *__restrict__ ctxmap, int tag_shift,
                                                   int want_order)
{
    if (!st->v0_ok) return;                                               // (uniform) vmode stays 0
    const uint32_t sigma = st->sigma;
    uint32_t fb = 1;
    while ((1u << fb) < sigma) fb++;
    const float fixed_d = (float)(56u / fb);
    const float a0 = (float)st->v0_wl / (float)st->v0_wtot, d0 = 56.f / a0;
    float d1 = 0.f, d2 = 0.f, a1 = 0.f;
    const bool o1ok = want_order >= 1 && ctab && st->o1_w > 0u && st->o1_maxlen <= 27u && st->o1_wl > 0u;
    if (o1ok) { a1 = (float)st->o1_wl / (float)st->o1_w; d1 = 1.f + (56.f - a0) / a1; }
    // order 2: the sampled triples behind a chosen pair are coded in its row, the others in the order-1 rows (14 triples to 15 pairs per sampled vector)
    const bool o2ok = o1ok && want_order >= 2 && ctxmap && st->nclass > 0u && st->o2_w > 0u && st->o2_wl > 0u && st->o2_maxlen <= 27u;
    if (o2ok) {
        float cov = (float)st->o2_w / ((float)st->o1_w * (14.f / 15.f));
        cov = cov > 1.f ? 1.f : cov;
        const float a2 = cov * ((float)st->o2_wl / (float)st->o2_w) + (1.f - cov) * a1;
        d2 = 2.f + (56.f - 2.f * a0) / a2;
    }
    uint32_t mode;
    if (o2ok && d2 >= fixed_d + 0.75f && d2 >= d1 + 0.5f && d2 >= d0 + 0.5f) mode = 3u;
    else if (o1ok && d1 >= fixed_d + 0.75f && d1 >= d0 + 0.5f) mode = 2u;
    else if (d0 >= fixed_d + 0.75f) mode = 1u;
    else return;
    const uint32_t code = st->vcode[threadIdx.x], len = st->vlen[threadIdx.x];
    const bool here = st->present[threadIdx.x] != 0u;
    if (here) {
        uint32_t c1 = code, l1 = len, c2 = code, l2 = len;                // the code of b behind b, and of b behind b b
        if (mode >= 2u) { const uint32_t e = ctab[threadIdx.x * 256u + threadIdx.x]; c1 = c2 = e & 0x7FFFFFFu; l1 = l2 = e >> 27; }
        if (mode == 3u) {                                                 // (order 2: a key's second symbol is in the order-0 code, see k_pack_keys_o2)
            const uint32_t e = ctab[(uint32_t)ctxmap[threadIdx.x * 257u] * 256u + threadIdx.x];
            c2 = e & 0x7FFFFFFu; l2 = e >> 27; c1 = code; l1 = len;
        }
        uint64_t k = code;
        uint32_t used = len, d = 1;
        for (;;) {                                                        // whole symbols while they fit, then the start of one more
            const uint32_t cn = d == 1u ? c1 : c2, ln = d == 1u ? l1 : l2;
            if (used + ln <= 56u) { k = (k << ln) | cn; used += ln; d++; if (used == 56u) break; }
            else { const uint32_t left = 56u - used; k = (k << left) | (uint64_t)(cn >> (ln - left)); break; }
        }
        st->vrun_d[threadIdx.x] = (uint8_t)d;
        st->vrunkey[threadIdx.x] = k;
    }
    {   // the byte whose (order-0) code, of at most 8 bits, starts the 8 bits x: byte b fills the range its code spans (prefix-free: disjoint)
        __shared__ uint16_t vt[256];
        vt[threadIdx.x] = (uint16_t)0xFFFFu;
        __syncthreads();
        if (here && len <= 8u)
            for (uint32_t x = code << (8u - len); x < (code + 1u) << (8u - len); x++) vt[x] = (uint16_t)threadIdx.x;
        __syncthreads();
        st->vtop[threadIdx.x] = vt[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        st->vmode = mode;
        st->tag_shift = (uint32_t)tag_shift;
        st->tag_max = (tag_shift <= 26 || tag_shift >= 32) ? 63u : ((1u << (32 - tag_shift)) - 1u);      // (32: the depths stay in their own array, see r0_short)
        const uint32_t avg_d = (uint32_t)(mode == 3u ? d2 : mode == 2u ? d1 : d0);   // symbols an average key holds (the statistics' key depth)
        st->depth = avg_d ? avg_d : 1u;
    }
}

// P[j] = key of slot j = suffix n-1-j (the order round 0's radix sort is fed in), low byte = T[i-1] (0 for suffix 0) -- the suffix's
// BWT byte rides through the sort -- or, in a group sort, the number of the suffix's block, which is the sort's last digit; a suffix
// ends with its block there.  One tile = 4096 slots = 4096 consecutive text positions: their bytes, codes and block numbers are
// staged in LDS (index q = position i_lo - 16 + q, so that a thread's sixteen positions are one aligned 16-byte read), every thread
// rolls the window over its sixteen positions (one shift and one code per position), the keys are turned into slot order in LDS
// and leave in whole lines.
static_assert(CT == 4096, "k_pack_keys counts the tile histogram of the radix sort's first pass: CT must be radix.hip's RS_TILE");
constexpr int PK_HALO = 80;                 // 16 positions in front (the byte before the tile), up to 55 + 9 behind
constexpr int PK_KO = CT + CT / 16;         // one spare word per sixteen keys: the threads' 128-byte strides fall on different banks
template <int B>
__device__ __forceinline__ void pack_tile(const uint8_t *cc, const uint8_t *cr, const uint8_t *cb, uint64_t *ko, int64_t i_lo, uint32_t n,
                                          const uint32_t *__restrict__ bend)
{
    constexpr int D = 56 / B;
    constexpr int NV = (16 + D + 15) / 16;   // the codes of positions [0, 16 + D) of the thread's stretch, in 16-byte reads
    constexpr uint64_t M56 = (1ull << 56) - 1ull;
    const int t = threadIdx.x;
    uint32_t cw[NV * 4], rw[4], bw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const uint4 x = reinterpret_cast<const uint4 *>(cc + 16 + 16 * t)[v];
        cw[4 * v] = x.x; cw[4 * v + 1] = x.y; cw[4 * v + 2] = x.z; cw[4 * v + 3] = x.w;
    }
    {
        const uint4 x = *reinterpret_cast<const uint4 *>(cr + 16 + 16 * t);
        rw[0] = x.x; rw[1] = x.y; rw[2] = x.z; rw[3] = x.w;
    }
    if (bend) {
        const uint4 x = *reinterpret_cast<const uint4 *>(cb + 16 + 16 * t);
        bw[0] = x.x; bw[1] = x.y; bw[2] = x.z; bw[3] = x.w;
    }
    const uint32_t before = cr[15 + 16 * t];                          // the byte in front of the thread's first position
#define JPK_BYTE(A, X) (((A)[(X) >> 2] >> (((X) & 3) * 8)) & 255u)
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < D; k++) v = (v << B) | JPK_BYTE(cw, k);
    v <<= 56 - B * D;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int64_t i = i_lo + 16 * t + s;
        if (i >= 0) {
            const uint32_t blkno = JPK_BYTE(bw, s);
            const uint32_t left = (bend ? bend[blkno] : n) - (uint32_t)i;               // bytes left in the suffix (its own block), >= 1
            const uint64_t vm = (left < (uint32_t)D) ? v & ~((1ull << (56u - (uint32_t)B * left)) - 1ull) : v;
            const uint32_t prev = s ? JPK_BYTE(rw, s ? s - 1 : 0) : before;
            const uint32_t low = bend ? blkno : (i ? prev : 0u);
            const uint32_t x = (uint32_t)(CT - 1 - 16 * t - s);
            ko[x + (x >> 4)] = (vm << 8) | low;
        }
        if (s < 15) v = ((v << B) & M56) | ((uint64_t)JPK_BYTE(cw, s + D) << (56 - B * D));
    }
#undef JPK_BYTE
}
// Variable-length keys (vmode 1 / 2): the same tile, every byte as its prefix code -- the FIRST symbol of a key in the order-0 code, every
// other one in the order-0 code as well (vmode 1) or in the code of its context, the byte in front of it (vmode 2, O1; k_ctx_plan's table,
// 256 KB, read through the caches).  Two suffixes that share k symbols read symbol k + 1 in the same context, so the concatenated codes
// compare like the suffixes.  Symbols past the end of the text (group sort: of the suffix's own block) are zero bits and do not count: a
// suffix whose depth reaches its end is shorter than anything it ties with and becomes a group of its own (k_r0_*).  D0[x] = depth of slot
// x = the WHOLE symbols in its key (< 64: rides in bits 26..31 of the slot's value through the radix sort).
// History of the kernel (round 5): (1) every thread rolled over SIXTEEN consecutive positions inside k_pack_keys -- a serial chain with two
// dependent LDS lookups per step, keys turned into slot order through 35 KB of LDS: 0.44 ms; (2) every position from scratch: ~260 vector
// instructions per position, 0.46-0.53 ms; (3) the key of the last of four consecutive positions from scratch, the other three by rolling
// backwards: 0.29 ms; (4, this form, which the order-1 code needs: a key from scratch would be ~19 table reads from memory) no key is
// built from scratch: a thread looks up the codes of ITS four symbols once and leaves their concatenation -- bits, codeword-end marks,
// length -- as one chunk in LDS; the code string behind its last position is the concatenation of the following chunks (four or five
// 8-byte reads until 64 bits are full), and it rolls backwards over its own four positions in registers:
//     R(i) = code(T[i+1] | T[i]) in front of R(i+1) >> its length,   key(i) = code0(T[i]) in front of R(i),
//     depth(i) = 1 + the codeword ends among the first 56 - len0 bits of R(i)     (a population count of the marks).
// The four keys are four consecutive slots and leave as two 16-byte stores, the four depths as one word.
constexpr int PV_NCH = (CT + 64) / 4;       // chunks of four staged positions q = 16 + 4 j + k: the tile's 1024 and 16 behind it
template <bool O1>
__global__ __launch_bounds__(TB) void k_pack_keys_var(const uint8_t *__restrict__ T, uint32_t n, const SaState *__restrict__ st, uint64_t *__restrict__ P,
                                                     const uint8_t *__restrict__ blk, const uint32_t *__restrict__ bend, uint8_t *__restrict__ D0,
                                                     const uint32_t *__restrict__ ctab)
{
    if (st->vmode != (O1 ? 2u : 1u)) return;                          // (the plan chose another code: its kernel does the tiles)
    __shared__ __align__(16) uint8_t cr[CT + PK_HALO];                // bytes: index q = position i_lo - 16 + q
    __shared__ __align__(16) uint8_t cb[CT + PK_HALO];                // block numbers (group sort)
    __shared__ uint32_t lcl[256];                                      // order-0: code | length << 27 (k_key_plan keeps codes below 28 bits)
    __shared__ uint64_t CB[PV_NCH], CM[PV_NCH];                        // a chunk's code bits, left-aligned, and the marks of its codeword ends
    __shared__ uint8_t CL[PV_NCH];                                     // its bits (<= 64) | 0x80: nothing follows it (the text or the block ends inside)
    lcl[threadIdx.x] = st->vcode[threadIdx.x] | ((uint32_t)st->vlen[threadIdx.x] << 27);
    const uint32_t tag_max = st->tag_max;
    const uint32_t ntiles = (n + CT - 1) / CT;
    const int t = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT, cnt = (n - base < (uint32_t)CT) ? n - base : (uint32_t)CT;
        const int64_t i_lo = (int64_t)n - 1 - base - (CT - 1);        // position of the tile's LAST slot (negative in the last tile: no such slot)
        __syncthreads();                                                // the table; the previous tile's bytes and chunks have been read
        for (int q = threadIdx.x; q < CT + PK_HALO; q += TB) {
            const int64_t p = i_lo - 16 + q;
            const bool in = p >= 0 && p < (int64_t)n;
            cr[q] = in ? T[p] : (uint8_t)0;
            if (blk) cb[q] = in ? blk[p] : (uint8_t)0;
        }
        __syncthreads();
        // chunk j: the codes of the symbols at q = 16 + 4 j .. + 3, each behind its predecessor.  e[k] = the symbol's table entry, 0 where
        // it cannot follow its predecessor (past the end, the first position of another block, position 0); the chunk as others see it stops there.
        auto chunk = [&](int j, uint32_t (&e)[4]) {
            uint64_t bits = 0, marks = 0;
            uint32_t used = 0;
            bool stop = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = 16 + 4 * j + k;
                const int64_t p = i_lo - 16 + q;
                const bool ok = p >= 1 && p < (int64_t)n && !(blk && cb[q] != cb[q - 1]);
                e[k] = ok ? (O1 ? ctab[(uint32_t)cr[q - 1] * 256u + cr[q]] : lcl[cr[q]]) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (e[k] == 0u) stop = true;
                if (!stop && used < 64u) {
                    const uint32_t l = e[k] >> 27, c = e[k] & 0x7FFFFFFu;
                    if (used + l <= 64u) { used += l; bits |= (uint64_t)c << (64u - used); marks |= 1ull << (64u - used); }
                    else { bits |= (uint64_t)c >> (used + l - 64u); used = 64u; }
                }
            }
            CB[j] = bits;
            CM[j] = marks;
            CL[j] = (uint8_t)(used | (stop ? 0x80u : 0u));
        };
        if (t < PV_NCH - CT / 4) { uint32_t e[4]; chunk(CT / 4 + t, e); }      // the sixteen chunks behind the tile
#pragma unroll 1
        for (int it = 0; it < CT / (TB * 4); it++) {
            // chunks in descending order: what a position needs behind it has been built in the steps before
            const int j = CT / 4 - 1 - (it * TB + t);
            const uint32_t x0 = (uint32_t)((it * TB + t) * 4);         // my four slots x0 .. x0 + 3 = positions i_hi, i_hi - 1, ..: chunk j, backwards
            uint32_t e[4];
            chunk(j, e);
            __syncthreads();
            if (x0 >= cnt) continue;
            const int q_hi = 16 + (CT - 1) - (int)x0;                  // = 16 + 4 j + 3
            uint64_t R = 0, M = 0, key[4];
            uint32_t used = 0, dep = 0;
            for (int jj = j + 1; jj < PV_NCH; jj++) {
                const uint32_t L = CL[jj];
                R |= CB[jj] >> used;
                M |= CM[jj] >> used;
                used += L & 0x7Fu;
                if ((L & 0x80u) || used >= 64u) break;
            }
#pragma unroll
            for (int jx = 0; jx < 4; jx++) {
                const int q = q_hi - jx;
                const int64_t i = i_lo - 16 + q;
                if (jx) {                                              // one position back: the code of the symbol behind it goes in front
                    const uint32_t en = e[4 - jx];
                    if (en == 0u) { R = 0; M = 0; }                    // nothing follows (the last position of its block)
                    else {
                        const uint32_t l = en >> 27;
                        R = ((uint64_t)(en & 0x7FFFFFFu) << (64u - l)) | (R >> l);
                        M = (1ull << (64u - l)) | (M >> l);
                    }
                }
                const uint32_t e0 = lcl[cr[q]], l0 = e0 >> 27;
                const uint64_t k56 = ((uint64_t)(e0 & 0x7FFFFFFu) << (56u - l0)) | (R >> (8u + l0));
                const uint32_t low = bend ? cb[q] : (i > 0 ? cr[q - 1] : 0u);   // T[i - 1] rides in the low byte -- the block number in a group sort (the sort's last digit)
                key[jx] = (k56 << 8) | low;
                const uint32_t d = 1u + (uint32_t)__popcll(M >> (8u + l0));
                dep |= (d < tag_max ? d : tag_max) << (8 * jx);        // (a clamped depth is still a number of symbols the key's group shares)
            }
            if (x0 + 4 <= cnt) {
                uint4 *o = reinterpret_cast<uint4 *>(P + base + x0);
                o[0] = make_uint4((uint32_t)key[0], (uint32_t)(key[0] >> 32), (uint32_t)key[1], (uint32_t)(key[1] >> 32));
                o[1] = make_uint4((uint32_t)key[2], (uint32_t)(key[2] >> 32), (uint32_t)key[3], (uint32_t)(key[3] >> 32));
                *reinterpret_cast<uint32_t *>(D0 + base + x0) = dep;
            } else {
#pragma unroll
                for (int jx = 0; jx < 4; jx++)
                    if (x0 + jx < cnt) { P[base + x0 + jx] = key[jx]; D0[base + x0 + jx] = (uint8_t)(dep >> (8 * jx)); }
            }
        }
        // the slots between the text's end and the end of the last tile: the largest key, so that the radix sort (stable, fed whole
        // tiles since round 6) leaves them behind every suffix; nothing reads them afterwards
        for (uint32_t x = cnt + threadIdx.x; x < (uint32_t)CT; x += TB) { P[base + x] = ~0ull; D0[base + x] = 0; }
    }
}

// vmode 3: the order-2 code.  key(i) = code0(T[i]), code0(T[i+1]), then U(i + 2) with U(q) = the codes of the symbols q, q + 1, ..
// each behind the two bytes in front of it (its row from ctxmap: a chosen pair's own, or the order-1 row of the one byte) -- a string that
// does not depend on where the key starts, so it rolls: U(q) = entry(q) in front of U(q + 1) >> its length.  A thread owns four positions
// q0 .. q0 + 3 and looks up the four symbols q0 + 2 .. q0 + 5 -- those whose contexts START at its positions: their concatenation is its
// chunk, what lies behind is the concatenation of the following chunks, and position q0 + k needs exactly U(q0 + k + 2): four steps
// backwards over the thread's own entries, nothing from its neighbours.  A key's SECOND symbol is coded without context as well (an LDS
// read instead of another scattered table read; the order-1 code would save that one symbol 0.3 bits).
constexpr int PO2_NCH = CT / 4 + 15;        // the tile's 1024 chunks and 15 behind it (symbols up to q = 18 + 4 * 1038 + 3 < CT + PK_HALO)
__global__ __launch_bounds__(TB) void k_pack_keys_o2(const uint8_t *__restrict__ T, uint32_t n, const SaState *__restrict__ st, uint64_t *__restrict__ P,
                                                    const uint8_t *__restrict__ blk, const uint32_t *__restrict__ bend, uint8_t *__restrict__ D0,
                                                    const uint32_t *__restrict__ ctab, const uint16_t *__restrict__ ctxmap)
{
    if (st->vmode != 3u) return;
    __shared__ __align__(16) uint8_t cr[CT + PK_HALO];                // bytes: index q = position i_lo - 16 + q
    __shared__ __align__(16) uint8_t cb[CT + PK_HALO];                // block numbers (group sort)
    __shared__ uint32_t lcl[256];
    __shared__ uint64_t CB[PO2_NCH], CM[PO2_NCH];
    __shared__ uint8_t CL[PO2_NCH];
    static_assert(18 + 4 * (PO2_NCH - 1) + 3 < CT + PK_HALO, "the last chunk's symbols are staged");
    lcl[threadIdx.x] = st->vcode[threadIdx.x] | ((uint32_t)st->vlen[threadIdx.x] << 27);
    const uint32_t tag_max = st->tag_max;
    const uint32_t ntiles = (n + CT - 1) / CT;
    const int t = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT, cnt = (n - base < (uint32_t)CT) ? n - base : (uint32_t)CT;
        const int64_t i_lo = (int64_t)n - 1 - base - (CT - 1);
        __syncthreads();
        for (int q = threadIdx.x; q < CT + PK_HALO; q += TB) {
            const int64_t p = i_lo - 16 + q;
            const bool in = p >= 0 && p < (int64_t)n;
            cr[q] = in ? T[p] : (uint8_t)0;
            if (blk) cb[q] = in ? blk[p] : (uint8_t)0;
        }
        __syncthreads();
        auto follows = [&](int q) {                                    // the symbol at q exists and belongs to the suffix that holds q - 1
            const int64_t p = i_lo - 16 + q;
            return p >= 1 && p < (int64_t)n && !(blk && cb[q] != cb[q - 1]);
        };
        // chunk j: the symbols q = 18 + 4 j .. + 3, each behind its two bytes; 0 = it cannot follow.  All table reads of a thread's chunks
        // are issued before anything is built from them.
        constexpr int NIT = CT / (TB * 4);
        uint32_t g[NIT + 1][4];
        auto lookup = [&](int j, uint32_t (&a)[4]) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = 18 + 4 * j + k;
                const bool ok = follows(q) && i_lo - 16 + q >= 2;
                const uint32_t row = ok ? ctxmap[((uint32_t)cr[q - 2] << 8) | cr[q - 1]] : 0u;
                a[k] = ok ? ctab[row * 256u + cr[q]] : 0u;
            }
        };
        auto chunk = [&](int j, const uint32_t (&a)[4]) {
            uint64_t bits = 0, marks = 0;
            uint32_t used = 0;
            bool stop = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (a[k] == 0u) stop = true;
                if (!stop && used < 64u) {
                    const uint32_t l = a[k] >> 27, c = a[k] & 0x7FFFFFFu;
                    if (used + l <= 64u) { used += l; bits |= (uint64_t)c << (64u - used); marks |= 1ull << (64u - used); }
                    else { bits |= (uint64_t)c >> (used + l - 64u); used = 64u; }
                }
            }
            CB[j] = bits;
            CM[j] = marks;
            CL[j] = (uint8_t)(used | (stop ? 0x80u : 0u));
        };
        {
            const bool halo = t < PO2_NCH - CT / 4;
            if (halo) lookup(CT / 4 + t, g[NIT]);
#pragma unroll
            for (int it = 0; it < NIT; it++) lookup(CT / 4 - 1 - (it * TB + t), g[it]);
            if (halo) chunk(CT / 4 + t, g[NIT]);
#pragma unroll
            for (int it = 0; it < NIT; it++) chunk(CT / 4 - 1 - (it * TB + t), g[it]);
        }
        __syncthreads();
        static_assert(NIT == 4, "the selection of a step's own entries below");
#pragma unroll 1
        for (int it = 0; it < NIT; it++) {
            const int j = CT / 4 - 1 - (it * TB + t);
            const uint32_t x0 = (uint32_t)((it * TB + t) * 4);         // my four slots x0 .. x0 + 3 = positions q0 + 3, q0 + 2, ..
            if (x0 >= cnt) continue;
            uint32_t own[4];                                           // this step's entries (the loop stays rolled: registers)
#pragma unroll
            for (int k = 0; k < 4; k++) own[k] = it == 0 ? g[0][k] : it == 1 ? g[1][k] : it == 2 ? g[2][k] : g[3][k];
            const int q0 = 16 + 4 * j;
            uint64_t U = 0, M = 0, key[4];
            uint32_t used = 0, dep = 0;
            for (int jj = j + 1; jj < PO2_NCH; jj++) {                 // U(q0 + 6): what lies behind my chunk
                const uint32_t L = CL[jj];
                U |= CB[jj] >> used;
                M |= CM[jj] >> used;
                used += L & 0x7Fu;
                if ((L & 0x80u) || used >= 64u) break;
            }
#pragma unroll
            for (int jx = 0; jx < 4; jx++) {
                const int q = q0 + 3 - jx;
                const int64_t i = i_lo - 16 + q;
                {                                                      // the symbol q + 2 in front: U(q + 2)
                    const uint32_t en = own[3 - jx];
                    if (en == 0u) { U = 0; M = 0; }                    // it cannot follow: nothing behind q + 1
                    else {
                        const uint32_t l = en >> 27;
                        U = ((uint64_t)(en & 0x7FFFFFFu) << (64u - l)) | (U >> l);
                        M = (1ull << (64u - l)) | (M >> l);
                    }
                }
                const uint32_t e0 = lcl[cr[q]], l0 = e0 >> 27;
                uint64_t k56 = (uint64_t)(e0 & 0x7FFFFFFu) << (56u - l0);
                uint32_t d = 1u;
                if (follows(q + 1)) {
                    const uint32_t en1 = lcl[cr[q + 1]], l1 = en1 >> 27;
                    k56 |= ((uint64_t)(en1 & 0x7FFFFFFu) << (56u - l0 - l1)) | (U >> (8u + l0 + l1));
                    d = 2u + (uint32_t)__popcll(M >> (8u + l0 + l1));
                }
                const uint32_t low = bend ? cb[q] : (i > 0 ? cr[q - 1] : 0u);
                key[jx] = (k56 << 8) | low;
                dep |= (d < tag_max ? d : tag_max) << (8 * jx);
            }
            if (x0 + 4 <= cnt) {
                uint4 *o = reinterpret_cast<uint4 *>(P + base + x0);
                o[0] = make_uint4((uint32_t)key[0], (uint32_t)(key[0] >> 32), (uint32_t)key[1], (uint32_t)(key[1] >> 32));
                o[1] = make_uint4((uint32_t)key[2], (uint32_t)(key[2] >> 32), (uint32_t)key[3], (uint32_t)(key[3] >> 32));
                *reinterpret_cast<uint32_t *>(D0 + base + x0) = dep;
            } else {
#pragma unroll
                for (int jx = 0; jx < 4; jx++)
                    if (x0 + jx < cnt) { P[base + x0 + jx] = key[jx]; D0[base + x0 + jx] = (uint8_t)(dep >> (8 * jx)); }
            }
        }
        // the slots between the text's end and the end of the last tile: the largest key, so that the radix sort (stable, fed whole
        // tiles since round 6) leaves them behind every suffix; nothing reads them afterwards
        for (uint32_t x = cnt + threadIdx.x; x < (uint32_t)CT; x += TB) { P[base + x] = ~0ull; D0[base + x] = 0; }
    }
}

// The tile is also a tile of the radix sort's first pass (same 4096 slots): for the two-pass form of the sort its digit histogram (key
// bits 15..8) is counted here, from LDS, so that pass has no histogram kernel of its own (tilehist: digit-major [256][ntiles], radix.hip;
// null for the one-pass form).
__global__ __launch_bounds__(TB) void k_pack_keys(const uint8_t *__restrict__ T, uint32_t n, const SaState *__restrict__ st, uint64_t *__restrict__ P,
                                                 const uint8_t *__restrict__ blk, const uint32_t *__restrict__ bend, uint32_t *__restrict__ tilehist,
                                                 uint8_t *__restrict__ D0)
{
    __shared__ uint32_t hd[WAVES][256];
    __shared__ __align__(16) uint8_t cc[CT + PK_HALO];                // codes
    __shared__ __align__(16) uint8_t cr[CT + PK_HALO];                // bytes
    __shared__ __align__(16) uint8_t cb[CT + PK_HALO];                // block numbers (group sort)
    __shared__ uint64_t ko[PK_KO];
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = st->lut[threadIdx.x];
    const uint32_t bits = st->bits;
    if (D0 && st->vmode) return;                                       // (uniform) variable-length keys: k_pack_keys_var does the tiles
    const uint32_t ntiles = (n + CT - 1) / CT;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT, cnt = (n - base < (uint32_t)CT) ? n - base : (uint32_t)CT;
        const int64_t i_lo = (int64_t)n - 1 - base - (CT - 1);        // position of the tile's LAST slot (negative in the last tile: no such slot)
        __syncthreads();                                                // lut; the previous tile's ko has been read
        if (D0) {                                                       // fixed-width keys carry no depth: tag 0 in every slot (the radix sort's first pass reads the bytes)
            if (cnt == (uint32_t)CT) reinterpret_cast<uint4 *>(D0 + base)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
            else for (uint32_t x = threadIdx.x; x < cnt; x += TB) D0[base + x] = 0;
        }
        for (int q = threadIdx.x; q < CT + PK_HALO; q += TB) {
            const int64_t p = i_lo - 16 + q;
            const bool in = p >= 0 && p < (int64_t)n;
            const uint32_t raw = in ? T[p] : 0u;
            cr[q] = (uint8_t)raw;
            cc[q] = in ? lut[raw] : (uint8_t)0;
            if (blk) cb[q] = in ? blk[p] : (uint8_t)0;
        }
        __syncthreads();
        switch (bits) {
        case 1: pack_tile<1>(cc, cr, cb, ko, i_lo, n, bend); break;
        case 2: pack_tile<2>(cc, cr, cb, ko, i_lo, n, bend); break;
        case 3: pack_tile<3>(cc, cr, cb, ko, i_lo, n, bend); break;
        case 4: pack_tile<4>(cc, cr, cb, ko, i_lo, n, bend); break;
        case 5: pack_tile<5>(cc, cr, cb, ko, i_lo, n, bend); break;
        case 6: pack_tile<6>(cc, cr, cb, ko, i_lo, n, bend); break;
        case 7: pack_tile<7>(cc, cr, cb, ko, i_lo, n, bend); break;
        default: pack_tile<8>(cc, cr, cb, ko, i_lo, n, bend); break;
        }
        for (int i = threadIdx.x; i < WAVES * 256; i += TB) (&hd[0][0])[i] = 0u;
        __syncthreads();
        {
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            const uint64_t lt = mask_below(l);
#pragma unroll
            for (int it = 0; it < CT_ITEMS; it++) {
                const uint32_t x = (uint32_t)(w * (64 * CT_ITEMS) + it * 64 + l);
                const bool valid = x < cnt;
                const uint64_t key = ko[x + (x >> 4)];
                if (valid) P[base + x] = key;
                if (tilehist) {                                    // (uniform; the one-pass radix form counts for itself)
                    const uint32_t dig = (uint32_t)(key >> 8) & 255u;
                    const uint64_t mm = match_any8(dig, valid);
                    if (valid && (mm & lt) == 0ull) hd[w][dig] += (uint32_t)__popcll(mm);      // one lane per digit value and wave: plain read-modify-write
                }
            }
        }
        for (uint32_t x = cnt + threadIdx.x; x < (uint32_t)CT; x += TB) { P[base + x] = ~0ull; if (D0) D0[base + x] = 0; }   // pad slots of the last tile: see k_pack_keys_var
        __syncthreads();
        if (tilehist) {
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < WAVES; k++) sum += hd[k][threadIdx.x];
            tilehist[(size_t)threadIdx.x * ntiles + tile] = sum;
        }
    }
}

// ---- round 0 ---------------------------------------------------------------------------------------------------------
// Round 0 sorts slot j = suffix n-1-j by key = first D bytes (as codes, see above), big-endian in bits 63..8, zero padded past the
// end of the text (k_pack_keys built the keys).  The LSD sort is stable, so suffixes that tie on the padded
// bytes come out in DESCENDING text position, i.e. a short suffix (a proper prefix of everything it ties with) lands in
// front -- plain suffix order even when the text contains 0x00 -- and the low byte of the key needs no sort pass.

// head of an equal-key run; a suffix with fewer than D bytes is always a group of its own
// Group sort (bend != null: several blocks sorted as one text, the key's low byte = block number, see radix.hip): the whole key
// takes part in the comparison, and a suffix is "short" when fewer than D bytes are left in ITS block.
__device__ __forceinline__ uint32_t r0_end(uint64_t key, uint32_t n, const uint32_t *__restrict__ bend) { return bend ? bend[(uint32_t)key & 255u] : n; }
// vmode (variable-length keys): the sorted value carries the key's depth in its upper bits (from SaState::tag_shift up) and a
// suffix is "short" (a group of its own) when its depth reaches the end of the text: every symbol it has is in the key
// (in vmode the parameter D of the helpers below is the tag shift, not a depth)
// Blocks above 2^28 bytes (round 6; format.hpp:22 allows 1000 MiB): a 29- or 30-bit suffix number leaves no room for a depth, so the tag
// shift is 32 -- nothing rides in the value -- and the depth of suffix s is read from the slots' own array, Dx[n - 1 - s] (slot j holds
// suffix n - 1 - j; sa_layout gives the array a buffer of its own there).  Only a suffix within 63 symbols of its end can be short, so
// the heads cost no extra read; what does is the depth of every unresolved group (one random byte per group head, k_r0_finish).
__device__ __forceinline__ bool r0_short(uint32_t v, uint64_t key, uint32_t n, const uint32_t *__restrict__ bend, uint32_t D, bool vmode,
                                         const uint8_t *__restrict__ Dx = nullptr)
{
    if (vmode && Dx) {
        const uint32_t e = r0_end(key, n, bend);
        return (uint64_t)v + 63u >= e && v + Dx[n - 1u - v] >= e;
    }
    return vmode ? (v & ((1u << D) - 1u)) + (v >> D) >= r0_end(key, n, bend) : v + D > r0_end(key, n, bend);
}
__device__ __forceinline__ bool r0_head(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sa, uint32_t j, uint32_t n, const uint32_t *__restrict__ bend,
                                        uint32_t D, bool vmode, const uint8_t *__restrict__ Dx)
{
    if (j == 0) return true;
    const uint64_t a = keys[j], b = keys[j - 1];
    return ((a ^ b) >> (bend ? 0 : 8)) != 0ull || r0_short(sa[j], a, n, bend, D, vmode, Dx) || r0_short(sa[j - 1], b, n, bend, D, vmode, Dx);   // bits 7..0 carry T[sa-1], not key
}

// head words of one 4096-slot tile: HE[word] = heads | slots past the end (so that "the next slot is a head" is one shift),
// HE[64] bit 0 = head flag of the first slot of the next tile.  Every thread loads its sixteen (key, suffix) pairs ONCE, all loads
// in flight together (clamped indices, no branch around a load), and hands them back to the caller; the key in front of a slot
// comes from the neighbouring lane (DPP wave shift; lane 0: lane 63 of the row before, the wave's first row: one extra load),
// and "the suffix in front is shorter than D bytes" is the shifted ballot of the row's own "short" bits.
// (vmode: sj[] comes back WITH the depth tag in its upper bits)
__device__ __forceinline__ void r0_tile_heads(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sa, uint32_t n, uint32_t base, uint64_t *HE,
                                              uint64_t (&kj)[CT_ITEMS], uint32_t (&sj)[CT_ITEMS], const uint32_t *__restrict__ bend, uint32_t D, bool vmode,
                                              const uint8_t *__restrict__ Dx)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint32_t j0 = base + w * (64 * CT_ITEMS);
#pragma unroll
    for (int k = 0; k < CT_ITEMS; k++) {
        const uint32_t j = j0 + k * 64 + l, jc = j < n ? j : n - 1;
        kj[k] = keys[jc];
        sj[k] = sa[jc];
    }
    // the pair in front of the wave's first slot (uniform)
    const uint32_t jb = (j0 && j0 <= n) ? j0 - 1 : 0;
    const uint64_t kb = keys[jb];
    const uint32_t sb = sa[jb];
    uint32_t plo = (uint32_t)kb, phi = (uint32_t)(kb >> 32);
    uint64_t carry_short = (j0 && r0_short(sb, kb, n, bend, D, vmode, Dx)) ? 1ull : 0ull;
    const int low_shift = bend ? 0 : 8;                                              // bits 7..0 carry T[sa-1], not key -- or the block number, which is key
#pragma unroll
    for (int k = 0; k < CT_ITEMS; k++) {
        const uint32_t j = j0 + k * 64 + l;
        const uint32_t lo = (uint32_t)kj[k], hi = (uint32_t)(kj[k] >> 32);
        const uint32_t qlo = (uint32_t)__builtin_amdgcn_update_dpp((int)plo, (int)lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const uint32_t qhi = (uint32_t)__builtin_amdgcn_update_dpp((int)phi, (int)hi, 0x138, 0xf, 0xf, false);
        const bool differs = (((lo ^ qlo) >> low_shift) | (hi ^ qhi)) != 0u;
        const uint64_t S = __ballot(r0_short(sj[k], kj[k], n, bend, D, vmode, Dx));      // a suffix with fewer than `depth` bytes is a group of its own
        const uint64_t b = __ballot(differs || j >= n || j == 0) | S | (S << 1) | carry_short;
        if (l == 0) HE[w * CT_ITEMS + k] = b;
        carry_short = S >> 63;
        plo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
        phi = (uint32_t)__builtin_amdgcn_readlane((int)hi, 63);
    }
    if (threadIdx.x == 0) {
        const uint32_t jn = base + CT;
        HE[64] = (jn >= n || r0_head(keys, sa, jn, n, bend, D, vmode, Dx)) ? 1ull : 0ull;
    }
}
__device__ __forceinline__ uint64_t valid_word(uint32_t word_base, uint32_t n)
{
    if (word_base >= n) return 0ull;
    const uint32_t left = n - word_base;
    return left >= 64u ? ~0ull : ((1ull << left) - 1ull);
}

// per tile: 1 + position of its last head (0: none), number of suffixes that stay unresolved
__global__ __launch_bounds__(TB) void k_r0_count(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sa, uint32_t n,
                                                uint32_t *__restrict__ tLast, uint32_t *__restrict__ tSurv, const uint32_t *__restrict__ bend,
                                                const SaState *__restrict__ st, const uint8_t *__restrict__ Dx)
{
    __shared__ uint64_t HE[65];
    const uint32_t ntiles = (n + CT - 1) / CT;
    const bool vmode = st->vmode != 0u;
    const uint32_t D = vmode ? st->tag_shift : st->depth;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT;
        __syncthreads();
        uint64_t kj[CT_ITEMS];
        uint32_t sj[CT_ITEMS];
        r0_tile_heads(keys, sa, n, base, HE, kj, sj, bend, D, vmode, Dx);
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const uint64_t he = HE[l], vm = valid_word(base + l * 64, n);
            const uint64_t hv = he & vm;
            const uint64_t nexth = (he >> 1) | (HE[l + 1] << 63);
            const uint64_t single = hv & nexth;
            uint32_t cnt = (uint32_t)__popcll(vm & ~single);
            uint32_t last = hv ? base + l * 64 + top_bit(hv) + 1u : 0u;
            cnt = wave_sum(cnt);
            last = wave_incl_max(last);
            if (l == 63) { tSurv[tile] = cnt; tLast[tile] = last; }
        }
    }
}

// one workgroup: carry-in head per tile (exclusive prefix max), output offset per tile (exclusive prefix sum), total -> state
__global__ __launch_bounds__(WG1) void k_r0_scan(uint32_t *__restrict__ tLast, uint32_t *__restrict__ tSurv, uint32_t n, SaState *__restrict__ st)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    const uint32_t ntiles = (n + CT - 1) / CT;
    wg_scan<OpMax, true, false>(tLast, tLast, ntiles, 0u, sm);
    const uint32_t total = wg_scan<OpSum, true, false>(tSurv, tSurv, ntiles, 0u, sm);
    if (threadIdx.x == 0) {
        st->m[1] = total;
        st->round_m[0] = n;
        st->round_m[1] = total;
        st->npieces = 0;
        st->lc = 0;
    }
}

// group rank (= index of the run head) -> ISA; singletons are finished: BWT byte at their SA position (and SA itself for the
// suffix-array probe); the rest is compacted into the active list
__global__ __launch_bounds__(TB) void k_r0_finish(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sa, uint32_t n,
                                                 const uint32_t *__restrict__ tCarry, const uint32_t *__restrict__ tOff,
                                                 uint32_t *__restrict__ ISA, uint8_t *__restrict__ bwt, uint32_t *__restrict__ SA,
                                                 uint32_t *__restrict__ a_sa, uint32_t *__restrict__ a_grp, uint8_t *__restrict__ a_prev, SaState *__restrict__ st,
                                                 const uint32_t *__restrict__ bend, uint32_t *__restrict__ GD, uint64_t *lb_status, uint32_t *lb_ticket,
                                                 const uint8_t *__restrict__ Dx)
{
    // lb_status != null (round 5): ONE pass -- the tile learns the survivors in front of it and the last head in front of it by decoupled
    // look-back over the tiles before it (ticket order; one 64-bit word per tile = flag | survivors << 31 | 1 + last head, agent-scope
    // atomics; a wave looks at 64 predecessors at a time) instead of from k_r0_count + k_r0_scan, which read the sorted pairs once more.
    __shared__ uint32_t s_tile, s_carry;
    __shared__ uint64_t HE[65];
    __shared__ uint64_t runkey[256];       // vmode: the key of a run of byte b, and the byte whose code starts a key's first 8 bits
    __shared__ uint64_t runsorted[256];    // ... and the run keys of the occurring bytes in byte order = ascending (codes above 8 bits: binary search)
    __shared__ uint16_t vtop[256];
    __shared__ uint32_t s_sigma;
    __shared__ uint64_t SV[64];            // survivor bits per word
    __shared__ uint32_t LHW[64];           // 1 + last head position at or before the end of word l (carry included; the two-pass form)
    __shared__ uint32_t LHL[64];           // ... inside the tile only (0: none yet)
    __shared__ uint32_t SW[64];            // output position of the first survivor of word l
    const uint32_t ntiles = (n + CT - 1) / CT;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool vmode = st->vmode != 0u;
    const uint32_t D = vmode ? st->tag_shift : st->depth, code_shift = 56u - st->bits;
    const uint64_t rep = st->rep;
    const uint32_t TAGM = (vmode && !Dx) ? (1u << D) - 1u : 0xFFFFFFFFu;       // (Dx: blocks above 2^28 bytes, no tag in the value -- r0_short)
    if (vmode) {
        runkey[threadIdx.x] = st->vrunkey[threadIdx.x];
        vtop[threadIdx.x] = st->vtop[threadIdx.x];
        runsorted[threadIdx.x] = ~0ull;
        __syncthreads();
        if (st->present[threadIdx.x]) runsorted[st->lut[threadIdx.x]] = runkey[threadIdx.x];     // lut = the byte's index among the occurring ones
        if (threadIdx.x == 255) s_sigma = (uint32_t)st->lut[255] + (st->present[255] ? 1u : 0u);
    }
    constexpr uint64_t LB_AGG = 1ull << 62, LB_PFX = 2ull << 62, LB_LO = (1ull << 31) - 1ull;
    for (uint32_t it = blockIdx.x; it < ntiles; it += gridDim.x) {
        __syncthreads();                                                // (s_tile / s_carry / s_off of the previous tile have been read)
        if (lb_status && threadIdx.x == 0) s_tile = atomicAdd(lb_ticket, 1u);
        __syncthreads();
        const uint32_t tile = lb_status ? s_tile : it;                  // look-back: tiles in the order their workgroups START
        const uint32_t base = tile * CT;
        uint64_t kj[CT_ITEMS];
        uint32_t sj[CT_ITEMS];
        r0_tile_heads(keys, sa, n, base, HE, kj, sj, bend, D, vmode, Dx);
        __syncthreads();
        uint32_t carry = lb_status ? 0u : tCarry[tile];
        uint32_t nrun = 0;
        // one slot of the tile.  PHASE 0: everything at once (the two-pass comparator: carry and offsets are known up front).  Look-back form,
        // round 6: PHASE 1 = what needs nothing from the tiles in front -- the rank store (the block's one 64 Mi-element random write), the
        // BWT byte and the depth of every slot whose group's head lies INSIDE the tile -- issued while wave 0 is still looking back; PHASE 2 =
        // the rest: the survivors' list entries (their positions start at the survivors in front of the tile) and the slots in front of the
        // tile's first head (their group's head is the last head in front of the tile).
        auto slot = [&](int k, int phase) {
            const int word = w * CT_ITEMS + k;
            const uint32_t j = base + word * 64 + l;
            if (j >= n) return;
            const uint64_t hv = HE[word] & valid_word(base + word * 64, n);
            const uint64_t le = hv & mask_upto(l);
            // 1 + the head of my group if it lies inside the tile (0: in front of it)
            const uint32_t local = le ? base + word * 64 + top_bit(le) + 1u : (word ? (phase == 0 ? LHW[word - 1] : LHL[word - 1]) : 0u);
            const bool inside = phase == 0 || local != 0u;
            if (phase == 1 && !inside) return;
            const uint32_t grp = (local ? local : carry) - 1u;
            const uint32_t s = sj[k] & TAGM;                        // (loaded once, by r0_tile_heads)
            const uint8_t pv = (uint8_t)kj[k];                      // T[s - 1], carried in the key's low byte since pass 0 (group sort: the block number)
            const uint64_t sv = SV[word];
            const bool survivor = (sv >> l) & 1ull;
            if (phase != 2 || !inside) {                            // (phase 2 repeats nothing phase 1 has stored)
                ISA[s] = grp;
                if (!survivor) {
                    bwt[j] = pv;
                    if (SA) SA[j] = s;
                } else if (vmode && ((HE[word] >> l) & 1ull)) GD[grp] = Dx ? (uint32_t)Dx[n - 1u - s] : sj[k] >> D;      // the group's depth, written by its first member
            }
            if (phase == 1 || !survivor) return;
            const uint32_t pos = SW[word] + (uint32_t)__popcll(sv & mask_below(l));
            // `depth` equal bytes (a survivor has all of them: short suffixes are groups of their own): a run member
            const uint64_t k7 = kj[k] >> 8;
            bool inrun;
            if (vmode) {
                // the key of a run is a function of its byte; the byte is the one whose code starts the key: a table on the
                // key's first 8 bits for codes up to 8 bits, a binary search over the (ascending) run keys for the rare longer ones
                const uint32_t c = vtop[(uint32_t)(k7 >> 48)];
                if (c != 0xFFFFu) inrun = k7 == runkey[c];
                else {
                    uint32_t lo = 0, hi = s_sigma;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (runsorted[mid] < k7) lo = mid + 1u; else hi = mid; }
                    inrun = lo < s_sigma && runsorted[lo] == k7;
                }
            } else inrun = k7 == (k7 >> code_shift) * rep;
            nrun += inrun ? 1u : 0u;
            a_sa[pos] = s;
            a_grp[pos] = grp | (inrun ? RUNF : 0u);
            a_prev[pos] = pv;
        };
        uint32_t cnt = 0, inc = 0, last = 0;                            // (wave 0: survivors of my word, their running sum, 1 + last head so far in the tile)
        if (threadIdx.x < 64) {
            const uint64_t he = HE[l], vm = valid_word(base + l * 64, n);
            const uint64_t hv = he & vm;
            const uint64_t nexth = (he >> 1) | (HE[l + 1] << 63);
            const uint64_t surv = vm & ~(hv & nexth);
            SV[l] = surv;
            cnt = (uint32_t)__popcll(surv);
            inc = wave_incl_sum(cnt);
            last = hv ? base + l * 64 + top_bit(hv) + 1u : 0u;
            last = wave_incl_max(last);
            LHL[l] = last;
        }
        if (!lb_status) {
            if (threadIdx.x < 64) {
                SW[l] = tOff[tile] + inc - cnt;
                LHW[l] = last > carry ? last : carry;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < CT_ITEMS; k++) slot(k, 0);
        } else {
            uint32_t cnt_tile = 0, last_tile = 0;
            if (threadIdx.x < 64) {                                     // the aggregate leaves before anything else
                cnt_tile = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                last_tile = (uint32_t)__builtin_amdgcn_readlane((int)last, 63);
                const uint64_t mine = ((uint64_t)cnt_tile << 31) | last_tile;
                if (l == 0) __hip_atomic_store(lb_status + tile, (tile == 0 ? LB_PFX : LB_AGG) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();                                            // SV, LHL
            if (threadIdx.x < 64) {
                uint32_t o_acc = 0, c_acc = 0;
                if (tile != 0) {
                    int64_t pos = (int64_t)tile - 1;                    // lane l looks at tile pos - l
                    for (;;) {
                        const int64_t t = pos - l;
                        uint64_t wv;
                        uint64_t need;                                  // lanes up to the first prefix
                        bool found;
                        for (;;) {
                            wv = t >= 0 ? __hip_atomic_load(lb_status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PFX;
                            const uint64_t pf = __ballot((wv >> 62) == 2ull), np = __ballot((wv >> 62) == 0ull);
                            found = pf != 0ull;
                            need = found ? mask_upto((int)__builtin_ctzll(pf)) : ~0ull;
                            if ((np & need) == 0ull) break;            // everybody between me and the first prefix has published
                            __builtin_amdgcn_s_sleep(1);
                        }
                        const bool in = (need >> l) & 1ull;
                        uint32_t so = in ? (uint32_t)((wv >> 31) & LB_LO) : 0u, sc = in ? (uint32_t)(wv & LB_LO) : 0u;
                        so = wave_sum(so);
                        sc = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max(sc), 63);
                        o_acc += so;
                        c_acc = c_acc > sc ? c_acc : sc;
                        if (found) break;                              // a prefix was among them (tiles before tile 0 count as one; lane 63's too)
                        pos -= 64;
                    }
                    const uint32_t lt = c_acc > last_tile ? c_acc : last_tile;
                    if (l == 0) __hip_atomic_store(lb_status + tile, LB_PFX | ((uint64_t)(o_acc + cnt_tile) << 31) | lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                SW[l] = o_acc + inc - cnt;
                if (l == 0) {
                    s_carry = c_acc;
                    if (tile + 1 == ntiles) {                           // the last tile knows the total: what k_r0_scan leaves in the state
                        st->m[1] = o_acc + cnt_tile;
                        st->round_m[0] = n;
                        st->round_m[1] = o_acc + cnt_tile;
                        st->npieces = 0;
                        st->lc = 0;
                    }
                }
            }
            // phase 1: waves 1..3 at once, wave 0 behind its look-back (its prefix is out before its own stores)
#pragma unroll
            for (int k = 0; k < CT_ITEMS; k++) slot(k, 1);
            __syncthreads();                                            // SW, s_carry
            carry = s_carry;
#pragma unroll
            for (int k = 0; k < CT_ITEMS; k++) slot(k, 2);
        }
        if (__ballot(nrun != 0)) {                                   // (rare: text has few runs that long)
            nrun = wave_sum(nrun);
            if (l == 0) atomicAdd(&st->nrun, nrun);
        }
    }
}

// ---- run lengths (only when round 0 left run members behind: every kernel returns at once otherwise) -------------------
// RL[i] = number of bytes equal to T[i] from i on (the remaining length of the run i lies in) = (next position whose byte differs
// from its successor) + 1 - i.  Per 4096-byte tile: first boundary position; suffix-min over the tiles; fill.
__device__ __forceinline__ bool run_ends_at(const uint8_t *__restrict__ T, const uint8_t *__restrict__ blk, uint32_t i, uint32_t n)
{
    return i + 1 == n || T[i] != T[i + 1] || (blk && blk[i] != blk[i + 1]);        // (group sort: a run stops at the end of its block)
}
__global__ __launch_bounds__(TB) void k_run_first(const uint8_t *__restrict__ T, uint32_t n, const SaState *__restrict__ st, uint32_t *__restrict__ tFirst,
                                                 const uint8_t *__restrict__ blk)
{
    if (st->nrun == 0) return;
    __shared__ uint32_t sm[TB / 64 + 1];
    const uint32_t ntiles = (n + CT - 1) / CT;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT;
        uint32_t first = NONE;
#pragma unroll
        for (int k = CT_ITEMS - 1; k >= 0; k--) {
            const uint32_t i = base + k * TB + threadIdx.x;
            if (i < n && run_ends_at(T, blk, i, n)) first = i;
        }
        uint32_t tot;
        block_incl_scan<OpMin>(first, sm, &tot);
        if (threadIdx.x == 0) tFirst[tile] = tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(WG1) void k_run_scan(uint32_t *__restrict__ tFirst, uint32_t n, const SaState *__restrict__ st)
{
    if (st->nrun == 0) return;
    __shared__ uint32_t sm[WG1 / 64 + 1];
    // tFirst[tile] <- first boundary in any LATER tile (exclusive suffix min); position n - 1 is always a boundary
    wg_scan<OpMin, true, true>(tFirst, tFirst, (n + CT - 1) / CT, NONE, sm);
}
__global__ __launch_bounds__(TB) void k_run_fill(const uint8_t *__restrict__ T, uint32_t n, const SaState *__restrict__ st, const uint32_t *__restrict__ tAfter,
                                                uint32_t *__restrict__ RL, const uint8_t *__restrict__ blk)
{
    if (st->nrun == 0) return;
    __shared__ uint32_t sm[TB / 64 + 1];
    const uint32_t ntiles = (n + CT - 1) / CT;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT, p0 = base + threadIdx.x * CT_ITEMS;        // blocked: sixteen consecutive positions per thread
        uint32_t bits = 0, first = NONE;
#pragma unroll
        for (int k = CT_ITEMS - 1; k >= 0; k--) {
            const uint32_t i = p0 + k;
            if (i < n && run_ends_at(T, blk, i, n)) { bits |= 1u << k; first = i; }
        }
        // first boundary in the segments of the threads AFTER me: inclusive min-scan over the threads in reverse order
        __shared__ uint32_t rv[TB];
        __syncthreads();                                                              // rv of the previous tile has been read
        rv[TB - 1 - threadIdx.x] = first;
        __syncthreads();
        const uint32_t rinc = block_incl_scan<OpMin>(rv[threadIdx.x], sm, nullptr);   // index u: min over the threads >= TB - 1 - u
        __syncthreads();
        rv[threadIdx.x] = rinc;
        __syncthreads();
        uint32_t nb = (threadIdx.x == TB - 1) ? NONE : rv[TB - 2 - threadIdx.x];
        if (nb == NONE) nb = tAfter[tile];
#pragma unroll
        for (int k = CT_ITEMS - 1; k >= 0; k--) {
            const uint32_t i = p0 + k;
            if (bits & (1u << k)) nb = i;
            if (i < n) RL[i] = nb + 1u - i;
        }
    }
}

// ---- doubling rounds -------------------------------------------------------------------------------------------------
// key2 of every active suffix + head words per window -> FH / LH = 1 + first / last head position of the window (0: none)
// Run members (RUNF, see the constant).  Among suffixes that start with the same byte c repeated, the order is
//   [run ends in a byte < c, or at the end of the text: ascending run length]  <  [run ends in a byte > c: descending run length]
// (A = c^a x.., B = c^b y.. with a < b differ at offset a: x against c), ties = same kind and length, decided by the suffix behind
// the run.  The invariant of prefix doubling -- at the start of the round with distance h every group shares its first h bytes,
// so every rank read at distance h resolves h more -- must hold for them too:
//   round 1 (h = D, round 0's depth; the group is everything that starts with c^D): the ordinary key2 = rankD(s + D) + 1 already places
//     the members with fewer than 2D equal bytes; those with 2D or more all read the rank G of their own group there.  They get
//     G + 1 + (L for the first kind, 2n - L for the second), keys above G + 1 move up by 2n: one key, every group 2D-ordered.
//   later rounds: a group of run members with L >= 2D has one kind and one run length, so it may compare at distance max(h, L)
//     -- the rank of the suffix behind the run, whatever the run's length: it shares L >= that many bytes (L > h) or is h-ordered
//     like everybody else, and gains >= h either way.  An all-zero block is sorted after round 1.
__global__ __launch_bounds__(TB) void k_gather_win(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, const SaState *__restrict__ st,
                                                  int par, uint32_t n, int hshift, const uint32_t *__restrict__ ISA, uint32_t *__restrict__ k2,
                                                  uint32_t *__restrict__ FH, uint32_t *__restrict__ LH,
                                                  const uint8_t *__restrict__ T, const uint32_t *__restrict__ RL, int first_round,
                                                  const uint8_t *__restrict__ a_blk, const uint32_t *__restrict__ bend, const uint32_t *__restrict__ GDr)
{
    __shared__ uint64_t H[16];
    const uint32_t m = st->m[par];
    // vmode (variable-length keys; GDr = the depth of every unresolved group by its rank): a group compares at ITS OWN depth -- the
    // symbols its members are known to share -- instead of the round's common distance.  The run rule needs no later-round case then:
    // round 1 gives the members it spreads by (kind, run length) the depth L (k_seg_round / k_lg_finish), so they read the suffix
    // behind their run like everybody reads the suffix behind its depth.
    const bool vmode = GDr != nullptr && st->vmode != 0u;
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    // the round's distance: round 0 resolved `depth` bytes (k_key_plan), every round doubles it
    const uint32_t D = st->depth;
    const uint64_t h64 = (uint64_t)D << hshift;
    const uint32_t h = (hshift < 32 && h64 < n) ? (uint32_t)h64 : n;
    for (uint32_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        const uint32_t base = win * SEG_TILE;
        __syncthreads();
        // every load of the window is issued without a branch around it (clamped indices, results masked afterwards), so that
        // the list reads and then the rank gathers are all in flight together
        uint32_t s[WIN_ITEMS], gj[WIN_ITEMS], gp[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l, jc = j < m ? j : m - 1;
            s[k] = a_sa[jc];
            gj[k] = a_grp[jc];
            gp[k] = a_grp[jc ? jc - 1 : 0];
        }
        uint32_t kv[WIN_ITEMS], lim[WIN_ITEMS];
        bool anyrun = false;
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) lim[k] = n;
        if (bend) {                          // group sort: a suffix ends with its block (the block number rides in the byte array)
#pragma unroll
            for (int k = 0; k < WIN_ITEMS; k++) {
                const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l;
                lim[k] = bend[a_blk[j < m ? j : m - 1]];
            }
        }
        uint32_t hk[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) hk[k] = vmode ? GDr[gj[k] & ~(RUNF | DONE)] : h;      // (vmode: ascending ranks along the list, neighbours share the word)
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint64_t s2 = (uint64_t)s[k] + hk[k];
            kv[k] = ISA[s2 < lim[k] ? s2 : 0];
            kv[k] = (s2 < lim[k]) ? kv[k] + 1u : 0u;
            anyrun |= (gj[k] & RUNF) != 0u;
        }
        if (__ballot(anyrun)) {                                      // wave-uniform and rare: text has few runs of `depth` equal bytes
#pragma unroll
            for (int k = 0; k < WIN_ITEMS; k++) {
                const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l;
                if (j < m && (gj[k] & RUNF)) {
                    const uint32_t G1 = (gj[k] & ~(RUNF | DONE)) + 1u;  // key2 of "the suffix D further is still in my group": 2D equal bytes
                    if (first_round) {
                        // every key keeps its place relative to the group's own rank; the members with >= 2D equal bytes, which
                        // all tie there, are spread over the 2n values behind it by (kind, run length)
                        if (kv[k] == G1) {
                            const uint32_t L = RL[s[k]];
                            const uint64_t e = (uint64_t)s[k] + L;  // first position behind the run
                            const bool down = e >= lim[k] || T[e] < T[s[k]];
                            kv[k] = G1 + (down ? L : 2u * n - L);    // L in [2D, n]: the kinds cannot collide (2n - L >= n >= L, equal only for L = n: one suffix)
                        } else if (kv[k] > G1) kv[k] += 2u * n;
                    } else if (!vmode) {
                        const uint32_t L = RL[s[k]];
                        if (L >= 2u * D && L > h) {                  // (a descendant of a c^D group with a shorter run is an ordinary suffix)
                            const uint64_t e = (uint64_t)s[k] + L;
                            kv[k] = e < lim[k] ? ISA[e] + 1u : 0u;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l;
            bool head = false;
            if (j < m) {
                head = (j == 0) || (gj[k] != gp[k]);
                k2[j] = kv[k];
            }
            const uint64_t b = __ballot(head);
            if (l == 0) H[w * WIN_ITEMS + k] = b;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint64_t hv = (l < 16) ? H[l] : 0ull;
            uint32_t first = hv ? base + l * 64 + (uint32_t)__builtin_ctzll(hv) + 1u : NONE;
            uint32_t last = hv ? base + l * 64 + top_bit(hv) + 1u : 0u;
            first = wave_incl_min(first);
            last = wave_incl_max(last);
            if (l == 63) { FH[win] = (first == NONE) ? 0u : first; LH[win] = last; }
        }
    }
}

// Geometry of the (at most two) pieces that groups larger than SEG_TILE cut out of window w: A = the tail of the group that
// spills in, B = the head of the window's last group.  PH = 1 + last head at or before the end of window w, NH = first head
// position after window w (m if none).
__device__ __forceinline__ void win_geometry(uint32_t w, uint32_t m, const uint32_t *FH, const uint32_t *LH, const uint32_t *PH, const uint32_t *NH,
                                             bool &hasA, uint32_t &a_end, uint32_t &a_gs, uint32_t &a_ge, bool &hasB, uint32_t &b_gs, uint32_t &b_ge)
{
    const uint32_t base = w * SEG_TILE;
    const uint32_t wend = (base + SEG_TILE < m) ? base + SEG_TILE : m;
    const uint32_t fh = FH[w], lh = LH[w];
    hasA = (fh != base + 1u);                         // the first slot of the window is not a head (then w > 0: slot 0 is one)
    a_gs = 0; a_ge = 0; a_end = 0;
    if (hasA) {
        a_gs = PH[w - 1] - 1u;
        a_end = fh ? fh - 1u : wend;
        a_ge = fh ? fh - 1u : NH[w];
        hasA = (a_ge - a_gs > (uint32_t)SEG_TILE);
    }
    hasB = (lh != 0u);
    b_gs = 0; b_ge = 0;
    if (hasB) {
        b_gs = lh - 1u;
        b_ge = NH[w];
        hasB = (b_ge - b_gs > (uint32_t)SEG_TILE);
    }
}

// window metadata in four small kernels: (1) one workgroup: PH, NH by scans; (2) all windows: piece counts; (3) one workgroup:
// exclusive prefix of the counts; (4) all windows: piece descriptors
__global__ __launch_bounds__(WG1) void k_win_scan1(const uint32_t *__restrict__ FH, const uint32_t *__restrict__ LH, uint32_t *PH, uint32_t *NH,
                                                  SaState *__restrict__ st, int par)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    if (threadIdx.x == 0) { st->npieces = 0; st->lc = 0; }
    if (nwin == 0) return;
    wg_scan<OpMax, false, false>(LH, PH, nwin, 0u, sm);
    // NH: exclusive suffix min of the first-head positions (FH holds position + 1, 0 = none)
    for (uint32_t w = threadIdx.x; w < nwin; w += WG1) NH[w] = FH[w] ? FH[w] - 1u : NONE;
    __syncthreads();
    wg_scan<OpMin, true, true>(NH, NH, nwin, m, sm);
}

__global__ __launch_bounds__(TB) void k_win_count(const uint32_t *__restrict__ FH, const uint32_t *__restrict__ LH, const uint32_t *__restrict__ PH,
                                                 const uint32_t *__restrict__ NH, uint32_t *__restrict__ PC, SaState *__restrict__ st, int par)
{
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    uint32_t lsum = 0;
    for (uint32_t w = blockIdx.x * TB + threadIdx.x; w < nwin; w += gridDim.x * TB) {
        bool hasA, hasB;
        uint32_t a_end, a_gs, a_ge, b_gs, b_ge;
        win_geometry(w, m, FH, LH, PH, NH, hasA, a_end, a_gs, a_ge, hasB, b_gs, b_ge);
        const uint32_t base = w * SEG_TILE;
        const uint32_t wend = (base + SEG_TILE < m) ? base + SEG_TILE : m;
        PC[w] = (hasA ? 1u : 0u) + (hasB ? 1u : 0u);
        lsum += (hasA ? a_end - base : 0u) + (hasB ? wend - b_gs : 0u);
    }
    lsum = wave_sum(lsum);
    if (lane_id() == 0 && lsum) atomicAdd(&st->lc, lsum);
}

__global__ __launch_bounds__(WG1) void k_win_scan2(uint32_t *PC, SaState *__restrict__ st, int par, int round)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    const uint32_t np = wg_scan<OpSum, true, false>(PC, PC, nwin, 0u, sm);
    if (threadIdx.x == 0) {
        st->npieces = np;
        if (round < JPK_SA_MAX_ROUNDS) st->round_lc[round] = st->lc;
    }
}

__global__ __launch_bounds__(TB) void k_win_pieces(const uint32_t *__restrict__ FH, const uint32_t *__restrict__ LH, const uint32_t *__restrict__ PH,
                                                  const uint32_t *__restrict__ NH, const uint32_t *__restrict__ PC, Piece *__restrict__ pieces,
                                                  const SaState *__restrict__ st, int par)
{
    if (st->npieces == 0) return;
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    for (uint32_t w = blockIdx.x * TB + threadIdx.x; w < nwin; w += gridDim.x * TB) {
        bool hasA, hasB;
        uint32_t a_end, a_gs, a_ge, b_gs, b_ge;
        win_geometry(w, m, FH, LH, PH, NH, hasA, a_end, a_gs, a_ge, hasB, b_gs, b_ge);
        const uint32_t base = w * SEG_TILE;
        const uint32_t wend = (base + SEG_TILE < m) ? base + SEG_TILE : m;
        uint32_t p = PC[w];
        if (hasA) {
            // the group's first piece is piece B of the window that holds its head
            const uint32_t w0 = a_gs / SEG_TILE;
            bool hA0, hB0;
            uint32_t t0, t1, t2, t3, t4;
            win_geometry(w0, m, FH, LH, PH, NH, hA0, t0, t1, t2, hB0, t3, t4);
            Piece q;
            q.begin = base; q.count = a_end - base; q.gs = a_gs; q.ge = a_ge;
            q.fp = PC[w0] + (hA0 ? 1u : 0u);
            q.nt = (a_ge - 1u) / SEG_TILE - w0 + 1u;
            q.tl = w - w0;
            q.pad = 0;
            pieces[p++] = q;
        }
        if (hasB) {
            Piece q;
            q.begin = b_gs; q.count = wend - b_gs; q.gs = b_gs; q.ge = b_ge;
            q.fp = p;
            q.nt = (b_ge - 1u) / SEG_TILE - w + 1u;
            q.tl = 0;
            q.pad = 0;
            pieces[p] = q;
        }
    }
}

// vmode: the symbols the members of a NEW group share = the old group's depth + the depth of the group their common key2 names (they all
// read the same rank at the old depth: Larsson-Sadakane's invariant per group instead of per round).  Exceptions, round 1 only, in a
// group of run members (RUNF): k_gather_win spread the members with >= 2 d equal bytes over G + 1 + (L | 2n - L) -- such a group shares
// its run, depth L -- and moved the ordinary keys above G + 1 up by 2n.  (A group of one needs no depth; key2 = 0 is always alone.)
__device__ __forceinline__ uint32_t new_group_depth(const uint32_t *__restrict__ GDr, uint32_t G, bool runf, uint32_t key2, bool first_round, uint32_t n)
{
    const uint32_t own = GDr[G];
    uint32_t tr = key2 - 1u;
    if (first_round && runf) {
        const uint32_t G1 = G + 1u;
        if (key2 > G1 && key2 - G1 <= 2u * n) {
            const uint32_t x = key2 - G1;
            return x <= n ? x : 2u * n - x;
        }
        if (key2 > G1) tr -= 2u * n;
    }
    const uint64_t d = (uint64_t)own + GDr[tr];
    return d > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)d;
}

// the sort / re-rank of all groups of <= SEG_TILE elements, one window per workgroup iteration
__global__ __launch_bounds__(TB) void k_seg_round(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, const uint32_t *__restrict__ k2g,
                                                 const SaState *__restrict__ st, int par, int key_bits, const uint32_t *__restrict__ PH,
                                                 const uint8_t *__restrict__ a_prev, uint32_t *__restrict__ ISA, uint8_t *__restrict__ bwt, uint32_t *__restrict__ SA,
                                                 uint32_t *__restrict__ b_sa, uint32_t *__restrict__ b_grp, uint8_t *__restrict__ b_prev,
                                                 const uint32_t *__restrict__ GDr, uint32_t *__restrict__ GDw, int first_round, uint32_t n)
{
    // LDS diet (30 KB, five workgroups per CU instead of three): the group ranks g[] are only needed while the group boundaries
    // are worked out and share their 8 KB with the two index permutations of the sort; the suffix numbers and the group rank of
    // the final positions are re-read from the (L2-resident) window instead of being kept; the scratch of the boundary scans
    // shares the digit counters' space.
    __shared__ uint32_t u_g_idx[SEG_SPAN];    // phase 1: g[] = group rank per loaded element; afterwards: idxA | idxB (uint16 each)
    __shared__ uint32_t k2[SEG_SPAN];         // sort key of the owned elements
    __shared__ uint16_t gsl[SEG_SPAN];        // group start (local position) per loaded element, 0xFFFF = spill-in
    __shared__ uint16_t lgid[SEG_SPAN];       // local group id of the owned elements
    __shared__ uint16_t cnt[TB / 64][SEG_DIGITS];
    __shared__ uint32_t dbase[SEG_DIGITS];    // also fz | rz of the boundary scans (2 x TB words)
    __shared__ uint32_t sm[TB / 64 + 1];
    __shared__ uint32_t s_fo, s_oe;
    __shared__ uint8_t firstflag[TB + 1];
    uint32_t *const g = u_g_idx;
    uint16_t *const idxA = reinterpret_cast<uint16_t *>(u_g_idx), *const idxB = idxA + SEG_SPAN;
    uint32_t *const fz = dbase, *const rz = dbase + TB;
    static_assert(SEG_DIGITS >= 2 * TB, "fz | rz live in dbase");

    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    const int tid = threadIdx.x;
    for (uint32_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        __syncthreads();                      // the LDS of the previous window is free
        const uint32_t base = win * SEG_TILE;
        const uint32_t avail = (m - base < (uint32_t)SEG_SPAN) ? m - base : (uint32_t)SEG_SPAN;
        const bool list_ends = (base + avail == m);

        {   // the span's eight loads per thread in flight together (clamped; only slots < avail are kept)
            uint32_t gl[SEG_ITEMS];
#pragma unroll
            for (int k = 0; k < SEG_ITEMS; k++) { const uint32_t p = tid + k * TB; gl[k] = a_grp[base + (p < avail ? p : avail - 1)]; }
#pragma unroll
            for (int k = 0; k < SEG_ITEMS; k++) { const uint32_t p = tid + k * TB; if (p < avail) g[p] = gl[k]; }
        }
        const uint32_t gprev = (base > 0) ? a_grp[base - 1] : 0xFFFFFFFFu;
        if (tid == 0) { s_fo = 0xFFFFFFFFu; s_oe = 0; }
        __syncthreads();

        // ---- group starts (max-scan of head positions) and group ends (next head), blocked 8 per thread ----
        const uint32_t p0 = tid * SEG_ITEMS;
        uint32_t hd = 0;                           // head bits of my 8 positions
        uint32_t lasth = 0;                        // 1 + last head position in my segment
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) {
            uint32_t p = p0 + k;
            if (p < avail) {
                bool head = (p == 0) ? (base == 0 || g[0] != gprev) : (g[p] != g[p - 1]);
                if (head) { hd |= 1u << k; lasth = p + 1; }
            }
        }
        uint32_t incl = block_incl_scan<OpMax>(lasth, sm, nullptr);
        uint32_t prev = __shfl_up(incl, 1, 64);
        if (lane_id() == 0) prev = (tid == 0) ? 0u : sm[(tid >> 6) - 1];
        // next head after my segment: suffix-min over the first-head positions of later threads
        uint32_t firsth = 0xFFFFFFFFu;
#pragma unroll
        for (int k = SEG_ITEMS - 1; k >= 0; k--)
            if (hd & (1u << k)) firsth = p0 + k;
        fz[tid] = firsth;
        __syncthreads();
        uint32_t rv = fz[TB - 1 - tid];
        uint32_t rinc = block_incl_scan<OpMin>(rv, sm, nullptr);
        rz[tid] = rinc;
        __syncthreads();
        uint32_t after = (tid == TB - 1) ? 0xFFFFFFFFu : rz[TB - 2 - tid];
        // per position: group start / end
        uint32_t ge[SEG_ITEMS];
        {
            uint32_t nn = after;
#pragma unroll
            for (int k = SEG_ITEMS - 1; k >= 0; k--) {
                ge[k] = nn;                        // first head strictly after position p0+k (or none)
                if (hd & (1u << k)) nn = p0 + k;
            }
            uint32_t run = prev;                   // 1 + start of the current group, 0 = spill-in
#pragma unroll
            for (int k = 0; k < SEG_ITEMS; k++) {
                uint32_t p = p0 + k;
                if (hd & (1u << k)) run = p + 1;
                if (p < avail) gsl[p] = run ? (uint16_t)(run - 1) : (uint16_t)0xFFFF;
            }
        }
        __syncthreads();
        // ---- classify: size of the group of each position; owned = starts in my window and size <= SEG_TILE.  A group whose
        // end lies beyond the loaded span is larger than SEG_TILE by construction (it starts inside the first half) ----
        const uint32_t ph = (win > 0) ? PH[win - 1] : 0u;                              // 1 + last head before my window
        const uint32_t spill_start = ph > 0 ? ph - 1 : 0u;                              // global index
        uint32_t my_fo = 0xFFFFFFFFu, my_oe = 0;
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) {
            uint32_t p = p0 + k;
            if (p < avail) {
                const uint32_t gs = gsl[p];
                uint32_t end = ge[k];
                bool end_known = true;
                if (end == 0xFFFFFFFFu) { end = avail; end_known = list_ends; }
                uint32_t size;
                if (!end_known) size = 0xFFFFFFFFu;
                else if (gs == 0xFFFFu) size = base + end - spill_start;
                else size = end - gs;
                const bool large = size > (uint32_t)SEG_TILE;
                if (gs != 0xFFFFu && gs < (uint32_t)SEG_TILE && !large) {
                    if (p < my_fo) my_fo = p;
                    if (p + 1 > my_oe) my_oe = p + 1;
                }
            }
        }
        if (my_fo != 0xFFFFFFFFu) { atomicMin(&s_fo, my_fo); atomicMax(&s_oe, my_oe); }
        __syncthreads();
        const uint32_t fo = s_fo, oe = s_oe;
        if (fo == 0xFFFFFFFFu) continue;           // nothing owned (uniform)
        const uint32_t no = oe - fo;               // owned elements: a contiguous range of whole groups

        // ---- stage owned elements: suffix, key2, local group id ----
        // local group id = number of heads in [fo, p] - 1  (block scan over head counts, blocked layout)
        uint32_t hc = 0;
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) {
            uint32_t p = p0 + k;
            if (p >= fo && p < oe && (hd & (1u << k))) hc++;
        }
        uint32_t hinc = block_incl_scan<OpSum>(hc, sm, nullptr);
        {
            uint32_t run = hinc - hc;
#pragma unroll
            for (int k = 0; k < SEG_ITEMS; k++) {
                uint32_t p = p0 + k;
                if (p >= fo && p < oe) {
                    if (hd & (1u << k)) run++;
                    lgid[p - fo] = (uint16_t)(run - 1);
                }
            }
        }
        uint32_t kmax = 0;                         // the window's largest key2: the sort only needs passes over the bits it has
        {
            uint32_t kl[SEG_ITEMS];
#pragma unroll
            for (int k = 0; k < SEG_ITEMS; k++) { const uint32_t q = tid + k * TB; kl[k] = k2g[base + fo + (q < no ? q : no - 1)]; }
#pragma unroll
            for (int k = 0; k < SEG_ITEMS; k++) {
                const uint32_t q = tid + k * TB;
                if (q < no) {
                    k2[q] = kl[k];
                    kmax |= kl[k];
                    idxA[q] = (uint16_t)q;    // g[] is dead from here on (last read: the classification above)
                }
            }
        }
        {
            uint32_t kall;
            block_incl_scan<OpMax>(kmax, sm, &kall);                  // (an OR would do; the maximum of the ORs has the same top bit)
            kmax = kall;
        }
        __syncthreads();
        // key2 <= n needs 27 bits on a 64 MiB block, the host's bound (3 n: round 1 spreads run members) 28 -- and with ten bits of group
        // number that is a fifth 9-bit pass which the window's own largest key usually does not need
        const int key_bits_w = kmax ? 32 - __clz((int)kmax) : 1;
        const int kbw = key_bits_w < key_bits ? key_bits_w : key_bits;
        const uint32_t ngroups = (uint32_t)lgid[no - 1] + 1u;

        // ---- LSD radix sort of the index permutation by the composite key (lgid << kbw) | key2, 9 bits per pass ----
        uint16_t *src = idxA, *dst = idxB;
        const int w = tid >> 6, l = tid & 63;
        const uint64_t lt = lanemask_lt();
        const int gbits = (ngroups > 1u) ? 32 - __clz((int)(ngroups - 1u)) : 0;
        const int npass = (kbw + gbits + SEG_DBITS - 1) / SEG_DBITS;
        // each wave ranks a contiguous quarter of the owned range: only ceil(no / 256) iterations of 64 are live
        const int nit = (int)((no + TB - 1) / TB);
        const uint32_t wspan = (uint32_t)nit * 64u;
        for (int pass = 0; pass < npass; pass++) {
            const int shift = SEG_DBITS * pass;
            const int part = (shift + SEG_DBITS <= kbw) ? 0 : (shift >= kbw ? 2 : 1);   // digit from key2 / both / group id
            for (int i = tid; i < (TB / 64) * SEG_DIGITS / 2; i += TB) reinterpret_cast<uint32_t *>(&cnt[0][0])[i] = 0;
            __syncthreads();
            uint32_t rk[SEG_ITEMS], dg[SEG_ITEMS];
#pragma unroll
            for (int it = 0; it < SEG_ITEMS; it++) {
                if (it >= nit) break;
                const uint32_t q = w * wspan + it * 64 + l;
                const bool valid = q < no;
                const uint32_t id = valid ? src[q] : 0u;
                uint32_t d;
                if (part == 0) d = k2[id] >> shift;
                else if (part == 2) d = (uint32_t)lgid[id] >> (shift - kbw);
                else d = (k2[id] >> shift) | ((uint32_t)lgid[id] << (kbw - shift));
                d = valid ? (d & (uint32_t)(SEG_DIGITS - 1)) : 0u;
                dg[it] = d | (id << SEG_DBITS);
                const uint64_t mm = match_any<SEG_DBITS>(d, valid);
                const uint32_t below = (uint32_t)__popcll(mm & lt);
                const uint32_t c = valid ? cnt[w][d] : 0u;
                rk[it] = c + below;
                if (valid && below == 0) cnt[w][d] = (uint16_t)(c + (uint32_t)__popcll(mm));
            }
            __syncthreads();
            {   // per digit: exclusive over waves, then exclusive over digits (two digits per thread, in digit order)
                uint32_t s2[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int d = 2 * tid + e;
                    uint32_t s = 0;
#pragma unroll
                    for (int k = 0; k < TB / 64; k++) { uint32_t t = cnt[k][d]; cnt[k][d] = (uint16_t)s; s += t; }
                    s2[e] = s;
                }
                const uint32_t inc = block_incl_scan<OpSum>(s2[0] + s2[1], sm, nullptr);
                dbase[2 * tid] = inc - s2[0] - s2[1];
                dbase[2 * tid + 1] = inc - s2[1];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < SEG_ITEMS; it++) {
                if (it >= nit) break;
                const uint32_t q = w * wspan + it * 64 + l;
                if (q < no) {
                    const uint32_t d = dg[it] & (uint32_t)(SEG_DIGITS - 1);
                    dst[dbase[d] + cnt[w][d] + rk[it]] = (uint16_t)(dg[it] >> SEG_DBITS);
                }
            }
            __syncthreads();
            uint16_t *t = src; src = dst; dst = t;
        }

        // ---- re-rank (blocked 8 per thread over the sorted order) ----
        uint32_t ap[SEG_ITEMS], nh[SEG_ITEMS];
        uint32_t hmax = 0;
        // the eight suffixes (and their carried BWT bytes) this thread will place, and the old ranks of its eight positions: all
        // loads issued here, in flight together (clamped), used after the scan below
        uint32_t sv[SEG_ITEMS], og[SEG_ITEMS];
        uint8_t pvv[SEG_ITEMS];
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) {
            const uint32_t q = p0 + k, qc = q < no ? q : no - 1;
            const uint32_t from = base + fo + src[qc];
            sv[k] = a_sa[from];
            pvv[k] = a_prev[from];                                  // T[s - 1] travels with the suffix: no gather from the text
            og[k] = a_grp[base + fo + qc];
        }
        uint32_t rf[SEG_ITEMS];                                     // run-member flag of the group at my positions (rides with the rank)
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) { rf[k] = og[k] & RUNF; og[k] &= ~RUNF; }
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) {
            const uint32_t q = p0 + k;
            ap[k] = 0; nh[k] = 0;
            if (q < no) {
                const uint32_t id = src[q];
                const uint32_t pos = fo + q;                        // groups keep their positions through the sort
                ap[k] = og[k] + (pos - (uint32_t)gsl[pos]);
                bool head = (q == 0);
                if (!head) {
                    const uint32_t pid = src[q - 1];
                    head = (lgid[pid] != lgid[id]) || (k2[pid] != k2[id]);
                }
                nh[k] = head ? 1u : 0u;
                if (head) hmax = ap[k];
            }
        }
        uint32_t rincl = block_incl_scan<OpMax>(hmax, sm, nullptr);
        uint32_t rprev = __shfl_up(rincl, 1, 64);
        if (lane_id() == 0) rprev = (tid == 0) ? 0u : sm[(tid >> 6) - 1];
        // next-head flag of the element after my segment
        firstflag[tid] = (uint8_t)(nh[0] | (p0 >= no ? 1u : 0u));
        if (tid == 0) firstflag[TB] = 1;
        __syncthreads();
        uint32_t run = rprev;
#pragma unroll
        for (int k = 0; k < SEG_ITEMS; k++) {
            const uint32_t q = p0 + k;
            if (q < no) {
                if (nh[k]) run = ap[k];
                const bool next_head = (q + 1 >= no) ? true : (k + 1 < SEG_ITEMS ? (nh[k + 1] != 0) : (firstflag[tid + 1] != 0));
                const bool single = nh[k] && next_head;
                const uint32_t s = sv[k];
                const uint8_t pv = pvv[k];
                if (GDw && nh[k] && !single) GDw[run] = new_group_depth(GDr, og[k], rf[k] != 0u, k2[src[q]], first_round != 0, n);   // vmode: the depth of the new group
                if (run != og[k]) ISA[s] = run;                     // the sub-group that sorts first keeps the old group's rank: no store
                if (single) {
                    bwt[ap[k]] = pv;
                    if (SA) SA[ap[k]] = s;
                }
                b_sa[base + fo + q] = s;
                b_grp[base + fo + q] = run | (single ? DONE : 0u) | rf[k];
                b_prev[base + fo + q] = pv;
            }
        }
    }
}

// ---- large groups: segmented LSD radix sort in place on (key2, sa), tiles = pieces -----------------------------------------
// table layout: entry of (group, digit d, piece t of the group) = fp * NB + d * nt + t  -- group-major in list order, so the flat
// exclusive scan S gives  S[entry] - S[fp * NB] = offset inside the group
template <int DB>
__global__ __launch_bounds__(TB) void k_lg_hist(const uint32_t *__restrict__ key, const Piece *__restrict__ pieces, const SaState *__restrict__ st, int shift,
                                               uint32_t *__restrict__ table)
{
    constexpr int NB = 1 << DB;
    // the members of one group share the upper digits of their keys (ranks inside one old group): counted by wave match -- the
    // lanes with equal digits are found by ballots and one of them adds their number to the wave's counter (plain LDS
    // read-modify-write, one lane per address) -- instead of LDS atomics that serialise on the shared bins
    __shared__ uint32_t h[WAVES * NB];
    const uint32_t np = st->npieces;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint32_t *mine = h + w * NB;
    const uint64_t lt = lanemask_lt();
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const Piece q = pieces[p];
        __syncthreads();
        for (int i = threadIdx.x; i < WAVES * NB; i += TB) h[i] = 0;
        __syncthreads();
        uint32_t kv[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {                // loads first (clamped), then the LDS atomics
            const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l;
            kv[k] = key[q.begin + (e < q.count ? e : q.count - 1)];
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l;
            const bool valid = e < q.count;
            const uint32_t d = (kv[k] >> shift) & (uint32_t)(NB - 1);
            const uint64_t m = match_any<DB>(d, valid);
            if (valid && (m & lt) == 0ull) mine[d] += (uint32_t)__popcll(m);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < NB; d += TB) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < WAVES; k++) s += h[k * NB + d];
            table[(size_t)q.fp * NB + (size_t)d * q.nt + q.tl] = s;
        }
    }
}

template <int DB>
__global__ __launch_bounds__(TB) void k_lg_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, const uint8_t *__restrict__ pin,
                                                  uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, uint8_t *__restrict__ pout,
                                                  const Piece *__restrict__ pieces, const SaState *__restrict__ st, int shift, const uint32_t *__restrict__ S)
{
    constexpr int NB = 1 << DB;
    __shared__ uint32_t cnt[WAVES][NB];
    __shared__ uint32_t gbase[NB];
    const uint32_t np = st->npieces;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const Piece q = pieces[p];
        __syncthreads();
        for (int i = threadIdx.x; i < WAVES * NB; i += TB) (&cnt[0][0])[i] = 0;
        const uint32_t s0 = S[(size_t)q.fp * NB];
        for (int d = threadIdx.x; d < NB; d += TB) gbase[d] = q.gs + (S[(size_t)q.fp * NB + (size_t)d * q.nt + q.tl] - s0);
        __syncthreads();
        uint32_t key[WIN_ITEMS], val[WIN_ITEMS], rnk[WIN_ITEMS];
        uint8_t prv[WIN_ITEMS];                      // the suffix's BWT byte rides along (third member of the sorted record)
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {                // the piece's loads in flight together (clamped; masked by `valid` below)
            const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l, ec = q.begin + (e < q.count ? e : q.count - 1);
            key[k] = kin[ec];
            val[k] = vin[ec];
            prv[k] = pin[ec];
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l;
            const bool valid = e < q.count;
            if (!valid) key[k] = 0u;
            const uint32_t d = (key[k] >> shift) & (uint32_t)(NB - 1);
            const uint64_t mm = match_any<DB>(d, valid);
            const uint32_t below = (uint32_t)__popcll(mm & lt);
            const uint32_t c = valid ? cnt[w][d] : 0u;
            rnk[k] = c + below;
            if (valid && below == 0) cnt[w][d] = c + (uint32_t)__popcll(mm);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < NB; d += TB) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < WAVES; k++) { uint32_t t = cnt[k][d]; cnt[k][d] = s; s += t; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l;
            if (e < q.count) {
                const uint32_t d = (key[k] >> shift) & (uint32_t)(NB - 1);
                const uint32_t dst = gbase[d] + cnt[w][d] + rnk[k];
                kout[dst] = key[k];
                vout[dst] = val[k];
                pout[dst] = prv[k];
            }
        }
    }
}

// head words of a piece in the sorted order: a new group starts where key2 changes (and at the group's first slot);
// H[16] bit 0 = the slot after the piece starts a group (or the old group ends there)
__device__ __forceinline__ void lg_piece_heads(const uint32_t *__restrict__ key, const Piece &q, uint64_t *H)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint32_t kj[WIN_ITEMS], kp[WIN_ITEMS];
#pragma unroll
    for (int k = 0; k < WIN_ITEMS; k++) {                    // loads first (clamped), all in flight
        const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l, j = q.begin + (e < q.count ? e : q.count - 1);
        kj[k] = key[j];
        kp[k] = key[j > q.gs ? j - 1 : j];
    }
#pragma unroll
    for (int k = 0; k < WIN_ITEMS; k++) {
        const uint32_t e = w * (64 * WIN_ITEMS) + k * 64 + l;
        bool head = false;
        if (e < q.count) {
            const uint32_t j = q.begin + e;
            head = (j == q.gs) || (kj[k] != kp[k]);
        }
        const uint64_t b = __ballot(head);
        if (l == 0) H[w * WIN_ITEMS + k] = b;
    }
    if (threadIdx.x == 0) {
        const uint32_t jn = q.begin + q.count;
        H[16] = (jn >= q.ge || key[jn] != key[jn - 1]) ? 1ull : 0ull;
    }
}

// 1 + position of the last new head inside every piece (0: none)
__global__ __launch_bounds__(TB) void k_lg_heads(const uint32_t *__restrict__ key, const Piece *__restrict__ pieces, const SaState *__restrict__ st,
                                                uint32_t *__restrict__ pLast)
{
    __shared__ uint64_t H[17];
    const uint32_t np = st->npieces;
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const Piece q = pieces[p];
        __syncthreads();
        lg_piece_heads(key, q, H);
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const uint64_t hv = (l < 16) ? H[l] : 0ull;
            uint32_t last = hv ? q.begin + l * 64 + top_bit(hv) + 1u : 0u;
            last = wave_incl_max(last);
            if (l == 63) pLast[p] = last;
        }
    }
}

__global__ __launch_bounds__(WG1) void k_lg_scan(uint32_t *__restrict__ pLast, const SaState *__restrict__ st)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    wg_scan<OpMax, true, false>(pLast, pLast, st->npieces, 0u, sm);
}

// new ranks of the members of large groups -> ISA, finished suffixes -> BWT byte, everything back to the b-list
__global__ __launch_bounds__(TB) void k_lg_finish(const uint32_t *__restrict__ key, const uint32_t *__restrict__ val, const uint8_t *__restrict__ prv,
                                                 const uint32_t *__restrict__ a_grp,
                                                 const Piece *__restrict__ pieces, const SaState *__restrict__ st, const uint32_t *__restrict__ pCarry,
                                                 uint32_t *__restrict__ ISA, uint8_t *__restrict__ bwt, uint32_t *__restrict__ SA,
                                                 uint32_t *__restrict__ b_sa, uint32_t *__restrict__ b_grp, uint8_t *__restrict__ b_prev,
                                                 const uint32_t *__restrict__ GDr, uint32_t *__restrict__ GDw, int first_round, uint32_t n)
{
    __shared__ uint64_t H[17];
    __shared__ uint32_t LHW[16];
    const uint32_t np = st->npieces;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const Piece q = pieces[p];
        __syncthreads();
        lg_piece_heads(key, q, H);
        __syncthreads();
        const uint32_t carry = pCarry[p];
        if (threadIdx.x < 64) {
            const uint64_t hv = (l < 16) ? H[l] : 0ull;
            uint32_t last = hv ? q.begin + l * 64 + top_bit(hv) + 1u : 0u;
            last = wave_incl_max(last);
            if (l < 16) LHW[l] = last > carry ? last : carry;
        }
        __syncthreads();
        const uint32_t G0 = a_grp[q.gs];
        const uint32_t G = G0 & ~(DONE | RUNF);            // rank of the old group = SA position of its first member
        const uint32_t rf = G0 & RUNF;                     // run-member flag: rides with the rank
        uint32_t sl[WIN_ITEMS];
        uint8_t pl[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {              // the piece's loads in flight together (clamped)
            const uint32_t e = (w * WIN_ITEMS + k) * 64 + l, j = q.begin + (e < q.count ? e : q.count - 1);
            sl[k] = val[j];
            pl[k] = prv[j];
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const int word = w * WIN_ITEMS + k;
            const uint32_t e = word * 64 + l;
            if (e < q.count) {
                const uint32_t j = q.begin + e;
                const uint64_t hv = H[word];
                const uint64_t le = hv & mask_upto(l);
                const uint32_t hp = le ? q.begin + word * 64 + top_bit(le) : (word ? LHW[word - 1] : carry) - 1u;
                const uint32_t rank = G + (hp - q.gs);
                const bool head = (hv >> l) & 1ull;
                bool nh;
                if (e + 1 == q.count) nh = H[16] & 1ull;
                else nh = (l < 63) ? ((hv >> (l + 1)) & 1ull) : (H[word + 1] & 1ull);
                const bool single = head && nh;
                const uint32_t s = sl[k];
                const uint8_t pv = pl[k];
                if (GDw && head && !single) GDw[rank] = new_group_depth(GDr, G, rf != 0u, key[j], first_round != 0, n);   // vmode: the depth of the new group
                if (hp != q.gs) ISA[s] = rank;                      // the first sub-group keeps the old group's rank
                if (single) {
                    const uint32_t ap = G + (j - q.gs);
                    bwt[ap] = pv;
                    if (SA) SA[ap] = s;
                }
                b_sa[j] = s;
                b_grp[j] = rank | (single ? DONE : 0u) | rf;
                b_prev[j] = pv;
            }
        }
    }
}

// ---- flat exclusive scan of the piece tables (size known on the device only) ---------------------------------------------
constexpr int SC_ITEMS = 16, SC_TILE = TB * SC_ITEMS;
__global__ __launch_bounds__(TB) void k_tab_reduce(const uint32_t *__restrict__ in, const SaState *__restrict__ st, uint32_t nb, uint32_t *__restrict__ partial)
{
    __shared__ uint32_t sm[TB / 64 + 1];
    const size_t n = (size_t)st->npieces * nb;
    const size_t ntiles = (n + SC_TILE - 1) / SC_TILE;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
        uint32_t acc = 0;
        if (base + SC_ITEMS <= n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
            for (int k = 0; k < SC_ITEMS / 4; k++) { uint4 v = p[k]; acc += v.x + v.y + v.z + v.w; }
        } else {
            for (int k = 0; k < SC_ITEMS; k++) if (base + k < n) acc += in[base + k];
        }
        uint32_t tot;
        block_incl_scan<OpSum>(acc, sm, &tot);
        if (threadIdx.x == 0) partial[tile] = tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(WG1) void k_tab_partials(uint32_t *__restrict__ partial, const SaState *__restrict__ st, uint32_t nb)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    const size_t n = (size_t)st->npieces * nb;
    wg_scan<OpSum, true, false>(partial, partial, (uint32_t)((n + SC_TILE - 1) / SC_TILE), 0u, sm);
}
__global__ __launch_bounds__(TB) void k_tab_down(const uint32_t *in, uint32_t *out, const SaState *__restrict__ st, uint32_t nb, const uint32_t *__restrict__ partial)
{
    __shared__ uint32_t sm[TB / 64 + 1];
    const size_t n = (size_t)st->npieces * nb;
    const size_t ntiles = (n + SC_TILE - 1) / SC_TILE;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
        uint32_t v[SC_ITEMS];
        if (base + SC_ITEMS <= n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
            for (int k = 0; k < SC_ITEMS / 4; k++) { uint4 q = p[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
        } else {
#pragma unroll
            for (int k = 0; k < SC_ITEMS; k++) v[k] = (base + k < n) ? in[base + k] : 0u;
        }
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) acc += v[k];
        const uint32_t inc = block_incl_scan<OpSum>(acc, sm, nullptr);
        uint32_t prev = __shfl_up(inc, 1, 64);
        if (lane_id() == 0) prev = (threadIdx.x == 0) ? 0u : sm[(threadIdx.x >> 6) - 1];
        uint32_t run = partial[tile] + prev;
        uint32_t o[SC_ITEMS];
#pragma unroll
        for (int k = 0; k < SC_ITEMS; k++) { o[k] = run; run += v[k]; }
        if (base + SC_ITEMS <= n) {
            uint4 *p = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
            for (int k = 0; k < SC_ITEMS / 4; k++) p[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < SC_ITEMS; k++) if (base + k < n) out[base + k] = o[k];
        }
        __syncthreads();
    }
}

// ---- compaction of a round's output: the unresolved suffixes, in order, become the next round's active list -------------
__global__ __launch_bounds__(TB) void k_cmp_count(const uint32_t *__restrict__ b_grp, const SaState *__restrict__ st, int par, uint32_t *__restrict__ tSurv)
{
    const uint32_t m = st->m[par];
    const uint32_t ntiles = (m + CT - 1) / CT;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __shared__ uint32_t ws[WAVES];
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT;
        uint32_t c = 0;
        uint32_t gv[CT_ITEMS];
#pragma unroll
        for (int k = 0; k < CT_ITEMS; k++) {                // loads first, all in flight (clamped, masked below)
            const uint32_t j = base + w * (64 * CT_ITEMS) + k * 64 + l;
            gv[k] = b_grp[j < m ? j : m - 1];
        }
#pragma unroll
        for (int k = 0; k < CT_ITEMS; k++) {
            const uint32_t j = base + w * (64 * CT_ITEMS) + k * 64 + l;
            c += (j < m && !(gv[k] & DONE)) ? 1u : 0u;
        }
        c = wave_sum(c);
        __syncthreads();
        if (l == 0) ws[w] = c;
        __syncthreads();
        if (threadIdx.x == 0) tSurv[tile] = ws[0] + ws[1] + ws[2] + ws[3];
    }
}
__global__ __launch_bounds__(WG1) void k_cmp_scan(uint32_t *__restrict__ tSurv, SaState *__restrict__ st, int par, int round)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    const uint32_t m = st->m[par];
    const uint32_t total = wg_scan<OpSum, true, false>(tSurv, tSurv, (m + CT - 1) / CT, 0u, sm);
    if (threadIdx.x == 0) {
        st->m[par ^ 1] = total;
        if (round + 1 < JPK_SA_MAX_ROUNDS) st->round_m[round + 1] = total;
    }
}
__global__ __launch_bounds__(TB) void k_cmp_scatter(const uint32_t *__restrict__ b_sa, const uint32_t *__restrict__ b_grp, const uint8_t *__restrict__ b_prev,
                                                   const SaState *__restrict__ st, int par, const uint32_t *__restrict__ tOff,
                                                   uint32_t *__restrict__ a_sa, uint32_t *__restrict__ a_grp, uint8_t *__restrict__ a_prev)
{
    __shared__ uint64_t SV[64];
    __shared__ uint32_t SW[64];
    const uint32_t m = st->m[par];
    const uint32_t ntiles = (m + CT - 1) / CT;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT;
        __syncthreads();
        uint32_t gv[CT_ITEMS], sv[CT_ITEMS];
        uint8_t pv[CT_ITEMS];
#pragma unroll
        for (int k = 0; k < CT_ITEMS; k++) {                // every load of the tile in flight at once (clamped, masked below)
            const uint32_t j = base + w * (64 * CT_ITEMS) + k * 64 + l, jc = j < m ? j : m - 1;
            gv[k] = b_grp[jc];
            sv[k] = b_sa[jc];
            pv[k] = b_prev[jc];
        }
#pragma unroll
        for (int k = 0; k < CT_ITEMS; k++) {
            const uint32_t j = base + w * (64 * CT_ITEMS) + k * 64 + l;
            if (j >= m) gv[k] = DONE;
            const uint64_t b = __ballot(!(gv[k] & DONE));
            if (l == 0) SV[w * CT_ITEMS + k] = b;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t cnt = (uint32_t)__popcll(SV[l]);
            const uint32_t inc = wave_incl_sum(cnt);
            SW[l] = tOff[tile] + inc - cnt;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CT_ITEMS; k++) {
            const int word = w * CT_ITEMS + k;
            if (!(gv[k] & DONE)) {
                const uint32_t pos = SW[word] + (uint32_t)__popcll(SV[word] & mask_below(l));
                a_sa[pos] = sv[k];
                a_grp[pos] = gv[k];
                a_prev[pos] = pv[k];
            }
        }
    }
}

// ---- the pair rule: long repeats do not double their way out (round 5) --------------------------------------------------------
// A repeat T[u .. u+L) == T[v .. v+L) leaves L groups {u+q, v+q} that doubling resolves only once its distance exceeds L - q:
// log2(L) rounds over all of them (a 1 MiB segment repeated: 24 rounds; divsufsort.cpp:1427-1520 has no such cliff -- it induces the
// order of most suffixes from their successors).  This is that induction step on the active list, between two doubling rounds.
// Members of a group sit in DESCENDING text position (round 0 is a stable sort fed in descending position, every later sort is
// stable).  For a member s that is not the first of its group let P[s] = s' - s, s' = the member in front of it; P = 0 elsewhere.
// s and s + P[s] are in one group, so they share their first byte, so  order(s, s + p) = order(s + 1, s + 1 + p).  Along a maximal
// stretch of positions [a, x] with one non-zero P = p the argument repeats: every pair (y, y + p) of the stretch is ordered like
// (x + 1, x + 1 + p), and THAT pair is decided now if the two suffixes lie in different groups (their ranks compare) or x + 1 + p is
// the end of the text / block (the empty suffix is the smaller one), or -- in a second pass over the stretches -- if they lie in one group
// and the neighbouring pairs between them all carry one verdict.  A group all of whose neighbouring pairs carry the same decided
// verdict is totally ordered by position: its members are finished with ranks G, G + 1, ...; every other group stays exactly as it
// was (the doubling distance does not change).  p = 1 is the run rule's case.  tests/pair_rule_model.py states the same in Python and
// tests/test_pair_rule_model.py checks it against a brute-force suffix sort on repeat-heavy texts.
//   k_pair_dist    P[s] (random 4-byte store per member), FH / LH of every window
//   k_pair_first / k_pair_scan / k_pair_fill   first stretch end at or after every position (the k_run_* scheme on P instead of T),
//                  verdict of that end -> V[y] for every y with P[y] != 0   (1: the lower position is smaller, 2: the higher, 0: open)
//   k_pair_mark    VL[j] = verdict of list slot j (0xFF for a group's first member); BAD[G] = 1 for a group with an open or a
//                  dissenting pair (G = the group's rank: the list is in rank order, so these accesses walk BAD upwards)
//   k_pair_finish  members of the other groups: rank -> ISA, BWT byte, DONE; everything to the b-list; compaction follows as in a round
constexpr uint8_t PV_HEAD = 0xFF;
constexpr uint32_t PREP = 0x80000000u;      // P[z]: the stretch was carried THROUGH z by k_pair_repair (z's own neighbour is nearer): distances are < 2^30
__global__ __launch_bounds__(TB) void k_pair_dist(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, const SaState *__restrict__ st, int par,
                                                 uint32_t *__restrict__ P, uint32_t *__restrict__ FH, uint32_t *__restrict__ LH)
{
    __shared__ uint64_t H[16];
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (uint32_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        const uint32_t base = win * SEG_TILE;
        __syncthreads();
        uint32_t s[WIN_ITEMS], sp[WIN_ITEMS], gj[WIN_ITEMS], gp[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l, jc = j < m ? j : m - 1, jp = jc ? jc - 1 : 0;
            s[k] = a_sa[jc];
            sp[k] = a_sa[jp];
            gj[k] = a_grp[jc];
            gp[k] = a_grp[jp];
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l;
            bool head = false;
            if (j < m) {
                head = (j == 0) || (gj[k] != gp[k]);
                if (!head) P[s[k]] = sp[k] - s[k];
            }
            const uint64_t b = __ballot(head);
            if (l == 0) H[w * WIN_ITEMS + k] = b;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint64_t hv = (l < 16) ? H[l] : 0ull;
            uint32_t first = hv ? base + l * 64 + (uint32_t)__builtin_ctzll(hv) + 1u : NONE;
            uint32_t last = hv ? base + l * 64 + top_bit(hv) + 1u : 0u;
            first = wave_incl_min(first);
            last = wave_incl_max(last);
            if (l == 63) { FH[win] = (first == NONE) ? 0u : first; LH[win] = last; }
        }
    }
}

// A group that mixes two repeats cuts the stretches of BOTH: its members' neighbours are nearer than the repeats' distance p, so the
// positions of an inner repeat (a phrase that occurs twice inside a segment that is itself repeated) end every stretch that reaches them
// in an open pair -- and a segment with a thousand inner repeats has a thousand stretch pieces, all but the last open.  But the
// induction only needs T[z] = T[z + p], and z and z + p ARE in one group there: the thread of a stretch end walks on through such
// positions and writes p (with PREP) over their own distance until the stretch's own distance returns, the two suffixes part, or
// the text ends.  The walk reads consecutive ranks (z and z + p advance together).  A position that was walked through gives up its own
// pair (V = 0: its group, a mixed one, waits for the doubling rounds).  Inner repeats start walks of their own through the same positions:
// the largest distance wins (atomicMax; PREP is the top bit, so any carried distance beats a position's own) -- the outer repeat's
// stretch is the long one.  Any winner is a true same-group distance.
constexpr int PAIR_WALK_MAX = 2048;
__global__ __launch_bounds__(TB) void k_pair_repair(uint32_t *P, uint32_t n, const uint32_t *__restrict__ ISA, const uint8_t *__restrict__ blk,
                                                   const uint32_t *__restrict__ bend, SaState *__restrict__ st, uint32_t budget)
{
    for (uint32_t x = blockIdx.x * TB + threadIdx.x; x + 1u < n; x += gridDim.x * TB) {
        const uint32_t p = P[x];
        if (p == 0u || (p & PREP)) continue;
        if ((P[x + 1u] & ~PREP) == p) continue;                   // not a stretch end
        const uint32_t lim = bend ? bend[blk[x]] : n;
        uint32_t z = x + 1u;
        for (int step = 0; step < PAIR_WALK_MAX; step++, z++) {
            // (all walks of a pair round together stay below 8 n positions: an input built to make every position a stretch end with a long
            // walk behind it costs a bounded pass, and the stretches it leaves cut wait for the doubling rounds)
            if ((step & 31) == 31 && atomicAdd(&st->pair_steps, 32u) > budget) break;
            if ((uint64_t)z + p >= lim) break;                    // the pair behind the stretch reaches the end of the text: decided there
            if ((P[z] & ~PREP) == p) break;                       // the stretch's own distance again: it runs on by itself
            if (ISA[z] != ISA[z + p]) break;                      // the two suffixes part: decided by their ranks
            atomicMax(&P[z], p | PREP);                            // the LARGEST distance carried through z wins: the outer repeat, not an inner one
        }
    }
}

// P has n + 1 entries, P[n] = 0: position x ends a stretch when P[x] != P[x + 1] (distances compared without PREP)
__global__ __launch_bounds__(TB) void k_pair_first(const uint32_t *__restrict__ P, uint32_t n, uint32_t *__restrict__ tFirst)
{
    __shared__ uint32_t sm[TB / 64 + 1];
    const uint32_t ntiles = (n + CT - 1) / CT;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT;
        uint32_t pa[CT_ITEMS], pb[CT_ITEMS];
#pragma unroll
        for (int k = 0; k < CT_ITEMS; k++) {
            const uint32_t i = base + k * TB + threadIdx.x, ic = i < n ? i : n - 1;
            pa[k] = P[ic] & ~PREP;
            pb[k] = P[ic + 1] & ~PREP;
        }
        uint32_t first = NONE;
#pragma unroll
        for (int k = CT_ITEMS - 1; k >= 0; k--) {
            const uint32_t i = base + k * TB + threadIdx.x;
            if (i < n && pa[k] != pb[k]) first = i;
        }
        uint32_t tot;
        block_incl_scan<OpMin>(first, sm, &tot);
        if (threadIdx.x == 0) tFirst[tile] = tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(WG1) void k_pair_scan(uint32_t *__restrict__ tFirst, uint32_t n)
{
    __shared__ uint32_t sm[WG1 / 64 + 1];
    wg_scan<OpMin, true, true>(tFirst, tFirst, (n + CT - 1) / CT, NONE, sm);      // first stretch end in any LATER tile
}
// verdict of the stretch that ends at x with distance p: the pair (x + 1, x + 1 + p).  When the two lie in ONE group with other
// members between them (x + 1 belongs to a group that mixes two repeats: its neighbour is nearer than p), the pair is decided by the
// chain of neighbouring pairs from x + 1 up to x + 1 + p if all of them carry one verdict already (`chain`: V holds the verdicts of an
// earlier pass over the stretches; a value only ever changes from open to decided, so reading it while this pass writes is safe).
__device__ __forceinline__ uint32_t pair_verdict(uint32_t x, uint32_t p, const uint32_t *__restrict__ ISA, uint32_t n, const uint8_t *__restrict__ blk,
                                                 const uint32_t *__restrict__ bend, const uint32_t *__restrict__ P, const uint8_t *V, bool chain)
{
    const uint32_t lim = bend ? bend[blk[x]] : n;
    const uint64_t b = (uint64_t)x + 1u + p;                     // x + p is a member's position (< lim), so b <= lim
    if (b >= lim) return 2u;
    const uint32_t ra = ISA[x + 1u], rb = ISA[b];
    if (ra != rb) return ra < rb ? 1u : 2u;
    if (!chain) return 0u;
    uint32_t z = x + 1u, v = 0u;
    for (int step = 0; step < 8 && z < (uint32_t)b; step++) {
        const uint32_t pz = P[z];
        if (pz == 0u || (pz & PREP)) return 0u;                  // (the first member of its group, or a carried-through position: the chain does not reach b)
        const uint32_t vz = V[z];
        if (vz == 0u || (v && vz != v)) return 0u;
        v = vz;
        z += pz;
    }
    return z == (uint32_t)b ? v : 0u;
}
__global__ __launch_bounds__(TB) void k_pair_fill(const uint32_t *__restrict__ P, uint32_t n, const uint32_t *__restrict__ tAfter, const uint32_t *__restrict__ ISA,
                                                 uint8_t *V, const uint8_t *__restrict__ blk, const uint32_t *__restrict__ bend, int chain)
{
    __shared__ uint32_t sm[TB / 64 + 1];
    __shared__ uint32_t rv[TB];
    const uint32_t ntiles = (n + CT - 1) / CT;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t base = tile * CT, p0 = base + threadIdx.x * CT_ITEMS;        // blocked: sixteen consecutive positions per thread
        uint32_t pv[CT_ITEMS + 1];
#pragma unroll
        for (int k = 0; k <= CT_ITEMS; k++) { const uint32_t i = p0 + k; pv[k] = P[i < n ? i : n]; }
        uint32_t bits = 0, first = NONE, any = 0, rep = 0;
#pragma unroll
        for (int k = CT_ITEMS - 1; k >= 0; k--) {
            const uint32_t i = p0 + k;
            if (i < n) {
                any |= pv[k];
                if (pv[k] & PREP) rep |= 1u << k;
                if ((pv[k] & ~PREP) != (pv[k + 1] & ~PREP)) { bits |= 1u << k; first = i; }
            }
        }
#pragma unroll
        for (int k = 0; k <= CT_ITEMS; k++) pv[k] &= ~PREP;
        __syncthreads();                                                              // rv of the previous tile has been read
        rv[TB - 1 - threadIdx.x] = first;
        __syncthreads();
        const uint32_t rinc = block_incl_scan<OpMin>(rv[threadIdx.x], sm, nullptr);   // index u: min over the threads >= TB - 1 - u
        __syncthreads();
        rv[threadIdx.x] = rinc;
        __syncthreads();
        uint32_t nb = (threadIdx.x == TB - 1) ? NONE : rv[TB - 2 - threadIdx.x];
        if (nb == NONE) nb = tAfter[tile];
        if (!any) continue;                                                           // (nothing of mine is a member; the barriers above are behind us)
        uint32_t vnb = NONE;                                                          // verdict of the stretch end nb: not computed yet
#pragma unroll
        for (int k = CT_ITEMS - 1; k >= 0; k--) {
            const uint32_t i = p0 + k;
            if (i < n) {
                const uint32_t p = pv[k];
                if (bits & (1u << k)) { nb = i; vnb = p ? pair_verdict(i, p, ISA, n, blk, bend, P, V, chain != 0) : 0u; }
                if (p) {
                    if (vnb == NONE) vnb = pair_verdict(nb, p, ISA, n, blk, bend, P, V, chain != 0);   // the stretch runs on into a later thread: P[nb] == p
                    V[i] = (rep & (1u << k)) ? (uint8_t)0 : (uint8_t)vnb;      // a position the stretch was carried through: ITS pair stays open
                }
            }
        }
    }
}

__global__ __launch_bounds__(TB) void k_pair_mark(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, const SaState *__restrict__ st, int par,
                                                 const uint8_t *__restrict__ V, uint8_t *__restrict__ VL, uint8_t *__restrict__ BAD)
{
    __shared__ uint8_t cs[SEG_TILE + 1];       // cs[q + 1] = code of local slot q (verdict, PV_HEAD for a group's first member); cs[0]: the slot in front of the window
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (uint32_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        const uint32_t base = win * SEG_TILE;
        __syncthreads();
        uint32_t s[WIN_ITEMS], gj[WIN_ITEMS], gp[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l, jc = j < m ? j : m - 1;
            s[k] = a_sa[jc];
            gj[k] = a_grp[jc];
            gp[k] = a_grp[jc ? jc - 1 : 0];
        }
        uint8_t c[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l;
            const bool head = (j == 0) || (gj[k] != gp[k]);
            c[k] = PV_HEAD;
            if (j < m && !head) c[k] = V[s[k]];
        }
        if (threadIdx.x == 0) {
            uint8_t c0 = PV_HEAD;
            if (base >= 2u && a_grp[base - 1] == a_grp[base - 2]) c0 = V[a_sa[base - 1]];
            cs[0] = c0;
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t q = w * (64 * WIN_ITEMS) + k * 64 + l;
            cs[q + 1] = c[k];
            if (base + q < m) VL[base + q] = c[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t q = w * (64 * WIN_ITEMS) + k * 64 + l;
            if (base + q < m && c[k] != PV_HEAD) {
                const uint8_t pc = cs[q];
                if (c[k] == 0 || (pc != PV_HEAD && pc != c[k])) BAD[gj[k] & ~(RUNF | DONE)] = 1;      // (same value from every writer)
            }
        }
    }
}

__global__ __launch_bounds__(TB) void k_pair_finish(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, const uint8_t *__restrict__ a_prev,
                                                   const SaState *__restrict__ st, int par, const uint32_t *__restrict__ PH, const uint32_t *__restrict__ NH,
                                                   const uint8_t *__restrict__ VL, const uint8_t *__restrict__ BAD,
                                                   uint32_t *__restrict__ ISA, uint8_t *__restrict__ bwt, uint32_t *__restrict__ SA,
                                                   uint32_t *__restrict__ b_sa, uint32_t *__restrict__ b_grp, uint8_t *__restrict__ b_prev)
{
    __shared__ uint64_t H[16];
    __shared__ uint32_t LHW[16];               // 1 + last head position at or before the end of word l (carry included)
    __shared__ uint32_t NHW[16];               // first head position in a word AFTER word l (the next window's included)
    const uint32_t m = st->m[par];
    const uint32_t nwin = (m + SEG_TILE - 1) / SEG_TILE;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (uint32_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        const uint32_t base = win * SEG_TILE;
        __syncthreads();
        uint32_t s[WIN_ITEMS], gj[WIN_ITEMS], gp[WIN_ITEMS];
        uint8_t pv[WIN_ITEMS], cj[WIN_ITEMS], cn[WIN_ITEMS], bad[WIN_ITEMS];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l, jc = j < m ? j : m - 1;
            s[k] = a_sa[jc];
            gj[k] = a_grp[jc];
            gp[k] = a_grp[jc ? jc - 1 : 0];
            pv[k] = a_prev[jc];
            cj[k] = VL[jc];
            cn[k] = VL[jc + 1 < m ? jc + 1 : jc];
        }
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) bad[k] = BAD[gj[k] & ~(RUNF | DONE)];
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const uint32_t j = base + w * (64 * WIN_ITEMS) + k * 64 + l;
            const bool head = j < m && ((j == 0) || (gj[k] != gp[k]));
            const uint64_t b = __ballot(head);
            if (l == 0) H[w * WIN_ITEMS + k] = b;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint64_t hv = (l < 16) ? H[l] : 0ull;
            const uint32_t carry = win ? PH[win - 1] : 0u;
            uint32_t last = hv ? base + l * 64 + top_bit(hv) + 1u : 0u;
            last = wave_incl_max(last);
            if (l < 16) LHW[l] = last > carry ? last : carry;
            // first head in the words after l: inclusive min-scan over the words in reverse order, shifted by one
            const int rl = 15 - l;                                                  // lane l holds word 15 - l
            const uint64_t hr = (l < 16) ? H[rl] : 0ull;
            uint32_t firstr = hr ? base + rl * 64 + (uint32_t)__builtin_ctzll(hr) : NONE;
            firstr = wave_incl_min(firstr);                                          // lane l: min over words >= 15 - l
            const uint32_t after = NH[win];
            const uint32_t prevlane = __shfl_up(firstr, 1, 64);                      // word x = 15 - l wants the min over words > x = lane l - 1's value
            if (l < 16) {
                uint32_t v = (l == 0) ? NONE : prevlane;
                if (v == NONE) v = after;
                NHW[rl] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WIN_ITEMS; k++) {
            const int word = w * WIN_ITEMS + k;
            const uint32_t j = base + word * 64 + l;
            if (j < m) {
                const uint64_t hv = H[word];
                const bool head = (hv >> l) & 1ull;
                const uint32_t G = gj[k] & ~(RUNF | DONE);
                uint32_t out_g = gj[k];
                if (!bad[k]) {
                    const uint64_t le = hv & mask_upto(l);
                    const uint32_t gs = le ? base + word * 64 + top_bit(le) : (word ? LHW[word - 1] : (win ? PH[win - 1] : 0u)) - 1u;
                    const uint64_t gt = (l < 63) ? (hv >> (l + 1)) : 0ull;
                    uint32_t ge = gt ? j + 1u + (uint32_t)__builtin_ctzll(gt) : NHW[word];
                    if (ge > m) ge = m;
                    const uint8_t v = head ? cn[k] : cj[k];              // a group has at least two members: the slot behind a head is its pair
                    const uint32_t r = (v == 2) ? G + (j - gs) : G + (ge - 1u - j);
                    if (r != G) ISA[s[k]] = r;
                    bwt[r] = pv[k];
                    if (SA) SA[r] = s[k];
                    out_g = r | DONE;
                }
                b_sa[j] = s[k];
                b_grp[j] = out_g;
                b_prev[j] = pv[k];
            }
        }
    }
}

// ---- BWT image (bwt.cpp:44-61) -------------------------------------------------------------------------------------
// bwt_sa[i] = T[SA[i] - 1] was emitted when suffix SA[i] was resolved; the image drops the row of suffix 0 (index idx = ISA[0])
// and starts with T[n-1]
__global__ __launch_bounds__(TB) void k_bwt_image(const uint8_t *__restrict__ T, const uint8_t *__restrict__ bwt_sa, const uint32_t *__restrict__ ISA,
                                                 uint32_t n, uint8_t *__restrict__ out)
{
    const uint32_t idx = ISA[0];
    for (uint32_t i = blockIdx.x * TB + threadIdx.x; i < n; i += gridDim.x * TB) {
        if (i == 0) out[0] = T[n - 1];
        if (i != idx) out[(i < idx) ? i + 1 : i] = bwt_sa[i];
    }
}

__global__ void k_bwt_trailer(const uint8_t *__restrict__ T, const uint32_t *__restrict__ ISA, uint32_t n, uint32_t len, uint8_t *__restrict__ out)
{
    uint32_t t = threadIdx.x;
    uint32_t step = n / JPK_BWT_UNITS;
    if (t < JPK_BWT_UNITS) {
        uint32_t v = ISA[(size_t)t * step] + 1u;
        uint8_t *p = out + len + 4 * t;
        p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
    }
    if (t < len - n) out[n + t] = T[n + t];      // raw tail (bwt.cpp:32-33), at most 119 bytes
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct SaBufs {
    uint64_t *keysA, *keysB;
    uint32_t *valsA, *valsB, *ISA, *SA, *a_sa, *a_grp, *b_sa, *b_grp, *k2, *k2alt, *sa_alt, *table;
    uint32_t *tA, *tB;          // per-tile scalars
    uint32_t *FH, *LH, *PH, *NH, *PC, *pLast, *partial, *scratch;
    uint8_t *bwt;
    uint8_t *a_prev, *b_prev, *p_alt;      // T[sa - 1] of every active suffix: travels with (sa, rank) through the rounds
    uint32_t *RL;                          // remaining run length per text position (written only when round 0 leaves run members behind)
    uint32_t *GD[2] = {nullptr, nullptr};  // variable-length keys: depth of every unresolved group by its rank, read side / write side of a round
    uint8_t *D0 = nullptr;                 // ... and the depth of every slot's key (rides through the radix sort in the value's upper bits up to 2^28 bytes)
    uint32_t *ctab = nullptr;              // ... and the table of the context codes (256 bytes + 1024 pairs of bytes, 256 codes each)
    uint16_t *ctxmap = nullptr;
    const uint8_t *blk = nullptr;          // group sort: block number of every text position, and where every block ends (device)
    const uint32_t *bend = nullptr;
    Piece *pieces;
    SaState *state;
};

int lg_digit_bits(uint32_t n, int *npass)
{
    const int kbits = jpk_bits_for(3u * n);        // key2 <= n, or <= 3n in a group of run members in round 1
    int np = (kbits + 7) / 8;
    if (np < 1) np = 1;
    int db = (kbits + np - 1) / np;
    if (db < 4) db = 4;
    *npass = np;
    return db;                                     // 4..8
}

// JPK_KEY_BITS=8 keeps round 0's keys at one byte per symbol whatever the alphabet (the comparator of the packed keys; 0 = from the alphabet)
int key_force_bits()
{
    static const int v = [] { const char *e = getenv("JPK_KEY_BITS"); const int x = e ? atoi(e) : 0; return x < 0 ? 0 : (x > 8 ? 8 : x); }();
    return v;
}

// JPK_R0_LOOKBACK=0: round 0's head / survivor bookkeeping in two passes (k_r0_count + k_r0_scan in front of k_r0_finish: the comparator)
bool r0_lookback()
{
    static const bool v = [] { const char *e = getenv("JPK_R0_LOOKBACK"); return e ? atoi(e) != 0 : true; }();
    return v;
}
// JPK_KEY_ORDER=0 / 1: the variable-length keys use nothing above the order-0 / order-1 code (comparators; default 2)
int key_order()
{
    static const int v = [] { const char *e = getenv("JPK_KEY_ORDER"); const int x = e ? atoi(e) : 2; return x < 0 ? 0 : x > 2 ? 2 : x; }();
    return v;
}
// JPK_VARKEYS=0: fixed-width keys whatever the block (the comparator of the variable-length keys)
bool var_keys_on()
{
    static const bool v = [] { const char *e = getenv("JPK_VARKEYS"); return e ? atoi(e) != 0 : true; }();
    return v;
}
// variable-length keys: sorts (one block, or a group of small ones) of at most 2^28 bytes (the key's depth rides in the spare bits of the 32-bit suffix number:
// six up to 2^26 bytes, five up to 2^27, four -- depths clamped at 15 -- up to 2^28), the one-pass radix form, no forced code width
// (round 6: above 2^28 bytes -- JPK_MAX_BLOCKSIZE is 1000 MiB, format.hpp:22 -- the depths stay in the slots' own array: tag shift 32, r0_short)
bool var_keys_eligible(size_t n, bool group) { (void)group; return n < ((size_t)1 << 30) && var_keys_on() && jpk_radix_onesweep() && key_force_bits() == 0; }
int var_tag_shift(size_t n) { if (n > ((size_t)1 << 28)) return 32; int s = 26; while (((size_t)1 << s) < n) s++; return s; }

void sa_layout(Arena &a, size_t n, SaBufs &b, bool var)
{
    const size_t nwin = n / SEG_TILE + 2, ntile = n / CT + 2;
    memset(&b, 0, sizeof b);
    // (+ CT: round 0 sorts whole tiles -- the pack kernels fill the slots behind the text's end with the largest key)
    b.keysA = a.get<uint64_t>(n + CT);
    b.keysB = a.get<uint64_t>(n + CT);
    b.valsA = a.get<uint32_t>(n + CT);
    b.valsB = a.get<uint32_t>(n + CT);
    b.ISA = a.get<uint32_t>(n);
    b.a_sa = a.get<uint32_t>(n);
    b.a_grp = a.get<uint32_t>(n);
    b.bwt = a.get<uint8_t>(n + CT);
    b.a_prev = a.get<uint8_t>(n);
    b.b_prev = a.get<uint8_t>(n);
    b.p_alt = a.get<uint8_t>(n);
    b.RL = a.get<uint32_t>(n);
    if (var) {
        b.GD[0] = a.get<uint32_t>(n);
        b.GD[1] = a.get<uint32_t>(n);
        // (the slots' depths are read by the radix sort's first pass; the BWT bytes arrive from k_r0_finish on -- which, above 2^28 bytes, still reads the depths)
        b.D0 = var_tag_shift(n) < 32 ? b.bwt : a.get<uint8_t>(n + CT);
        // the context codes: sampled counts, then code | length << 27 of byte s behind byte c (rows 0..255) or behind a chosen pair of bytes
        b.ctab = a.get<uint32_t>(256 * (256 + JPK_O2_CLASSES));
        b.ctxmap = a.get<uint16_t>(65536);    // the row that codes a symbol behind the bytes c2 c1
    }
    const size_t nbmax = 256;
    b.table = a.get<uint32_t>(nbmax * 2 * nwin);
    b.partial = a.get<uint32_t>(nbmax * 2 * nwin / SC_TILE + 64);
    b.tA = a.get<uint32_t>(ntile);
    b.tB = a.get<uint32_t>(ntile);
    b.FH = a.get<uint32_t>(nwin);
    b.LH = a.get<uint32_t>(nwin);
    b.PH = a.get<uint32_t>(nwin);
    b.NH = a.get<uint32_t>(nwin);
    b.PC = a.get<uint32_t>(nwin);
    b.pLast = a.get<uint32_t>(2 * nwin);
    b.pieces = a.get<Piece>(2 * nwin);
    b.state = a.get<SaState>(1);
    b.scratch = a.get<uint32_t>(jpk_radix_scratch_words(n + CT));
}

inline unsigned cap_grid(size_t work, unsigned per_block, unsigned cap)
{
    size_t g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

template <int DB>
void launch_lg_pass(jpk_ctx *ctx, SaBufs &b, const uint32_t *kin, const uint32_t *vin, const uint8_t *pin, uint32_t *kout, uint32_t *vout, uint8_t *pout, int shift,
                    unsigned gp, unsigned gt)
{
    constexpr uint32_t NB = 1u << DB;
    JPK_LAUNCH(ctx, PROF_LG_HIST, 0, (k_lg_hist<DB>), dim3(gp), dim3(TB), kin, b.pieces, b.state, shift, b.table);
    JPK_LAUNCH(ctx, PROF_SCAN, 0, k_tab_reduce, dim3(gt), dim3(TB), b.table, b.state, NB, b.partial);
    JPK_LAUNCH(ctx, PROF_SCAN, 0, k_tab_partials, dim3(1), dim3(WG1), b.partial, b.state, NB);
    JPK_LAUNCH(ctx, PROF_SCAN, 0, k_tab_down, dim3(gt), dim3(TB), b.table, b.table, b.state, NB, b.partial);
    JPK_LAUNCH(ctx, PROF_LG_SCATTER, 0, (k_lg_scatter<DB>), dim3(gp), dim3(TB), kin, vin, pin, kout, vout, pout, b.pieces, b.state, shift, b.table);
}

// JPK_PAIR_SHIFT: a round from the third on is a pair round (k_pair_*) when at least n >> shift suffixes are unresolved (default 6);
// negative = never (the comparator: plain prefix doubling)
int pair_rule_shift()
{
    static const int v = [] { const char *e = getenv("JPK_PAIR_SHIFT"); const int x = e ? atoi(e) : 6; return x > 31 ? 31 : x; }();
    return v;
}
// JPK_PAIR_FROM: the first round that may be a pair round (default 3; 2 is possible since every round is enqueued on exact counts, and
// measured worse: a 4 KiB period 17.2 -> 13.5 ms, but the silesia-like block 11.9 -> 16.6, long runs 18.9 -> 40.5 ms -- after one doubling
// round the groups of a repeat still mix everything that shares 30 symbols, and the rule's passes over the text are not free.  Round 1
// has to be a doubling round in any case: it is the one that spreads the run members)
int pair_rule_from()
{
    static const int v = [] { const char *e = getenv("JPK_PAIR_FROM"); const int x = e ? atoi(e) : 3; return x < 2 ? 2 : x; }();
    return v;
}
// JPK_PAIR_EARLY=0: no pair round at round 2 for lists that round 1 left as they were (comparator)
bool pair_rule_early()
{
    static const bool v = [] { const char *e = getenv("JPK_PAIR_EARLY"); return e ? atoi(e) != 0 : true; }();
    return v;
}
// JPK_PAIR_MIN: ... and at least this many (default 4096; the tests lower it so that tiny inputs take the path);
// JPK_PAIR_GAP: rounds from one pair round to the next (default 3 = two doubling rounds in between, at least 2)
uint32_t pair_rule_min()
{
    static const uint32_t v = [] { const char *e = getenv("JPK_PAIR_MIN"); const long x = e ? atol(e) : 4096L; return (uint32_t)(x < 2 ? 2 : x); }();
    return v;
}
// JPK_PAIR_RATIO: ... and the previous round left at least this percentage of ITS list unresolved -- 0 = whatever the previous round did.
// Default 90 since the end of round 6 (60 before): what the rule is for -- exact repeats, periodic data, a block that holds a file twice --
// keeps 99-100 % of its list through a doubling round (tools/pair_yield.py) and a pair round then resolves 84-100 % of it; REAL trees of
// near-duplicate files (64 MiB of this image's Python and ROCm sources: 16 rounds, each leaving 60-85 %) crossed the old threshold three
// times, every pair round there left 74-90 % of its list and cost 4 ms (k_pair_repair's walks): 35.0 against 22.9 ms per block without them.
uint32_t pair_rule_ratio()
{
    static const uint32_t v = [] { const char *e = getenv("JPK_PAIR_RATIO"); const int x = e ? atoi(e) : 90; return (uint32_t)(x < 0 ? 0 : (x > 100 ? 100 : x)); }();
    return v;
}
// JPK_PAIR_BUDGET: positions all walks of k_pair_repair together may visit in one pair round, in eighths of n (default 1; 8 n until the end of
// round 6: no block whose pair rounds pay notices the difference, a pair round that does not pay costs 7 ms less on 64 MiB of real binaries)
uint32_t pair_rule_budget(uint32_t n)
{
    static const uint32_t e8 = [] { const char *e = getenv("JPK_PAIR_BUDGET"); const long x = e ? atol(e) : 1; return (uint32_t)(x < 0 ? 0 : (x > 64 ? 64 : x)); }();
    const uint64_t b = (uint64_t)n * e8 / 8u;
    return b > 0xF0000000ull ? 0xF0000000u : (uint32_t)b;
}
// JPK_PAIR_KEEP: a pair round that leaves more than this percentage of its list did not pay (default 50; 100 = every one pays): the next one
// waits twice as long (two doubling rounds, then six, fourteen, thirty).  The Fibonacci word took nine pair rounds that resolved NOTHING, every
// one a round in which the doubling distance stands still (105 -> 60 ms per 32 MiB); a block that holds a real tree TWICE -- near-duplicate files
// inside an exact copy -- takes pair rounds that leave 93-100 % until doubling has dissolved the inner repeats, and then one that leaves nothing
// (round 12-15 of 23): giving up after the first would cost such a block its best round (tools/pair_yield.py, profiles/r06_real_files_pair_rounds.txt).
uint32_t pair_rule_keep()
{
    static const uint32_t v = [] { const char *e = getenv("JPK_PAIR_KEEP"); const int x = e ? atoi(e) : 50; return (uint32_t)(x < 0 ? 0 : (x > 100 ? 100 : x)); }();
    return v;
}
// JPK_PAIR_ITERS: passes of k_pair_fill per pair round (default 1; 2..4: later passes decide an end pair inside one group by the chain
// of its neighbouring pairs -- built for groups that mix two repeats, where k_pair_repair turned out to be what helps; kept as an option)
int pair_rule_iters()
{
    static const int v = [] { const char *e = getenv("JPK_PAIR_ITERS"); const int x = e ? atoi(e) : 1; return x < 1 ? 1 : (x > 4 ? 4 : x); }();
    return v;
}
// JPK_PAIR_REPAIR=0: stretches end at every position whose own neighbour is nearer (the comparator of k_pair_repair)
bool pair_rule_repair()
{
    static const bool v = [] { const char *e = getenv("JPK_PAIR_REPAIR"); return e ? atoi(e) != 0 : true; }();
    return v;
}
int pair_rule_gap()
{
    static const int v = [] { const char *e = getenv("JPK_PAIR_GAP"); const int x = e ? atoi(e) : 3; return x < 2 ? 2 : x; }();
    return v;
}

// When a round is a pair round: the host's rule as a function of what it knows -- the list every round started with -- so that the CPU suite can
// run it on recorded lists (jpk_debug_pair_schedule, tests/test_abi_and_host.py).  `step` is called once per round >= 1 whose list size is known
// before it is enqueued, in order.
struct PairSchedule {
    int last_pair = -8;
    int gap = pair_rule_gap();          // rounds from the last pair round to the next: doubles (+ 1) behind one that did not pay (pair_rule_keep)
    bool prev_pair = false;
    uint32_t m_prev;                    // the list the previous round started with
    explicit PairSchedule(uint32_t n) : m_prev(n) {}
    bool step(int round, uint32_t m_now, uint32_t n, bool runs_heavy, bool exact_from_round_1)
    {
        if (prev_pair) gap = ((uint64_t)m_now * 100u > (uint64_t)m_prev * pair_rule_keep()) ? 2 * gap + 1 : pair_rule_gap();
        const bool sizeable = pair_rule_shift() >= 0 && m_now >= pair_rule_min() && m_now >= (uint32_t)((uint64_t)n >> pair_rule_shift());
        bool pair = sizeable && round >= pair_rule_from() && round - last_pair >= gap && (uint64_t)m_now * 100u >= (uint64_t)m_prev * pair_rule_ratio();
        // ... and round 2 already when round 1 resolved next to nothing (99 % of its list is still there: periodic data, a block
        // that holds everything twice -- doubling is futile) unless the block is mostly runs, whose groups the run rule is splitting
        if (!pair && round == 2 && exact_from_round_1 && pair_rule_early() && sizeable && !runs_heavy && (uint64_t)m_now * 100u >= (uint64_t)m_prev * 99u) pair = true;
        m_prev = m_now;
        prev_pair = pair;
        if (pair) last_pair = round;
        return pair;
    }
};

// builds the BWT-in-SA-order bytes (b.bwt), the complete inverse suffix array (b.ISA) and, if b.SA is set, the suffix array
int build_sa(jpk_ctx *ctx, const uint8_t *T, uint32_t n, SaBufs &b)
{
    hipStream_t st = ctx->stream;
    ctx->stats.sa_rounds = 0;
    ctx->stats.sa_sorted_elems = 0;
    ctx->stats.sa_pair_rounds = 0;
    memset(ctx->stats.sa_round_active, 0, sizeof ctx->stats.sa_round_active);
    memset(ctx->stats.sa_round_large, 0, sizeof ctx->stats.sa_round_large);
    // One workgroup per tile / window / piece of the host's (one round old) upper bound; the surplus workgroups of a shrunken
    // list read the true count and leave.  Not persistent on purpose: a workgroup that has issued its random stores exits and
    // its slot is refilled at once, whereas a grid-stride loop would wait for those stores at its next barrier (measured:
    // k_seg_round 9.5 ms persistent against 8.1 ms).  The loops inside the kernels only matter beyond 2^20 tiles.
    constexpr unsigned CAP = 1u << 20;
    constexpr unsigned CAP_SEG = 1u << 20;

    // round 0: sort by the first `depth` bytes, packed into 56 bits (7 passes; ties keep descending text position)
    JPK_HIP(hipMemsetAsync(b.state, 0, sizeof(SaState), st));
    uint64_t *ks = b.keysA;
    uint32_t *vs = b.valsA;
    JPK_LAUNCH(ctx, PROF_SA_PACK, n, k_sym_present, dim3(cap_grid(n, 16 * TB * 4, 768)), dim3(TB), T, n, b.state);      // (every workgroup ends with up to 256 atomics on the same counters: few, fat workgroups)
    const bool var = b.GD[0] != nullptr;            // (sa_layout: var_keys_eligible)
    const int order = var ? key_order() : 0;
    const bool o1 = order >= 1, o2 = order >= 2;
    if (o1) {       // sampled counts for the context codes: about a million pairs / triples (65 536 vectors of 16 bytes) whatever the block
        JPK_HIP(hipMemsetAsync(b.ctab, 0, sizeof(uint32_t) * 256 * (o2 ? 256 + JPK_O2_CLASSES : 256), st));
        const uint32_t nv = n / 16u, stride = (nv >> 16) ? (nv >> 16) : 1u;
        JPK_LAUNCH(ctx, PROF_SA_PACK, 0, k_pair_counts, dim3(16, 16), dim3(256), T, n, stride, b.ctab);
        if (o2) {
            JPK_LAUNCH(ctx, PROF_SCAN, 0, k_ctx_select, dim3(1), dim3(256), b.ctab, b.ctxmap, b.state);
            JPK_LAUNCH(ctx, PROF_SA_PACK, 0, k_triple_counts, dim3(16, 16), dim3(256), T, n, stride, b.ctxmap, b.ctab);
        }
    }
    JPK_LAUNCH(ctx, PROF_SCAN, 0, k_key_plan, dim3(1), dim3(256), b.state, key_force_bits(), var ? 1 : 0);
    if (var) {
        if (o1) JPK_LAUNCH(ctx, PROF_SCAN, 0, k_ctx_plan, dim3(o2 ? 256 + JPK_O2_CLASSES : 256), dim3(256), b.state, b.ctab);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_key_final, dim3(1), dim3(256), b.state, o1 ? b.ctab : (uint32_t *)nullptr, o2 ? b.ctxmap : (uint16_t *)nullptr,
                   var_tag_shift(n), order);
    }
    // Which code did k_key_final choose?  One 4-byte read back (round 6; the host waits in front of every round anyway): exactly one pack
    // kernel is launched.  Rounds 4-5 enqueued all four and let the device pick: the three that "leave at once" averaged 524 + 210 + 189 us
    // in the timed loop -- every workgroup of a full grid first has to get its LDS on a CU that the other blocks in flight are using --
    // and held the block's stream meanwhile (profiles/r05_kernel_stats_bench_loop.txt).
    uint32_t vmode_h = 0;
    if (var) {
        JPK_HIP(hipMemcpyAsync(&ctx->h_mail[120], &b.state->vmode, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        JPK_HIP(hipEventRecord(ctx->ev_sa[1], st));
        JPK_HIP(hipEventSynchronize(ctx->ev_sa[1]));
        vmode_h = ctx->h_mail[120];
        if (vmode_h > 3u) return JPK_E_DEVICE;
    }
    const unsigned g_pack = cap_grid(n, CT, CAP);
    switch (vmode_h) {
    case 0:
        JPK_LAUNCH(ctx, PROF_SA_PACK, n, k_pack_keys, dim3(g_pack), dim3(TB), T, n, b.state, b.keysA, b.blk, b.bend, jpk_radix_onesweep() ? (uint32_t *)nullptr : b.scratch, b.D0);
        break;
    case 1:
        JPK_LAUNCH(ctx, PROF_SA_PACK, n, (k_pack_keys_var<false>), dim3(g_pack), dim3(TB), T, n, b.state, b.keysA, b.blk, b.bend, b.D0, (const uint32_t *)nullptr);
        break;
    case 2:
        JPK_LAUNCH(ctx, PROF_SA_PACK, n, (k_pack_keys_var<true>), dim3(g_pack), dim3(TB), T, n, b.state, b.keysA, b.blk, b.bend, b.D0, (const uint32_t *)b.ctab);
        break;
    default:
        JPK_LAUNCH(ctx, PROF_SA_PACK, n, k_pack_keys_o2, dim3(g_pack), dim3(TB), T, n, b.state, b.keysA, b.blk, b.bend, b.D0, (const uint32_t *)b.ctab, (const uint16_t *)b.ctxmap);
        break;
    }
    // whole tiles for the one-pass radix sort (the pack kernels have filled the pad slots): no pass needs a second, one-workgroup launch
    // for a partial last tile (seven per block, 200-360 us each in the timed loop)
    const uint32_t n_sort = jpk_radix_onesweep() ? (uint32_t)(((size_t)n + CT - 1) / CT * CT) : n;
    const bool tagless = var && var_tag_shift(n) >= 32;        // blocks above 2^28 bytes: nothing rides in the value (r0_short)
    JPK_TRY(jpk_radix_sort_slot_keys(ctx, n_sort, b.keysA, b.valsA, b.keysB, b.valsB, b.scratch, &ks, &vs, b.blk != nullptr, tagless ? (const uint8_t *)nullptr : b.D0, tagless ? 26 : var_tag_shift(n), n));
    const uint8_t *Dx = tagless ? b.D0 : (const uint8_t *)nullptr;
    ctx->stats.sa_sorted_elems += n;
    // The sorted pairs sit in (ks, vs).  The other pair of radix buffers is free from here on, the pair that holds the result
    // once k_r0_finish has read it: the doubling rounds live in them.
    uint64_t *kfree = (ks == b.keysA) ? b.keysB : b.keysA;
    uint32_t *vfree = (vs == b.valsA) ? b.valsB : b.valsA;
    b.b_sa = vfree;
    b.b_grp = reinterpret_cast<uint32_t *>(kfree);
    b.k2 = reinterpret_cast<uint32_t *>(kfree) + n;
    b.k2alt = reinterpret_cast<uint32_t *>(ks);
    b.sa_alt = reinterpret_cast<uint32_t *>(ks) + n;

    const unsigned g_ct = cap_grid(n, CT, CAP);
    if (r0_lookback()) {
        // one pass over the sorted pairs: the radix sort's scratch (free from here on) holds one status word per tile and the ticket
        uint64_t *lb_status = reinterpret_cast<uint64_t *>(b.scratch);
        const size_t ntiles0 = ((size_t)n + CT - 1) / CT;
        uint32_t *lb_ticket = reinterpret_cast<uint32_t *>(lb_status + ntiles0 + 1);
        JPK_HIP(hipMemsetAsync(lb_status, 0, sizeof(uint64_t) * (ntiles0 + 2), st));
        JPK_LAUNCH(ctx, PROF_SA_RERANK, n, k_r0_finish, dim3(g_ct), dim3(TB), ks, vs, n, b.tA, b.tB, b.ISA, b.bwt, b.SA, b.a_sa, b.a_grp, b.a_prev, b.state, b.bend, b.GD[0],
                   lb_status, lb_ticket, Dx);
    } else {
        JPK_LAUNCH(ctx, PROF_SA_RERANK, n, k_r0_count, dim3(g_ct), dim3(TB), ks, vs, n, b.tA, b.tB, b.bend, b.state, Dx);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_r0_scan, dim3(1), dim3(WG1), b.tA, b.tB, n, b.state);
        JPK_LAUNCH(ctx, PROF_SA_RERANK, n, k_r0_finish, dim3(g_ct), dim3(TB), ks, vs, n, b.tA, b.tB, b.ISA, b.bwt, b.SA, b.a_sa, b.a_grp, b.a_prev, b.state, b.bend, b.GD[0],
                   (uint64_t *)nullptr, (uint32_t *)nullptr, Dx);
    }
    ctx->stats.sa_rounds = 1;
    uint32_t *h_m = ctx->h_mail + 16;              // pinned: h_m[8 * (r & 1) ..] receives {m[0], m[1], npieces, lc, nrun} as round r leaves them
    static_assert(offsetof(SaState, nrun) == 16, "the rounds' copy takes m[2], npieces, lc, nrun in one piece");
    JPK_HIP(hipMemcpyAsync(&h_m[0], &b.state->m[0], 20, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipEventRecord(ctx->ev_sa[0], st));
    // JPK_SA_WAIT_ROUND: the first round that is enqueued on exact counts (default 1 since the context codes: round 1 of text starts with
    // 27 M of 67 M suffixes, round 2 with 49 K -- 48 windows, no large group; enqueued blind they were 65 K / 26 K workgroups per kernel and, in
    // round 2, 23 launches for nothing.  The wait is a few microseconds in front of a round; 3 = round 4's rule)
    static const int wait_round = [] { const char *e = getenv("JPK_SA_WAIT_ROUND"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    // remaining run lengths, only if round 0 left members of runs of >= depth equal bytes behind.  The host knows (round 6: it waits for
    // round 0's counts here, where round 1 would wait a moment later): the three kernels -- 130-160 us each in the timed loop to find
    // nothing to do -- are launched only when there is (enqueued blind they return at once otherwise)
    bool runs_possible = true;
    if (wait_round <= 1) {
        JPK_HIP(hipEventSynchronize(ctx->ev_sa[0]));
        runs_possible = h_m[4] != 0u;
    }
    if (runs_possible) {
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_run_first, dim3(cap_grid(n, CT, 4096)), dim3(TB), T, n, b.state, b.tA, b.blk);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_run_scan, dim3(1), dim3(WG1), b.tA, n, b.state);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_run_fill, dim3(cap_grid(n, CT, 4096)), dim3(TB), T, n, b.state, b.tA, b.RL, b.blk);
    }

    const int kbits = jpk_bits_for(3u * n);        // key2 <= n (a rank + 1); round 1 spreads the keys of groups of run members up to 3n; group rank < n
    int lg_pass = 0;
    const int lg_db = lg_digit_bits(n, &lg_pass);
    // Rounds 1 and 2 (the long ones: milliseconds each) are enqueued without waiting: the host learns the number of unresolved
    // suffixes one round late and enqueues round r with the grid bound of round r-2's result while the GPU is busy with round r-1.
    // From round 3 on the rounds are short and mostly empty, and what costs is their ~45 dependent launches each (a launch that has
    // nothing to do still waits its turn behind the other blocks' kernels: 90-190 us apiece in the timed loop): the host waits for
    // the previous round's counts first -- a few microseconds while other blocks keep the GPU busy -- and then enqueues exactly what is
    // needed: nothing when no suffix is unresolved (round 4: the trailing empty round is gone), no large-group passes (23 launches) once
    // a round has had no group above 1024 (groups only split: the count of their members never grows).
    uint32_t bound = n;                            // upper bound of the active count of the round being enqueued
    bool large_possible = true;                    // a group above 1024 members may still exist
    // The pair rule (k_pair_*): from round 3 on -- the host knows the exact count there -- a round whose list is still a sizeable share
    // of the block is a pair round instead of a doubling round; the doubling distance stays where it was.  Two doubling rounds lie
    // between two pair rounds (a group with dissenting pairs has to split before the rule can say more about it).
    int hshift = 0;                                // the next doubling round compares at distance depth << hshift
    int gd = 0;                                    // variable-length keys: GD[gd] holds the groups' depths, the next doubling round writes GD[gd ^ 1]
    PairSchedule sched(n);
    bool prev_pair = false;
    bool lg_heavy = false, runs_heavy = false;
    uint64_t pair_mask = 0;
    for (int round = 1;; round++) {
        const int par = round & 1;
        bool pair = false;
        if (round >= wait_round) {
            JPK_HIP(hipEventSynchronize(ctx->ev_sa[par ^ 1]));
            const uint32_t m_now = h_m[8 * (par ^ 1) + par];        // round r-1 wrote m[(r-1 & 1) ^ 1] = m[par]
            if (m_now == 0) break;                                    // nothing left: no empty round
            bound = m_now;
            if (round >= 2 && !prev_pair && h_m[8 * (par ^ 1) + 3] == 0) large_possible = false; // lc of round r-1 (a pair round does not count large groups)
            // many members of large groups ahead (round 1: run members, which round 0 counts; later: what the round before had): the full grid
            lg_heavy = round == 1 ? h_m[8 * (par ^ 1) + 4] > n / 64u : h_m[8 * (par ^ 1) + 3] > (1u << 22);
            if (round == 1) runs_heavy = lg_heavy;
            ctx->stats.sa_rounds = round + 1;
            pair = sched.step(round, m_now, n, runs_heavy, wait_round <= 1);
        }
        const unsigned g_win = cap_grid(bound, SEG_TILE, CAP);
        const unsigned g_seg = cap_grid(bound, SEG_TILE, CAP_SEG);
        const unsigned g_cmp = cap_grid(bound, CT, CAP);
        const size_t pc_bound = 2 * ((size_t)bound / SEG_TILE + 1);
        // (the large-group kernels walk the pieces with a grid stride.  The bound is two pieces per window of the list -- 131 K workgroups
        // in round 1, of which text uses a few hundred: 21 us per launch to start and retire the rest, 0.3 ms per block.  The grid is capped
        // at JPK_LG_GRID (default 8192; 0 = the bound) unless the block is known to be mostly large groups -- runs, all-zero: there a
        // workgroup per piece is worth 4-7 %)
        static const unsigned lg_cap = [] { const char *e = getenv("JPK_LG_GRID"); const long v = e ? atol(e) : 8192; return (unsigned)(v <= 0 ? CAP : v); }();
        const unsigned g_pc = cap_grid(pc_bound, 1, lg_heavy ? CAP : lg_cap);
        const unsigned g_tab = cap_grid(pc_bound << lg_db, SC_TILE, CAP);
        prev_pair = pair;
        if (pair) {
            // P lives in the sorted suffix numbers of round 0 (read for the last time by k_r0_finish), V | BAD | VL in the key2 buffer
            // (no gather in this round)
            uint32_t *P = vs;
            uint8_t *V = reinterpret_cast<uint8_t *>(b.k2), *BAD = V + n, *VL = V + 2 * (size_t)n;
            if (round < 64) pair_mask |= 1ull << round;
            JPK_HIP(hipMemsetAsync(P, 0, sizeof(uint32_t) * ((size_t)n + 1), st));
            JPK_HIP(hipMemsetAsync(BAD, 0, n, st));
            JPK_HIP(hipMemsetAsync(&b.state->pair_steps, 0, sizeof(uint32_t), st));
            JPK_LAUNCH(ctx, PROF_SA_KEYS, 0, k_pair_dist, dim3(g_win), dim3(TB), b.a_sa, b.a_grp, b.state, par, P, b.FH, b.LH);
            JPK_LAUNCH(ctx, PROF_SCAN, 0, k_win_scan1, dim3(1), dim3(WG1), b.FH, b.LH, b.PH, b.NH, b.state, par);
            if (pair_rule_repair()) JPK_LAUNCH(ctx, PROF_SA_KEYS, 0, k_pair_repair, dim3(cap_grid(n, TB * 4, 8192)), dim3(TB), P, n, b.ISA, b.blk, b.bend, b.state, pair_rule_budget(n));
            JPK_LAUNCH(ctx, PROF_SCAN, 0, k_pair_first, dim3(cap_grid(n, CT, 4096)), dim3(TB), P, n, b.tB);
            JPK_LAUNCH(ctx, PROF_SCAN, 0, k_pair_scan, dim3(1), dim3(WG1), b.tB, n);
            for (int it = 0; it < pair_rule_iters(); it++)     // later passes decide stretches that end in a group mixing two repeats (pair_verdict)
                JPK_LAUNCH(ctx, PROF_SA_KEYS, 0, k_pair_fill, dim3(cap_grid(n, CT, CAP)), dim3(TB), P, n, b.tB, b.ISA, V, b.blk, b.bend, it);
            JPK_LAUNCH(ctx, PROF_SA_KEYS, 0, k_pair_mark, dim3(g_win), dim3(TB), b.a_sa, b.a_grp, b.state, par, V, VL, BAD);
            JPK_LAUNCH(ctx, PROF_SA_RERANK, 0, k_pair_finish, dim3(g_win), dim3(TB), b.a_sa, b.a_grp, b.a_prev, b.state, par, b.PH, b.NH, VL, BAD, b.ISA, b.bwt, b.SA,
                       b.b_sa, b.b_grp, b.b_prev);
        } else {
        // (the groups' depths exist only when k_key_final chose a variable-length code: with the fixed-width code nothing wrote GD[0], and
        // every new group would cost two random reads of uninitialised memory and a random write nobody uses -- ADVICE r5)
        const uint32_t *gdr = vmode_h ? b.GD[gd] : nullptr;
        uint32_t *gdw = vmode_h ? b.GD[gd ^ 1] : nullptr;
        JPK_LAUNCH(ctx, PROF_SA_KEYS, 0, k_gather_win, dim3(g_win), dim3(TB), b.a_sa, b.a_grp, b.state, par, n, hshift, b.ISA, b.k2, b.FH, b.LH, T, b.RL, round == 1 ? 1 : 0,
                   b.a_prev, b.bend, gdr);
        hshift++;
        gd ^= 1;
        const unsigned g_wm = cap_grid((size_t)bound / SEG_TILE + 1, TB, 256);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_win_scan1, dim3(1), dim3(WG1), b.FH, b.LH, b.PH, b.NH, b.state, par);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_win_count, dim3(g_wm), dim3(TB), b.FH, b.LH, b.PH, b.NH, b.PC, b.state, par);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_win_scan2, dim3(1), dim3(WG1), b.PC, b.state, par, round);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_win_pieces, dim3(g_wm), dim3(TB), b.FH, b.LH, b.PH, b.NH, b.PC, b.pieces, b.state, par);
        JPK_LAUNCH(ctx, PROF_SA_SEG, 0, k_seg_round, dim3(g_seg), dim3(TB), b.a_sa, b.a_grp, b.k2, b.state, par, kbits, b.PH, b.a_prev, b.ISA, b.bwt, b.SA,
                   b.b_sa, b.b_grp, b.b_prev, gdr, gdw, round == 1 ? 1 : 0, n);
        if (large_possible) {   // large groups: lg_pass LSD passes over (key2, sa), ping-pong between (k2, a_sa) and (k2alt, sa_alt)
            uint32_t *kin = b.k2, *vin = b.a_sa, *kout = b.k2alt, *vout = b.sa_alt;
            uint8_t *pin = b.a_prev, *pout = b.p_alt;
            for (int p = 0; p < lg_pass; p++) {
                const int shift = p * lg_db;
                switch (lg_db) {
                case 4: launch_lg_pass<4>(ctx, b, kin, vin, pin, kout, vout, pout, shift, g_pc, g_tab); break;
                case 5: launch_lg_pass<5>(ctx, b, kin, vin, pin, kout, vout, pout, shift, g_pc, g_tab); break;
                case 6: launch_lg_pass<6>(ctx, b, kin, vin, pin, kout, vout, pout, shift, g_pc, g_tab); break;
                case 7: launch_lg_pass<7>(ctx, b, kin, vin, pin, kout, vout, pout, shift, g_pc, g_tab); break;
                default: launch_lg_pass<8>(ctx, b, kin, vin, pin, kout, vout, pout, shift, g_pc, g_tab); break;
                }
                uint32_t *tk = kin; kin = kout; kout = tk;
                uint32_t *tv = vin; vin = vout; vout = tv;
                uint8_t *tp = pin; pin = pout; pout = tp;
            }
            JPK_LAUNCH(ctx, PROF_SA_RERANK, 0, k_lg_heads, dim3(g_pc), dim3(TB), kin, b.pieces, b.state, b.pLast);
            JPK_LAUNCH(ctx, PROF_SCAN, 0, k_lg_scan, dim3(1), dim3(WG1), b.pLast, b.state);
            JPK_LAUNCH(ctx, PROF_SA_RERANK, 0, k_lg_finish, dim3(g_pc), dim3(TB), kin, vin, pin, b.a_grp, b.pieces, b.state, b.pLast, b.ISA, b.bwt, b.SA, b.b_sa,
                       b.b_grp, b.b_prev, gdr, gdw, round == 1 ? 1 : 0, n);
        }
        }
        JPK_LAUNCH(ctx, PROF_SA_RERANK, 0, k_cmp_count, dim3(g_cmp), dim3(TB), b.b_grp, b.state, par, b.tA);
        JPK_LAUNCH(ctx, PROF_SCAN, 0, k_cmp_scan, dim3(1), dim3(WG1), b.tA, b.state, par, round);
        JPK_LAUNCH(ctx, PROF_SA_RERANK, 0, k_cmp_scatter, dim3(g_cmp), dim3(TB), b.b_sa, b.b_grp, b.b_prev, b.state, par, b.tA, b.a_sa, b.a_grp, b.a_prev);
        JPK_HIP(hipGetLastError());
        JPK_HIP(hipMemcpyAsync(&h_m[8 * par], &b.state->m[0], 20, hipMemcpyDeviceToHost, st));
        JPK_HIP(hipEventRecord(ctx->ev_sa[par], st));
        if (round < wait_round) {
            // what the PREVIOUS round (or round 0) left behind: known without draining the queue
            JPK_HIP(hipEventSynchronize(ctx->ev_sa[par ^ 1]));
            const uint32_t m_start = h_m[8 * (par ^ 1) + par];      // = the active count this round started with (round r-1 wrote m[par])
            if (m_start == 0) break;                // this round was empty: done
            ctx->stats.sa_rounds = round + 1;
            bound = m_start;
            sched.m_prev = m_start;
        }
        if (round >= 2 * JPK_SA_MAX_ROUNDS) return JPK_E_DEVICE;     // cannot happen: the distance doubles, every suffix is unique once it is >= n
    }
    ctx->stats.sa_pair_rounds = (int64_t)pair_mask;
    // statistics: one small copy, read by sa_collect_stats() after the caller has synchronised the stream
    JPK_HIP(hipMemcpyAsync(ctx->h_mail + 32, b.state->round_m, sizeof(uint32_t) * (2 * JPK_SA_MAX_ROUNDS + 3), hipMemcpyDeviceToHost, st));   // + bits, depth, vmode
    ctx->sa_stats_pending = true;
    return JPK_OK;
}

}  // namespace

void jpk_sa_stats_sync(jpk_ctx *ctx)
{
    if (!ctx->sa_stats_pending) return;
    ctx->sa_stats_pending = false;
    const uint32_t *rm = ctx->h_mail + 32, *rl = ctx->h_mail + 32 + JPK_SA_MAX_ROUNDS;
    ctx->stats.sa_key_depth = (int32_t)ctx->h_mail[32 + 2 * JPK_SA_MAX_ROUNDS + 1];
    ctx->stats.sa_key_order = (int32_t)ctx->h_mail[32 + 2 * JPK_SA_MAX_ROUNDS + 2] - 1;      // vmode - 1: -1 = the fixed-width code
    for (int r = 0; r < JPK_SA_MAX_ROUNDS; r++) {
        const bool live = r < ctx->stats.sa_rounds;
        ctx->stats.sa_round_active[r] = live ? (int32_t)rm[r] : 0;
        ctx->stats.sa_round_large[r] = live ? (int32_t)rl[r] : 0;
        if (r >= 1 && live) ctx->stats.sa_sorted_elems += rm[r];
        if (r >= 1 && live && ctx->prof_on) {     // units of the round kernels are only known now
            ctx->prof_units[PROF_SA_KEYS] += rm[r];
            ctx->prof_units[PROF_SA_SEG] += rm[r] - rl[r];
            ctx->prof_units[PROF_SA_RERANK] += rm[r];
            ctx->prof_units[PROF_LG_HIST] += (uint64_t)rl[r] * 4;
            ctx->prof_units[PROF_LG_SCATTER] += (uint64_t)rl[r] * 4;
        }
    }
}

// arena bytes of one forward BWT of n sorted bytes (jpk_ctx_reserve)
size_t jpk_fwd_bwt_arena_bytes(uint32_t n)
{
    SaBufs b;
    jpk_ctx dummy;
    Arena plan(&dummy, true);
    sa_layout(plan, n ? n : 1, b, var_keys_eligible(n ? n : 1, false));
    return plan.need;
}

int jpk_suffix_array_device(jpk_ctx *ctx, const uint8_t *d_t, int32_t n, int32_t *d_sa)
{
    if (n <= 0) return JPK_OK;
    SaBufs b;
    Arena plan(ctx, true);
    sa_layout(plan, (size_t)n, b, var_keys_eligible((size_t)n, false));
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    sa_layout(real, (size_t)n, b, var_keys_eligible((size_t)n, false));
    b.SA = reinterpret_cast<uint32_t *>(d_sa);
    return build_sa(ctx, d_t, (uint32_t)n, b);
}

// ---- group sort: the forward BWTs of several (small) blocks as ONE suffix sort ------------------------------------------------
// Jampack's default block is 8 MiB and its smallest 1 MiB (format.hpp:20-22); a block costs ~200 dependent launches whatever its
// size, so small blocks are sorted together: their sorted parts are laid end to end as one text, every suffix stops at the end of
// its own block, and the block number is the sort's most significant digit -- block b's suffixes then occupy exactly the slots
// [start_b, start_b + nlen_b) of the common suffix array, ranks and positions are global, and each block's image and trailer come
// out of its own slice (bwt.cpp:44-61 per block).  In this mode the byte that rides with every active suffix is its block number
// (the end of its block is one table lookup away), so the BWT bytes are gathered from the text at the end: T[SA[i] - 1].
namespace {
struct GroupBlk {
    const uint8_t *src;        // the block as the caller gave it
    uint8_t *img;              // where its image goes (len + 480 bytes)
    uint32_t start, nlen, len; // slice of the common text / suffix array; bytes of the block
    uint32_t pad;
};
__global__ __launch_bounds__(TB) void k_group_concat(const GroupBlk *__restrict__ gb, uint8_t *__restrict__ C, uint8_t *__restrict__ blk)
{
    const GroupBlk g = gb[blockIdx.y];
    for (uint32_t base = blockIdx.x * TB * 16; base < g.nlen; base += gridDim.x * TB * 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t p = base + k * TB + threadIdx.x;
            if (p < g.nlen) { C[g.start + p] = g.src[p]; blk[g.start + p] = (uint8_t)blockIdx.y; }
        }
    }
}
__global__ __launch_bounds__(TB) void k_group_image(const GroupBlk *__restrict__ gb, const uint8_t *__restrict__ C, const uint32_t *__restrict__ SA,
                                                   const uint32_t *__restrict__ ISA)
{
    const GroupBlk g = gb[blockIdx.y];
    if (blockIdx.x == 0) {
        // raw tail (bwt.cpp:32-33) and, if anything was sorted, the 120 sampled ranks (bwt.cpp:58-61); a block shorter than 120
        // bytes leaves its trailer alone (bwt.cpp:35; the caller has zeroed the image)
        const uint32_t t = threadIdx.x;
        if (t < g.len - g.nlen) g.img[g.nlen + t] = g.src[g.nlen + t];
        if (g.nlen && t < JPK_BWT_UNITS) {
            const uint32_t v = ISA[g.start + (size_t)t * (g.nlen / JPK_BWT_UNITS)] - g.start + 1u;
            uint8_t *p = g.img + g.len + 4 * t;
            p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
        }
    }
    if (g.nlen == 0) return;
    const uint32_t idx = ISA[g.start] - g.start;                     // row of the block's suffix 0: dropped, the image starts with T[nlen-1]
    for (uint32_t i = blockIdx.x * TB + threadIdx.x; i < g.nlen; i += gridDim.x * TB) {
        if (i == 0) g.img[0] = C[g.start + g.nlen - 1];
        if (i != idx) g.img[(i < idx) ? i + 1 : i] = C[SA[g.start + i] - 1];
    }
}
}  // namespace

size_t jpk_fwd_bwt_group_arena_bytes(uint32_t total_nlen, int nblk)
{
    SaBufs b;
    jpk_ctx dummy;
    Arena plan(&dummy, true);
    sa_layout(plan, total_nlen ? total_nlen : 1, b, var_keys_eligible(total_nlen ? total_nlen : 1, true));
    plan.get<uint8_t>(total_nlen);            // common text
    plan.get<uint8_t>(total_nlen);            // block number per position
    plan.get<uint32_t>(total_nlen);           // suffix array (the BWT bytes are gathered through it)
    plan.get<uint32_t>((size_t)nblk + 1);
    plan.get<GroupBlk>((size_t)nblk);
    return plan.need;
}

// d_in[b] / len[b]: the blocks; d_img[b]: len[b] + 480 bytes each (zeroed by the caller when len[b] < 120).  At most 256 blocks, the
// sum of their sorted parts < 2^30.  Everything is enqueued on ctx->stream; the arena (at ctx->arena_base) must hold
// jpk_fwd_bwt_group_arena_bytes().
int jpk_fwd_bwt_group_device(jpk_ctx *ctx, int nblk, const uint8_t *const *d_in, const int32_t *len, uint8_t *const *d_img)
{
    if (nblk <= 0) return JPK_OK;
    if (nblk > 256) return JPK_E_ARG;
    // the two small tables are staged in the context's pinned page (16 KB: 256 x 32 B + 257 x 4 B), which outlives this call -- the
    // copies below are asynchronous; a context runs one call at a time and every caller synchronises before the next
    static_assert(sizeof(GroupBlk) * 256 + 4 * 257 <= 4096 * 4, "tables fit the pinned page");
    GroupBlk *gb = reinterpret_cast<GroupBlk *>(ctx->h_map);
    uint32_t *bend = ctx->h_map + sizeof(GroupBlk) * 256 / 4;
    uint64_t total = 0;
    uint32_t maxlen = 1;
    for (int b = 0; b < nblk; b++) {
        const uint32_t l = (uint32_t)len[b], nl = l - l % JPK_BWT_UNITS;
        gb[b] = GroupBlk{d_in[b], d_img[b], (uint32_t)total, nl, l, 0u};
        total += nl;
        bend[b] = (uint32_t)total;
        if (l > maxlen) maxlen = l;
    }
    if (total >= (1ull << 30)) return JPK_E_ARG;
    const uint32_t N = (uint32_t)total;
    const JpkCompressInflight inflight(ctx->device);
    SaBufs sb;
    Arena real(ctx, false);
    sa_layout(real, N ? N : 1, sb, var_keys_eligible(N ? N : 1, true));
    uint8_t *C = real.get<uint8_t>(N);
    uint8_t *blk = real.get<uint8_t>(N);
    uint32_t *SA = real.get<uint32_t>(N);
    uint32_t *d_bend = real.get<uint32_t>((size_t)nblk + 1);
    GroupBlk *d_gb = real.get<GroupBlk>((size_t)nblk);
    if (ctx->arena_off > ctx->arena_cap) return JPK_E_ALLOC;
    JPK_HIP(hipMemcpyAsync(d_bend, bend, sizeof(uint32_t) * (size_t)nblk, hipMemcpyHostToDevice, ctx->stream));
    JPK_HIP(hipMemcpyAsync(d_gb, gb, sizeof(GroupBlk) * (size_t)nblk, hipMemcpyHostToDevice, ctx->stream));
    const unsigned gx = cap_grid(maxlen, TB * 16, 1024);
    if (N) {
        hipLaunchKernelGGL(k_group_concat, dim3(gx, nblk), dim3(TB), 0, ctx->stream, d_gb, C, blk);
        sb.SA = SA;
        sb.blk = blk;
        sb.bend = d_bend;
        JPK_TRY(build_sa(ctx, C, N, sb));
    }
    JPK_LAUNCH(ctx, PROF_BWT_GATHER, N, k_group_image, dim3(cap_grid(maxlen, TB * 4, 4096), nblk), dim3(TB), d_gb, C, SA, sb.ISA);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

int jpk_fwd_bwt_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out)
{
    // ranks keep bit 30 and bit 31 for flags (above): a block of 2^30 bytes or more is refused, not sorted wrongly.  The format's largest
    // block (JPK_MAX_BLOCKSIZE = 1000 MiB, format.hpp:22) is below that
    if (len < 0 || (uint32_t)len >= JPK_FWD_BWT_LIMIT) return JPK_E_ARG;
    const JpkCompressInflight inflight(ctx->device);      // counted while this block is in its suffix sort
    const int32_t rem = len % JPK_BWT_UNITS, nlen = len - rem;
    if (nlen <= 0) {
        // bwt.cpp:29-35: only the raw tail is produced; the 480 trailer bytes are left untouched
        if (rem > 0) JPK_HIP(hipMemcpyAsync(d_out, d_in, (size_t)rem, hipMemcpyDeviceToDevice, ctx->stream));
        return JPK_OK;
    }
    SaBufs b;
    Arena plan(ctx, true);
    sa_layout(plan, (size_t)nlen, b, var_keys_eligible((size_t)nlen, false));
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    sa_layout(real, (size_t)nlen, b, var_keys_eligible((size_t)nlen, false));
    // heavy-phase gate: a caller that goes on to the entropy stage (jpk_dev_block_compress) already holds it and releases it
    // there; a stand-alone forward BWT holds it for the sort only
    const bool outer = ctx->gate_held;
    JPK_TRY(jpk_gate_enter(ctx));
    int rc = build_sa(ctx, d_in, (uint32_t)nlen, b);
    if (rc == JPK_OK) rc = jpk_gate_mark(ctx, ctx->stream);
    if (!outer) jpk_gate_leave(ctx);
    JPK_TRY(rc);
    JPK_LAUNCH(ctx, PROF_BWT_GATHER, nlen, k_bwt_image, dim3(cap_grid((size_t)nlen, TB * 16, 4096)), dim3(TB), d_in, b.bwt, b.ISA, (uint32_t)nlen, d_out);
    hipLaunchKernelGGL(k_bwt_trailer, dim3(1), dim3(128), 0, ctx->stream, d_in, b.ISA, (uint32_t)nlen, (uint32_t)len, d_out);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

// host-logic probe (include/jampack_abi.h): the pair-round schedule on recorded lists
extern "C" JPK_API int jpk_debug_pair_schedule(int64_t n, int32_t nrounds, const uint32_t *list, int32_t runs_heavy, int32_t *is_pair)
{
    if (n <= 0 || n >= (int64_t)JPK_FWD_BWT_LIMIT || nrounds < 1 || !list || !is_pair) return JPK_E_ARG;
    PairSchedule sched((uint32_t)n);
    int count = 0;
    is_pair[0] = 0;
    for (int r = 1; r < nrounds; r++) {
        is_pair[r] = list[r] ? (sched.step(r, list[r], (uint32_t)n, runs_heavy != 0, true) ? 1 : 0) : 0;
        count += is_pair[r];
    }
    return count;
}
