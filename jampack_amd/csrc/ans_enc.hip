// ans_enc.hip -- Ans::Encode (ans.cpp:113-234) on gfx950: per 1 MiB chunk
//   sorted-rank coding (rank.cpp:45-90) -> RLE0 (rle.cpp:22-47) -> exponent/mantissa models (model.cpp) ->
//   4-way interleaved byte rANS (rans_byte.hpp:62-110) -> LEB128 header (ans.cpp:272-285).
//
// Parallel restructuring (bytes identical to the reference):
//   rank coding   MTF rank of byte i = #symbols whose last occurrence is later than the previous occurrence of
//                 T[i]; last-occurrence tables are carried across 4 KiB tiles by a per-symbol max-scan, so every
//                 tile is an independent wave-sequential pass over its run heads (lanes hold the 256 time stamps,
//                 v_cmp -> popcount); the bucket scatter uses per-tile per-symbol prefix counts.
//   RLE0          run heads found per tile, run lengths through a per-chunk "zeros that follow the tile" table,
//                 output offsets by scans.
//   models        QuasiModel tables only change at fixed per-class symbol counts -> built in parallel from
//                 per-interval histograms; the AdaptiveModel CDF entries are independent scalar recurrences.
//   rANS          pair j lives in state lane j mod 4 -> four independent sequential chains per chunk that record
//                 (bytes, count) per step; byte positions are a prefix sum; payload scattered afterwards.
#include <vector>

#include "ans_common.hpp"
#include "common.hpp"

using namespace jpk;

namespace {

constexpr int TB = 256;

struct EncDims {
    uint32_t len;         // total bytes
    uint32_t chunk;       // chunk size (ANS_CHUNK, or len for the stand-alone Postcoder entry)
    uint32_t nch;         // chunks
    uint32_t tpc;         // tiles per chunk
    uint32_t ncl;         // chunks of this launch group (<= nch)
    const uint32_t *cmap; // launch index -> chunk id (null: identity).  The densest chunks are launched first as their own
                          // group so that their rANS chains start while the other chunks are still in the parallel stages.
    // group encode (the images of several blocks, each starting at a chunk boundary of one staging buffer: jpk_ans_encode_group_device)
    const uint32_t *clen; // bytes of every chunk (null: one block, the formula below)
    const uint32_t *cblk; // block of every chunk
    uint8_t *const *bout; // output buffer of every block (device array)
    // compact symbol layout (round 4): the arrays indexed by RLE0 symbol (class bytes, ordinals, model outputs, step records, states)
    // give every chunk as many slots as it HAS symbols (rounded up to 256) instead of one per byte: sbase[c] = first slot of chunk c
    // (sbase[nch] = total), lbase[c] = first record of its four rANS lanes.  null: one slot per byte, chunk c at c * stride (probes).
    const uint32_t *sbase;
    const uint32_t *lbase;
};
__device__ __forceinline__ uint32_t chunk_of(const EncDims &d, uint32_t i) { return d.cmap ? d.cmap[i] : i; }
// slots of chunk c in the symbol-indexed arrays, its first slot, and the first record of its lanes (see EncDims::sbase)
__device__ __forceinline__ size_t sym_stride(const EncDims &d, uint32_t c, size_t rle_stride) { return d.sbase ? (size_t)(d.sbase[c + 1] - d.sbase[c]) : rle_stride; }
__device__ __forceinline__ size_t sym_base(const EncDims &d, uint32_t c, size_t rle_stride) { return d.sbase ? (size_t)d.sbase[c] : (size_t)c * rle_stride; }
__device__ __forceinline__ uint32_t chunk_len(const EncDims &d, uint32_t c)
{
    if (d.clen) return d.clen[c];
    uint64_t beg = (uint64_t)c * d.chunk;
    uint64_t left = d.len - beg;
    return left < d.chunk ? (uint32_t)left : d.chunk;
}

// ---------------------------------------------------------------------------------------------------------------
// rank coding
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_enc_hist(const uint8_t *__restrict__ in, EncDims d, uint32_t *__restrict__ tilecnt,
                                                int32_t *__restrict__ lastpos)
{
    const uint32_t c = chunk_of(d, blockIdx.y), t = blockIdx.x;
    const uint32_t clen = chunk_len(d, c), ts = t * ATILE;
    if (ts >= clen) return;
    __shared__ uint32_t h[TB / 64][256];
    __shared__ int32_t lp[TB / 64][256];
#pragma unroll
    for (int k = 0; k < TB / 64; k++) { h[k][threadIdx.x] = 0; lp[k][threadIdx.x] = -1; }
    __syncthreads();
    const uint8_t *src = in + (size_t)c * d.chunk;
    uint8_t sy[ATILE / TB];
#pragma unroll
    for (int it = 0; it < ATILE / TB; it++) {          // the tile's loads in flight together (clamped; masked below)
        const uint32_t i = ts + it * TB + threadIdx.x;
        sy[it] = src[i < clen ? i : clen - 1];
    }
    // The input is a BWT image: runs of equal bytes, so most lanes of a wave hit the same bin.  Counted by wave match (ballots; the
    // first lane of each set of equal bytes adds their number and records the position of the set's LAST lane -- positions grow
    // with the lane and with the iteration, so a plain store keeps the maximum) on per-wave tables, no LDS atomics.
    const int w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < ATILE / TB; it++) {
        const uint32_t i = ts + it * TB + threadIdx.x;
        const bool valid = i < clen;
        const uint32_t s = sy[it];
        const uint64_t m = match_any8(s, valid);
        if (valid && (m & lt) == 0ull) {
            h[w][s] += (uint32_t)__popcll(m);
            lp[w][s] = (int32_t)(i - (uint32_t)lane_id() + (63u - (uint32_t)__clzll((long long)m)));
        }
    }
    __syncthreads();
    size_t o = ((size_t)c * d.tpc + t) * 256 + threadIdx.x;
    uint32_t hs = 0;
    int32_t lm = -1;
#pragma unroll
    for (int k = 0; k < TB / 64; k++) { hs += h[k][threadIdx.x]; lm = lp[k][threadIdx.x] > lm ? lp[k][threadIdx.x] : lm; }
    tilecnt[o] = hs;
    lastpos[o] = lm;
}

// per chunk: Freq[], per-tile prefix counts, carried last-occurrence table, bucket starts in the order of
// GenerateSortedMap (rank.cpp:15-39: descending frequency, ties -> smaller symbol first)
__global__ __launch_bounds__(256) void k_enc_prep(EncDims d, uint32_t *__restrict__ tilecnt, int32_t *__restrict__ lastpos,
                                                 int32_t *__restrict__ freq, uint32_t *__restrict__ bstart)
{
    const uint32_t c = chunk_of(d, blockIdx.x), s = threadIdx.x;
    const uint32_t clen = chunk_len(d, c);
    const uint32_t nt = (clen + ATILE - 1) / ATILE;
    uint32_t f = 0;
    int32_t p = -1;
    for (uint32_t t = 0; t < nt; t++) {
        size_t o = ((size_t)c * d.tpc + t) * 256 + s;
        uint32_t n = tilecnt[o];
        tilecnt[o] = f;
        f += n;
        int32_t l = lastpos[o];
        lastpos[o] = p;
        p = l > p ? l : p;
    }
    __shared__ uint32_t sf[256];
    sf[s] = f;
    freq[(size_t)c * 256 + s] = (int32_t)f;
    __syncthreads();
    uint32_t b = 0;
    for (uint32_t k = 0; k < 256; k++) {
        uint32_t fk = sf[k];
        if (fk > f || (fk == f && k < s)) b += fk;
    }
    bstart[(size_t)c * 256 + s] = b;
}

// one wave per tile: wave-sequential MTF on the recency list itself + bucket scatter (rank.cpp:69-87).
// The list lives in 4 registers per lane: lane p of lst_j holds the symbol at list position 64 j + p (0xFFFF behind the symbols seen
// so far in the chunk: a symbol's first occurrence has rank = number of distinct symbols before it, i.e. it sits at the first free
// position).  A head's rank is its position -- one v_cmp + s_ff1 when it is among the first 64, which is nearly every head of a BWT
// image -- and the move to front is one DPP wave_shr:1 under a lane mask (positions below it move down by one, the symbol enters at
// lane 0).  Positions 64.. are searched, and the shift carried from register to register, only when the first compare misses.  The
// list at the start of the tile is rebuilt from the carried last-occurrence stamps: position = number of symbols with a later stamp.
// Only run HEADS are walked serially (a repeat has rank 0, leaves the list unchanged and lands right behind its head in the same
// bucket): a ballot marks them in each 64-byte group, the loop visits the set bits, and every lane then derives its own
// (destination, rank) from the record of the head at or before it.  All loop control is wave-uniform (scalar branches, no exec-mask
// loops): the wave index is read with readfirstlane.  A per-wave LDS table holds the bucket write positions (wave-uniform accesses).
__device__ __forceinline__ uint32_t mtf_shr1(uint32_t carry_in, uint32_t v)          // lane i <- lane i - 1, lane 0 <- carry_in
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry_in, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__global__ __launch_bounds__(TB) void k_enc_mtf(const uint8_t *__restrict__ in, EncDims d, const uint32_t *__restrict__ tilebase,
                                               const int32_t *__restrict__ prevlast, const uint32_t *__restrict__ bstart,
                                               uint8_t *__restrict__ ranks)
{
    struct WaveLds {                       // one per wave; neighbours in one struct so that pairs of accesses share an address register
        uint32_t pos[256];                 // next write position of every symbol's bucket
        uint32_t dst[64], rk[64];          // (bucket position, rank) of the heads of the group in flight
        uint32_t lst[256];                 // the list at the start of the tile, while it is built
    };
    __shared__ WaveLds s_w[TB / 64];
    const uint32_t c = chunk_of(d, blockIdx.y);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t t = blockIdx.x * (TB / 64) + (uint32_t)w;
    const uint32_t clen = chunk_len(d, c), ts = t * ATILE;
    if (ts >= clen) return;
    const int l = lane_id();
    const size_t o = ((size_t)c * d.tpc + t) * 256;
    const int32_t last0 = prevlast[o + l], last1 = prevlast[o + 64 + l], last2 = prevlast[o + 128 + l], last3 = prevlast[o + 192 + l];
    const uint32_t *bs = bstart + (size_t)c * 256;
    WaveLds &L = s_w[w];
    const uint8_t *src = in + (size_t)c * d.chunk;
    uint8_t *dst = ranks + (size_t)c * d.chunk;
    const uint32_t te = (ts + ATILE < clen) ? ts + ATILE : clen;
    uint32_t bn = ((uint32_t)l < te - ts) ? src[ts + l] : 0x100u;             // group in flight
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
        L.pos[qq * 64 + l] = bs[qq * 64 + l] + tilebase[o + qq * 64 + l];
        L.lst[qq * 64 + l] = 0xFFFFu;
    }
    // the list before the tile: every symbol seen so far goes to position #{symbols with a later stamp} (stamps are positions: distinct)
    uint32_t nseen = 0;
#define JPK_MTF_PLACE(LASTJ, J)                                                                                                   \
    {                                                                                                                             \
        uint64_t sm = __ballot(LASTJ >= 0);                                                                                       \
        nseen += (uint32_t)__popcll(sm);                                                                                          \
        while (sm) {                                                                                                              \
            const uint32_t ln = (uint32_t)__builtin_ctzll(sm);                                                                    \
            sm &= sm - 1;                                                                                                         \
            const int32_t own = __builtin_amdgcn_readlane(LASTJ, (int)ln);                                                        \
            const uint32_t at = (uint32_t)__popcll(__ballot(last0 > own)) + (uint32_t)__popcll(__ballot(last1 > own)) +           \
                                (uint32_t)__popcll(__ballot(last2 > own)) + (uint32_t)__popcll(__ballot(last3 > own));            \
            L.lst[at] = (uint32_t)(J) * 64u + ln;                              /* every lane stores the same value */              \
        }                                                                                                                         \
    }
    JPK_MTF_PLACE(last0, 0)
    JPK_MTF_PLACE(last1, 1)
    JPK_MTF_PLACE(last2, 2)
    JPK_MTF_PLACE(last3, 3)
#undef JPK_MTF_PLACE
    uint32_t lst0 = L.lst[l], lst1 = L.lst[64 + l], lst2 = L.lst[128 + l], lst3 = L.lst[192 + l];
    uint32_t prevc = 256;                                                     // no byte before the tile: its first byte is a head
    for (uint32_t i0 = ts; i0 < te; i0 += 64) {
        const uint32_t nvalid = (te - i0 < 64u) ? te - i0 : 64u;
        const uint32_t b = bn;
        if (i0 + 64 < te) bn = (i0 + 64 + (uint32_t)l < te) ? src[i0 + 64 + l] : 0x100u;      // next group, loaded under this one's walk
        // byte before mine (lane 0: the last byte of the previous group)
        uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp((int)prevc, (int)b, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        uint64_t rem = __ballot((uint32_t)l < nvalid && (b != pb || l == 0));     // lane 0 is always walked: a repeat gets rank 0 there
        const uint64_t heads = rem;
        // length of the run that starts at my lane (used by head lanes only): distance to the next head, or to the end of the group
        const uint64_t above = (heads >> 1) >> l;
        const uint32_t runlen = above ? (uint32_t)__builtin_ctzll(above) + 1u : nvalid - (uint32_t)l;
        while (rem) {
            const uint32_t k = (uint32_t)__builtin_ctzll(rem);
            rem &= rem - 1;
            const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)k);
            const uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)runlen, (int)k);
            const uint32_t dpos = L.pos[cc];
            const uint64_t m0 = __ballot(lst0 == cc);
            uint32_t rank;
            if (m0) {                                          // among the first 64: the common case
                rank = (uint32_t)__builtin_ctzll(m0);
            } else {                                           // search the rest; shift the registers behind the first here
                const uint64_t m1 = __ballot(lst1 == cc), m2 = __ballot(lst2 == cc), m3 = __ballot(lst3 == cc);
                if (m1) rank = 64u + (uint32_t)__builtin_ctzll(m1);
                else if (m2) rank = 128u + (uint32_t)__builtin_ctzll(m2);
                else if (m3) rank = 192u + (uint32_t)__builtin_ctzll(m3);
                else rank = nseen++;                           // first occurrence in the chunk: enters from the first free position
                const uint32_t j = rank >> 6, q = rank & 63u;
                const bool part = (uint32_t)l <= q;            // lanes that move in the register holding the position
                if (j >= 1) {
                    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)lst0, 63);
                    if (j == 1) {
                        const uint32_t sh = mtf_shr1(c1, lst1);
                        lst1 = part ? sh : lst1;
                    } else {
                        const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)lst1, 63);
                        lst1 = mtf_shr1(c1, lst1);
                        if (j == 2) {
                            const uint32_t sh = mtf_shr1(c2, lst2);
                            lst2 = part ? sh : lst2;
                        } else {
                            const uint32_t c3 = (uint32_t)__builtin_amdgcn_readlane((int)lst2, 63);
                            lst2 = mtf_shr1(c2, lst2);
                            const uint32_t sh = mtf_shr1(c3, lst3);
                            lst3 = part ? sh : lst3;
                        }
                    }
                }
            }
            {                                                  // first register: positions 0 .. min(rank, 63) move down, the symbol enters at 0
                const uint32_t sh = mtf_shr1(cc, lst0);
                lst0 = ((uint32_t)l <= rank) ? sh : lst0;
            }
            L.pos[cc] = dpos + run;
            L.dst[k] = dpos;
            L.rk[k] = rank;
        }
        prevc = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)(nvalid - 1u));
        if ((uint32_t)l < nvalid) {
            const uint32_t hl = 63u - (uint32_t)__builtin_clzll(heads & ((2ull << l) - 1ull));     // my head (bit 0 is always set)
            const uint32_t dp = L.dst[hl] + ((uint32_t)l - hl);
            dst[dp] = ((uint32_t)l == hl) ? (uint8_t)L.rk[hl] : (uint8_t)0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RLE0 (rle.cpp:22-47)
// ---------------------------------------------------------------------------------------------------------------
// leading zeros of every tile of the rank array
__global__ __launch_bounds__(TB) void k_rle_lz(const uint8_t *__restrict__ ranks, EncDims d, uint32_t *__restrict__ lz)
{
    const uint32_t c = chunk_of(d, blockIdx.y), t = blockIdx.x;
    const uint32_t clen = chunk_len(d, c), ts = t * ATILE;
    if (ts >= clen) return;
    const uint32_t tl = (clen - ts < (uint32_t)ATILE) ? clen - ts : (uint32_t)ATILE;
    const uint8_t *src = ranks + (size_t)c * d.chunk + ts;
    uint32_t first = 0xFFFFFFFFu;
    uint8_t rv[ATILE / TB];
#pragma unroll
    for (int k = 0; k < ATILE / TB; k++) {            // loads first (clamped), all in flight
        const uint32_t p = k * TB + threadIdx.x;
        rv[k] = src[p < tl ? p : tl - 1];
    }
#pragma unroll
    for (int k = ATILE / TB - 1; k >= 0; k--) {
        uint32_t p = k * TB + threadIdx.x;
        if (p < tl && rv[k] != 0) first = p;
    }
    __shared__ uint32_t sm[TB / 64 + 1];
    uint32_t tot;
    block_incl_scan<OpMin>(first, sm, &tot);
    if (threadIdx.x == 0) lz[(size_t)c * d.tpc + t] = tot < tl ? tot : tl;
}

// ext[t] = zeros that follow the end of tile t without interruption (within the chunk)
__global__ void k_rle_ext(EncDims d, const uint32_t *__restrict__ lz, uint32_t *__restrict__ ext)
{
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= d.ncl) return;
    const uint32_t c = chunk_of(d, ci);
    const uint32_t clen = chunk_len(d, c);
    const uint32_t nt = (clen + ATILE - 1) / ATILE;
    uint32_t e = 0;
    for (int t = (int)nt - 1; t >= 0; t--) {
        ext[(size_t)c * d.tpc + t] = e;
        uint32_t tl = (clen - t * ATILE < (uint32_t)ATILE) ? clen - t * ATILE : (uint32_t)ATILE;
        uint32_t z = lz[(size_t)c * d.tpc + t];
        e = (z == tl) ? z + e : z;
    }
}

// EMIT=false: symbols per tile -> tcount ; EMIT=true: write symbols at toff[tile]
template <bool EMIT>
__global__ __launch_bounds__(TB) void k_rle_tiles(const uint8_t *__restrict__ ranks, EncDims d, const uint32_t *__restrict__ ext,
                                                 uint32_t *__restrict__ tcount, const uint32_t *__restrict__ toff, uint16_t *__restrict__ rle,
                                                 size_t rle_stride)
{
    const uint32_t c = chunk_of(d, blockIdx.y), t = blockIdx.x;
    const uint32_t clen = chunk_len(d, c), ts = t * ATILE;
    if (ts >= clen) return;
    const uint32_t tl = (clen - ts < (uint32_t)ATILE) ? clen - ts : (uint32_t)ATILE;
    const uint8_t *src = ranks + (size_t)c * d.chunk + ts;
    const uint32_t p0 = threadIdx.x * 16;
    uint8_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = (p0 + k < tl) ? src[p0 + k] : (uint8_t)1;
    uint32_t first = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 15; k >= 0; k--)
        if (p0 + k < tl && v[k] != 0) first = p0 + k;
    // exclusive suffix-min over threads of `first`
    __shared__ uint32_t fz[TB];
    __shared__ uint32_t res[TB];
    __shared__ uint32_t sm[TB / 64 + 1];
    fz[threadIdx.x] = first;
    __syncthreads();
    uint32_t rv = fz[TB - 1 - threadIdx.x];
    uint32_t inc = block_incl_scan<OpMin>(rv, sm, nullptr);
    res[threadIdx.x] = inc;
    __syncthreads();
    uint32_t after = (threadIdx.x == TB - 1) ? 0xFFFFFFFFu : res[TB - 2 - threadIdx.x];
    if (after == 0xFFFFFFFFu) after = tl + ext[(size_t)c * d.tpc + t];
    // next non-zero position for every zero of my segment
    uint32_t nxt[16];
    uint32_t nn = after;
#pragma unroll
    for (int k = 15; k >= 0; k--) {
        nxt[k] = nn;
        if (p0 + k < tl && v[k] != 0) nn = p0 + k;
    }
    uint32_t prev;
    if (p0 == 0) prev = (ts == 0) ? 1u : src[-1];
    else prev = (p0 - 1 < tl) ? src[p0 - 1] : 1u;
    uint32_t cnt = 0;
    uint32_t pv = prev;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (p0 + k < tl) {
            if (v[k] != 0) cnt += 1;
            else if (pv != 0) {
                uint32_t L = nxt[k] - (p0 + k) + 1;
                cnt += 31 - __clz((int)L);
            }
            pv = v[k];
        }
    }
    uint32_t tot;
    uint32_t incl = block_incl_scan<OpSum>(cnt, sm, &tot);
    if (!EMIT) {
        if (threadIdx.x == 0) tcount[(size_t)c * d.tpc + t] = tot;
        return;
    }
    uint16_t *out = rle + (size_t)c * rle_stride + toff[(size_t)c * d.tpc + t] + (incl - cnt);
    pv = prev;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (p0 + k < tl) {
            if (v[k] != 0) *out++ = (uint16_t)(v[k] + 1);
            else if (pv != 0) {
                uint32_t L = nxt[k] - (p0 + k) + 1;
                for (int b = 30 - __clz((int)L); b >= 0; b--) *out++ = (uint16_t)((L >> b) & 1u);
            }
            pv = v[k];
        }
    }
}

// serial prefix over the tiles of each chunk (<= 256 tiles in the ANS path)
__global__ void k_tile_prefix(EncDims d, const uint32_t *__restrict__ tcount, uint32_t *__restrict__ toff, uint32_t *__restrict__ total,
                              uint32_t unit_from_len /*1: tiles from chunk_len, 0: tiles from total_in*/, const uint32_t *__restrict__ count_in)
{
    const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= d.ncl) return;
    const uint32_t c = chunk_of(d, ci);
    const uint32_t n = unit_from_len ? chunk_len(d, c) : count_in[c];
    const uint32_t nt = (n + ATILE - 1) / ATILE;
    uint32_t s = 0;
    for (uint32_t t = 0; t < nt; t++) {
        uint32_t v = tcount[(size_t)c * d.tpc + t];
        toff[(size_t)c * d.tpc + t] = s;
        s += v;
    }
    total[c] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// models (ans.cpp:152-187)
// ---------------------------------------------------------------------------------------------------------------
// class histogram per tile of RLE symbols
__global__ __launch_bounds__(TB) void k_cls_count(const uint16_t *__restrict__ rle, size_t rle_stride, EncDims d, const uint32_t *__restrict__ rlen,
                                                 uint32_t *__restrict__ clscnt)
{
    const uint32_t c = chunk_of(d, blockIdx.y), t = blockIdx.x;
    const uint32_t n = rlen[c], ts = t * ATILE;
    if (ts >= n) return;
    __shared__ uint32_t h[8];
    if (threadIdx.x < 8) h[threadIdx.x] = 0;
    __syncthreads();
    const uint16_t *src = rle + (size_t)c * rle_stride;
    uint32_t loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint16_t sv[ATILE / TB];
#pragma unroll
    for (int it = 0; it < ATILE / TB; it++) {          // loads first (clamped), all in flight
        const uint32_t i = ts + it * TB + threadIdx.x;
        sv[it] = src[i < n ? i : n - 1];
    }
#pragma unroll
    for (int it = 0; it < ATILE / TB; it++) {
        uint32_t i = ts + it * TB + threadIdx.x;
        if (i < n) {
            int e = sym_class(sv[it]);
#pragma unroll
            for (int k = 0; k < 8; k++) loc[k] += (e == k);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t s = wave_sum(loc[k]);
        if (lane_id() == 0 && s) atomicAdd(&h[k], s);
    }
    __syncthreads();
    if (threadIdx.x < 8) clscnt[((size_t)c * d.tpc + t) * 8 + threadIdx.x] = h[threadIdx.x];
}

__global__ void k_cls_prefix(EncDims d, const uint32_t *__restrict__ rlen, uint32_t *__restrict__ clscnt, uint32_t *__restrict__ clstotal)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = g & 7u;
    if ((g >> 3) >= d.ncl) return;
    const uint32_t c = chunk_of(d, g >> 3);
    const uint32_t nt = (rlen[c] + ATILE - 1) / ATILE;
    uint32_t s = 0;
    for (uint32_t t = 0; t < nt; t++) {
        size_t o = ((size_t)c * d.tpc + t) * 8 + k;
        uint32_t v = clscnt[o];
        clscnt[o] = s;
        s += v;
    }
    clstotal[(size_t)c * 8 + k] = s;
}

// ordinal of every symbol inside its class (stable) + per-interval mantissa histograms of the quasi classes
__global__ __launch_bounds__(TB) void k_cls_ord(const uint16_t *__restrict__ rle, size_t rle_stride, EncDims d, const uint32_t *__restrict__ rlen,
                                               const uint32_t *__restrict__ clsbase, uint32_t *__restrict__ ord, uint32_t *__restrict__ qhist,
                                               uint8_t *__restrict__ cls8, uint32_t *__restrict__ clist)
{
    const uint32_t c = chunk_of(d, blockIdx.y), t = blockIdx.x;
    const uint32_t n = rlen[c], ts = t * ATILE;
    if (ts >= n) return;
    constexpr int W = TB / 64, IT = ATILE / TB;
    __shared__ uint32_t cnt[W][8];
    if (threadIdx.x < W * 8) (&cnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint16_t *src = rle + (size_t)c * rle_stride;
    const size_t sb = sym_base(d, c, rle_stride), cs = sym_stride(d, c, rle_stride);
    const uint64_t lt = lanemask_lt();
    uint32_t sy[IT], rk[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {                  // the tile's loads in flight together (clamped; masked below)
        const uint32_t i = ts + w * (64 * IT) + it * 64 + l;
        sy[it] = src[i < n ? i : n - 1];
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        uint32_t i = ts + w * (64 * IT) + it * 64 + l;
        bool valid = i < n;
        uint32_t s = valid ? sy[it] : 0;
        sy[it] = s;
        uint32_t e = (uint32_t)sym_class(s);
        const uint64_t m = match_any<3>(e, valid);
        uint32_t below = (uint32_t)__popcll(m & lt);
        uint32_t cc = valid ? cnt[w][e] : 0;
        rk[it] = cc + below;
        if (valid && below == 0) cnt[w][e] = cc + (uint32_t)__popcll(m);
    }
    __syncthreads();
    // mantissa histograms: the interval in which the tile starts (per class) is accumulated in LDS and flushed once;
    // the few symbols that fall into later intervals go straight to global atomics
    __shared__ uint32_t lh[6][QSTRIDE];
    __shared__ int q0[8];
    for (int i = threadIdx.x; i < 6 * QSTRIDE; i += TB) (&lh[0][0])[i] = 0;
    if (threadIdx.x < 8) {
        uint32_t s = clsbase[((size_t)c * d.tpc + t) * 8 + threadIdx.x];
        q0[threadIdx.x] = qinterval(s);
#pragma unroll
        for (int k = 0; k < W; k++) { uint32_t v = cnt[k][threadIdx.x]; cnt[k][threadIdx.x] = s; s += v; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; it++) {
        uint32_t i = ts + w * (64 * IT) + it * 64 + l;
        if (i < n) {
            uint32_t s = sy[it];
            int e = sym_class(s);
            uint32_t k = cnt[w][e] + rk[it];
            ord[sb + i] = k;
            cls8[sb + i] = (uint8_t)(e | ((s & 1u) << 3));      // class | mantissa bit of classes 0/1
            if (e < 2) clist[2 * sb + (size_t)e * ((cs + 3) & ~(size_t)3) + k] = i | ((s & 1u) << 31);   // compact list of the class
            else {
                const int q = qinterval(k);
                const uint32_t m = s - (uint32_t)class_base(e);
                if (q == q0[e]) atomicAdd(&lh[e - 2][m], 1u);
                else atomicAdd(&qhist[(((size_t)c * 6 + (e - 2)) * NQ + q) * QSTRIDE + m], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * QSTRIDE; i += TB) {
        const uint32_t v = (&lh[0][0])[i];
        if (v) {
            const int e2 = i / QSTRIDE, m = i % QSTRIDE;
            atomicAdd(&qhist[(((size_t)c * 6 + e2) * NQ + q0[e2 + 2]) * QSTRIDE + m], v);
        }
    }
}

// one wave per (interval, class, chunk): CDF of interval r from the histogram of interval r-1 (model.cpp:160-204)
__global__ __launch_bounds__(64) void k_quasi_build(EncDims d, const uint32_t *__restrict__ clstotal, const uint32_t *__restrict__ qhist,
                                                   uint32_t *__restrict__ qcdf)
{
    const int r = blockIdx.x, e = blockIdx.y + 2;
    const uint32_t c = chunk_of(d, blockIdx.z);
    const int A = class_alpha(e), l = lane_id();
    uint32_t *cdf = qcdf + (((size_t)c * 6 + (e - 2)) * NQ + r) * QSTRIDE;
    if (r == 0) {
        for (int i = l; i <= A; i += 64) cdf[i] = uniform_cdf(A, i);
        return;
    }
    if (clstotal[(size_t)c * 8 + e] < qbound(r)) return;   // this table is never reached
    const uint32_t *h = qhist + (((size_t)c * 6 + (e - 2)) * NQ + (r - 1)) * QSTRIDE;
    uint32_t F[3];
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int i = l + 64 * k;
        F[k] = (i < A) ? 16u * h[i] : 0u;
        tot += F[k];
    }
    tot = wave_sum(tot);
    int lg = 0;
    while ((tot >> lg) + (uint32_t)A > 65536u) lg++;
    uint32_t t2 = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int i = l + 64 * k;
        F[k] = (i < A) ? (F[k] >> lg) + 1u : 0u;
        t2 += F[k];
    }
    t2 = wave_sum(t2);
    uint32_t t3 = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        F[k] = (65536u * F[k]) / t2;      // unsigned 32-bit, as model.cpp:183
        t3 += F[k];
    }
    t3 = wave_sum(t3);
    if (l == 0) F[0] += 65536u - t3;
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int i = l + 64 * k;
        uint32_t inc = wave_incl_sum(F[k]);
        if (i < A) cdf[i] = carry + inc - F[k];
        carry += __shfl(inc, 63, 64);
    }
    if (l == 0) cdf[A] = 65536u;
}

// ---- AdaptiveModel recurrences, parallel in time ---------------------------------------------------------------
// Nine independent scalar recurrences per chunk: exponent model cdf[1..7] (alphabet 8) and cdf[1] of the two
// alphabet-2 mantissa models.  One step is x += (mix - x) >> 5 with mix in {i, i + 65536 - A} (model.cpp:60-77):
// a monotone map that shrinks any interval of states by >= floor(width/32) per update, so after >= ~500 updates
// the set of reachable states is an interval of width <= 31 whatever the history was.
// Item streams: the exponent entries see every symbol (class byte stream cls8); a mantissa model only sees the symbols
// of its class, which k_cls_ord has compacted into a list (position | mantissa bit << 31), so its "time" is the
// class ordinal and a warm-up is always the 1280 list entries in front of the segment.  Per 4096-item segment:
//   A    warm up the two extreme states over the preceding 1280 items -> [lo, hi].  lo == hi: the start state is
//        exact; run the segment, write the outputs, record the end state.  Otherwise record [lo, hi].
//   tab  unresolved segments: 32 lanes walk the segment from the 32 candidate start states -> transfer table.
//   B    one lane per (chunk, recurrence) walks the segments in order composing exact states through the tables.
//   C    the unresolved segments are run again from their now exact start state, writing the outputs.
// Every output is produced from an exact state: the result is bit-identical to the sequential reference.
// Warm-up items in front of a segment.  The interval of reachable states contracts by >= floor(width / 32) per item whatever the
// item is (floor is superadditive), so from the full range it is <= 31 wide after 259 items: any warm-up >= 320 keeps the guarantee
// the table of 32 candidates rests on.  Beyond that a longer warm-up only decides how many segments are already a single state
// when the segment starts (k_adapt_ext / k_adapt_tab take the rest).  Rounds 1-3 used 1280, which is best for a block ALONE (the
// five k_adapt_* kernels: 4.3 ms per 64 MiB block against 4.9 with 320) -- but the warm-up is pure vector work (two states per item,
// +62 % on k_adapt_a), and with blocks in flight vector issue is what is short: the bench line prefers 320 (4.41 / 4.38 / 4.29 /
// 4.15 GB/s for 320 / 384 / 512 / 1280 in one sequence, 4.22 / 4.13 / 4.11 for 320 / 384 / 1280 interleaved three times;
// profiles/r04_adapt_warm.txt).  JPK_AD_WARM overrides (multiple of 64, 320..4096).
constexpr uint32_t AD_WARM_DEFAULT = 320;

struct AdRec {                          // one recurrence
    bool exp;                           // exponent model entry (true) or alphabet-2 mantissa model (false)
    int i;                              // cdf index (exp) / 1
    int cls;                            // class of the mantissa model
    int A;
    __device__ __forceinline__ AdRec(uint32_t rec) : exp(rec < 7), i(rec < 7 ? (int)rec + 1 : 1), cls((int)rec - 7), A(rec < 7 ? 8 : 2) {}
    __device__ __forceinline__ int32_t init() const { return (int32_t)uniform_cdf(A, i); }
    __device__ __forceinline__ int32_t smin() const { return i; }
    __device__ __forceinline__ int32_t smax() const { return i + 65536 - A; }
};

// the item stream of a recurrence inside one chunk
__host__ __device__ __forceinline__ size_t clist_stride(size_t rle_stride) { return (rle_stride + 3) & ~(size_t)3; }   // 16-byte rows

struct AdStream {
    const uint8_t *c8;        // exponent entries: class byte of every symbol
    const uint32_t *list;     // mantissa models: compacted (position | bit << 31) of the class
    uint32_t n;               // items
};
__device__ __forceinline__ AdStream ad_stream(const EncDims &d, const AdRec &r, uint32_t c, const uint8_t *cls8, const uint32_t *clist, size_t rle_stride,
                                              const uint32_t *rlen, const uint32_t *clstotal)
{
    AdStream st;
    const size_t sb = sym_base(d, c, rle_stride), cs = sym_stride(d, c, rle_stride);
    st.c8 = cls8 + sb;
    st.list = r.exp ? nullptr : clist + 2 * sb + (size_t)r.cls * clist_stride(cs);
    st.n = r.exp ? rlen[c] : clstotal[(size_t)c * 8 + r.cls];
    return st;
}

// f(position, coded symbol) for the items [t0, t1) of a stream; 16-byte loads where aligned, the next load is
// issued (unconditionally, clamped) before the current group is consumed
template <class F>
__device__ __forceinline__ void ad_for_each(const AdRec &r, const AdStream &st, uint32_t t0, uint32_t t1, F f)
{
    uint32_t t = t0;
    if (r.exp) {
        const uint8_t *p = st.c8;
        while (t < t1 && (t & 15u)) { f(t, (int)(p[t] & 7u)); t++; }
        // Lanes walk segments 4 KiB apart: every load of a wave touches 64 different lines, and a lane that keeps ONE 16-byte load in
        // flight spends its time waiting for it (the walk ran at ~30 cycles per instruction).  Round 4: a whole 64-byte line per lane
        // and step, the next line requested before the current one is consumed (four loads in flight per lane).
        if (t + 64 <= t1) {
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4 *>(p + t + 16 * q);
            for (; t + 64 <= t1; t += 64) {
                const uint32_t tn = (t + 128 <= t1) ? t + 64 : t;
                uint4 nv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) nv[q] = *reinterpret_cast<const uint4 *>(p + tn + 16 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int k = 0; k < 16; k++) f(t + 16 * q + k, (int)((w[k >> 2] >> (8 * (k & 3))) & 7u));
                }
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = nv[q];
            }
        }
        if (t + 16 <= t1) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + t);
            for (; t + 16 <= t1; t += 16) {
                const uint32_t tn = (t + 32 <= t1) ? t + 16 : t;
                const uint4 nv = *reinterpret_cast<const uint4 *>(p + tn);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; k++) f(t + k, (int)((w[k >> 2] >> (8 * (k & 3))) & 7u));
                v = nv;
            }
        }
        while (t < t1) { f(t, (int)(p[t] & 7u)); t++; }
    } else {
        const uint32_t *p = st.list;
        while (t < t1 && (t & 3u)) { const uint32_t e = p[t]; f(e & 0x7FFFFFFFu, (int)(e >> 31)); t++; }
        if (t + 16 <= t1) {                                      // (a 64-byte line per step, the next one in flight: see above)
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4 *>(p + t + 4 * q);
            for (; t + 16 <= t1; t += 16) {
                const uint32_t tn = (t + 32 <= t1) ? t + 16 : t;
                uint4 nv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) nv[q] = *reinterpret_cast<const uint4 *>(p + tn + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f(v[q].x & 0x7FFFFFFFu, (int)(v[q].x >> 31));
                    f(v[q].y & 0x7FFFFFFFu, (int)(v[q].y >> 31));
                    f(v[q].z & 0x7FFFFFFFu, (int)(v[q].z >> 31));
                    f(v[q].w & 0x7FFFFFFFu, (int)(v[q].w >> 31));
                }
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = nv[q];
            }
        }
        if (t + 4 <= t1) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + t);
            for (; t + 4 <= t1; t += 4) {
                const uint32_t tn = (t + 8 <= t1) ? t + 4 : t;
                const uint4 nv = *reinterpret_cast<const uint4 *>(p + tn);
                f(v.x & 0x7FFFFFFFu, (int)(v.x >> 31));
                f(v.y & 0x7FFFFFFFu, (int)(v.y >> 31));
                f(v.z & 0x7FFFFFFFu, (int)(v.z >> 31));
                f(v.w & 0x7FFFFFFFu, (int)(v.w >> 31));
                v = nv;
            }
        }
        while (t < t1) { const uint32_t e = p[t]; f(e & 0x7FFFFFFFu, (int)(e >> 31)); t++; }
    }
}

// run one exact trajectory over the items [t0,t1) writing the model outputs (ans.cpp:159-176).
// Exponent entries: the entry's value BEFORE every symbol goes to its own history row hist[t] -- sixteen values per two 16-byte
// stores, no condition and no branch in the loop; k_pairs picks entry e (low) and entry e + 1 (high) of its symbol from the seven
// rows.  (Round 2 stored low / high per symbol from whichever lane matched the symbol's class: two compares, two exec-masked
// 2-byte stores and their branches per symbol and lane -- twelve of the eighteen instructions of the loop.)
__device__ __forceinline__ int32_t ad_run_write(const AdRec &r, const AdStream &st, uint32_t t0, uint32_t t1, int32_t x,
                                                uint16_t *__restrict__ hist, uint32_t *__restrict__ ma)
{
    if (r.exp) {
        const int i = r.i;
        const uint8_t *p = st.c8;
        uint32_t t = t0;
        while (t < t1 && (t & 15u)) { hist[t] = (uint16_t)x; x = adapt_step(x, i, (int)(p[t] & 7u), 8); t++; }
        if (t + 64 <= t1) {                                      // a 64-byte line of class bytes per step, the next one in flight (see ad_for_each)
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const uint4 *>(p + t + 16 * q);
            for (; t + 64 <= t1; t += 64) {
                const uint32_t tn = (t + 128 <= t1) ? t + 64 : t;
                uint4 nv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) nv[q] = *reinterpret_cast<const uint4 *>(p + tn + 16 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    uint32_t o[8];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        if (k & 1) o[k >> 1] |= (uint32_t)x << 16;
                        else o[k >> 1] = (uint32_t)x & 0xffffu;
                        x = adapt_step(x, i, (int)((w[k >> 2] >> (8 * (k & 3))) & 7u), 8);
                    }
                    uint4 *qq = reinterpret_cast<uint4 *>(hist + t + 16 * q);
                    qq[0] = make_uint4(o[0], o[1], o[2], o[3]);
                    qq[1] = make_uint4(o[4], o[5], o[6], o[7]);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = nv[q];
            }
        }
        if (t + 16 <= t1) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + t);
            for (; t + 16 <= t1; t += 16) {
                const uint32_t tn = (t + 32 <= t1) ? t + 16 : t;
                const uint4 nv = *reinterpret_cast<const uint4 *>(p + tn);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k & 1) o[k >> 1] |= (uint32_t)x << 16;
                    else o[k >> 1] = (uint32_t)x & 0xffffu;
                    x = adapt_step(x, i, (int)((w[k >> 2] >> (8 * (k & 3))) & 7u), 8);
                }
                uint4 *q = reinterpret_cast<uint4 *>(hist + t);
                q[0] = make_uint4(o[0], o[1], o[2], o[3]);
                q[1] = make_uint4(o[4], o[5], o[6], o[7]);
                v = nv;
            }
        }
        while (t < t1) { hist[t] = (uint16_t)x; x = adapt_step(x, i, (int)(p[t] & 7u), 8); t++; }
    } else {
        ad_for_each(r, st, t0, t1, [&](uint32_t t, int m) {
            const uint32_t l0 = m ? (uint32_t)x : 0u, fr = m ? 65536u - (uint32_t)x : (uint32_t)x;
            ma[t] = l0 | (fr << 16);
            x = adapt_step(x, 1, m, 2);
        });
    }
    return x;
}

struct AdArgs {
    const uint8_t *cls8; const uint32_t *clist; size_t rle_stride; const uint32_t *rlen; const uint32_t *clstotal;
    const uint32_t *clsbase;                // [chunk][tile][8]: symbols of every class in front of the tile (k_cls_prefix)
    uint16_t *exph; uint32_t *mantad;       // exph: [chunk][7 exponent entries][rle_stride] history of every entry
    uint32_t warm;                          // warm-up items in front of a segment (AD_WARM_DEFAULT)
    uint32_t *seg_flag; int32_t *seg_lo, *seg_end, *seg_start; uint16_t *seg_tab;
};

// lanes = 64 consecutive segments of ONE recurrence (grid y): the exponent / mantissa code paths and the item widths differ, a wave
// that mixed recurrences would walk both paths one after the other with half its lanes off
__global__ __launch_bounds__(64) void k_adapt_a(EncDims d, AdArgs a)
{
    const uint32_t c = chunk_of(d, blockIdx.z), rec = blockIdx.y;
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    const AdRec r(rec);
    const AdStream st = ad_stream(d, r, c, a.cls8, a.clist, a.rle_stride, a.rlen, a.clstotal);
    const uint32_t nt = (st.n + ATILE - 1) / ATILE;
    if (k >= nt) return;
    const uint32_t t0 = k * ATILE, t1 = (t0 + ATILE < st.n) ? t0 + ATILE : st.n;
    int32_t lo, hi;
    if (k == 0) lo = hi = r.init();
    else {
        lo = r.smin(); hi = r.smax();
        const int i = r.i, A = r.A;
        ad_for_each(r, st, t0 - a.warm, t0, [&](uint32_t, int sy) { lo = adapt_step(lo, i, sy, A); hi = adapt_step(hi, i, sy, A); });
    }
    const size_t so = ((size_t)c * 9 + rec) * d.tpc + k;
    if (lo == hi) {
        a.seg_flag[so] = 1u;
        a.seg_end[so] = ad_run_write(r, st, t0, t1, lo, a.exph + 7 * sym_base(d, c, a.rle_stride) + (size_t)(rec < 7 ? rec : 0) * sym_stride(d, c, a.rle_stride), a.mantad + sym_base(d, c, a.rle_stride));
        return;
    }
    // unresolved: k_adapt_tab tabulates the 32 candidate start states lo .. lo+31 (hi - lo <= 31 after the warm-up).
    // Except the one case that makes most of them: an exponent entry above every class that occurs (rare large classes).  Its mix is
    // smax at every step, hi never left smax (one other step would have dropped it for good: nothing climbs back to smax from
    // below), lo stalled at smax - 31, and all 32 states in between are fixed points of that step.  If no symbol of the segment
    // reaches the entry either, the segment's transfer is the identity and its history row a constant: no table, no second walk.
    uint32_t flag = 0u;
    if (r.exp && hi == r.smax() && lo == r.smax() - 31) {
        const uint32_t *cb = a.clsbase + ((size_t)c * d.tpc + k) * 8;
        uint32_t reach = 0;                                    // symbols of the segment with class >= i
        for (int e = r.i; e < 8; e++) reach += ((k + 1 < nt) ? cb[8 + e] : a.clstotal[(size_t)c * 8 + e]) - cb[e];
        if (reach == 0) flag = 2u;
    }
    a.seg_flag[so] = flag;
    a.seg_lo[so] = lo;
    a.seg_end[so] = hi;          // re-used as the interval's upper end until k_adapt_b has run
}

// unresolved segments, second chance: walk the two ends of the interval through the segment itself.  The step is monotone in the
// state, so if the ends meet every start state in between reaches that same end state: the segment's transfer is a constant and no
// table is needed (flag 3; k_adapt_c still re-runs it for the outputs once k_adapt_b knows its start).  That is the usual fate of
// an interval left open by a quiet warm-up (no symbol reached the entry for 1280 items, then one does inside the segment: the
// 32 stalled states drop together, contract by 31/32 per step on the climb back and merge within ~130 steps).
__global__ __launch_bounds__(64) void k_adapt_ext(EncDims d, AdArgs a)
{
    const uint32_t c = chunk_of(d, blockIdx.z), rec = blockIdx.y;
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    const AdRec r(rec);
    const AdStream st = ad_stream(d, r, c, a.cls8, a.clist, a.rle_stride, a.rlen, a.clstotal);
    const uint32_t nt = (st.n + ATILE - 1) / ATILE;
    if (k >= nt) return;
    const size_t so = ((size_t)c * 9 + rec) * d.tpc + k;
    if (a.seg_flag[so]) return;
    const uint32_t t0 = k * ATILE, t1 = (t0 + ATILE < st.n) ? t0 + ATILE : st.n;
    int32_t lo = a.seg_lo[so], hi = a.seg_end[so];
    const int i = r.i, A = r.A;
    ad_for_each(r, st, t0, t1, [&](uint32_t, int sy) { lo = adapt_step(lo, i, sy, A); hi = adapt_step(hi, i, sy, A); });
    if (lo == hi) {
        a.seg_flag[so] = 3u;
        a.seg_end[so] = lo;
    }
}

// transfer table of a segment that is still unresolved: 32 lanes = 32 candidate start states walk the segment together (the
// item loads are wave-uniform), instead of one lane walking it 32 times
__global__ __launch_bounds__(64) void k_adapt_tab(EncDims d, AdArgs a)
{
    const uint32_t c = chunk_of(d, blockIdx.z), rec = blockIdx.y;
    const uint32_t k = blockIdx.x * 2 + (threadIdx.x >> 5);
    const int q = threadIdx.x & 31;
    const AdRec r(rec);
    const AdStream st = ad_stream(d, r, c, a.cls8, a.clist, a.rle_stride, a.rlen, a.clstotal);
    const uint32_t nt = (st.n + ATILE - 1) / ATILE;
    if (k >= nt) return;
    const size_t so = ((size_t)c * 9 + rec) * d.tpc + k;
    if (a.seg_flag[so]) return;
    const uint32_t t0 = k * ATILE, t1 = (t0 + ATILE < st.n) ? t0 + ATILE : st.n;
    const int32_t lo = a.seg_lo[so], hi = a.seg_end[so];
    int32_t x = (lo + q < hi) ? lo + q : hi;
    const int i = r.i, A = r.A;
    ad_for_each(r, st, t0, t1, [&](uint32_t, int sy) { x = adapt_step(x, i, sy, A); });
    a.seg_tab[so * 32 + q] = (uint16_t)(x - 1);     // states are in [1, 65535]
}

__global__ __launch_bounds__(64) void k_adapt_b(EncDims d, AdArgs a)
{
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    const uint32_t rec = g & 15u;
    if ((g >> 4) >= d.ncl || rec >= 9) return;
    const uint32_t c = chunk_of(d, g >> 4);
    const AdRec r(rec);
    const AdStream st = ad_stream(d, r, c, a.cls8, a.clist, a.rle_stride, a.rlen, a.clstotal);
    const uint32_t nt = (st.n + ATILE - 1) / ATILE;
    int32_t x = r.init();
    for (uint32_t k = 0; k < nt; k++) {
        const size_t so = ((size_t)c * 9 + rec) * d.tpc + k;
        const uint32_t flag = a.seg_flag[so];
        if (flag == 1u) x = a.seg_end[so];
        else if (flag == 2u) a.seg_start[so] = x;             // identity segment: the state passes through
        else if (flag == 3u) { a.seg_start[so] = x; x = a.seg_end[so]; }      // constant transfer
        else {
            a.seg_start[so] = x;
            int q = x - a.seg_lo[so];
            q = q < 0 ? 0 : (q > 31 ? 31 : q);
            x = (int32_t)a.seg_tab[so * 32 + q] + 1;
        }
    }
}

__global__ __launch_bounds__(64) void k_adapt_c(EncDims d, AdArgs a)
{
    const uint32_t c = chunk_of(d, blockIdx.z), rec = blockIdx.y;
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    const AdRec r(rec);
    const AdStream st = ad_stream(d, r, c, a.cls8, a.clist, a.rle_stride, a.rlen, a.clstotal);
    const uint32_t nt = (st.n + ATILE - 1) / ATILE;
    if (k >= nt) return;
    const size_t so = ((size_t)c * 9 + rec) * d.tpc + k;
    const uint32_t flag = a.seg_flag[so];
    if (flag == 1u) return;
    const uint32_t t0 = k * ATILE, t1 = (t0 + ATILE < st.n) ? t0 + ATILE : st.n;
    if (flag == 2u) {                                          // identity segment (exponent entries only): a constant row
        uint16_t *hist = a.exph + 7 * sym_base(d, c, a.rle_stride) + (size_t)rec * sym_stride(d, c, a.rle_stride);
        const uint32_t x = (uint32_t)a.seg_start[so] & 0xffffu, xx = x | (x << 16);
        uint32_t t = t0;                                       // t0 is a multiple of ATILE: 16-byte aligned in its row
        for (; t + 8 <= t1; t += 8) *reinterpret_cast<uint4 *>(hist + t) = make_uint4(xx, xx, xx, xx);
        for (; t < t1; t++) hist[t] = (uint16_t)x;
        return;
    }
    ad_run_write(r, st, t0, t1, a.seg_start[so], a.exph + 7 * sym_base(d, c, a.rle_stride) + (size_t)(rec < 7 ? rec : 0) * sym_stride(d, c, a.rle_stride), a.mantad + sym_base(d, c, a.rle_stride));
}

// rANS records in coding order.  Pair j = 2t (exponent) / 2t+1 (mantissa) belongs to state lane j & 3; the
// records are stored lane-major (rec[lane][j >> 2]) so that every lane streams its own array.  A record is
// {low | freq << 16, Alverson reciprocal of freq}: x / freq == mulhi(x, rcp) >> (ceil(log2 freq) - 1) for x < 2^31.
// records per state lane, a multiple of the 128-record staging tile
// (a chain is padded with identity steps to whole rounds of the sixteen-batch ring, RANS_PAD = 256 steps: k_rans_lanes)
__host__ __device__ __forceinline__ size_t rans_lane_stride(size_t rle_stride) { return (rle_stride / 2 + 384) & ~(size_t)127; }
// records per lane of chunk c and the first record of its four lanes (see EncDims::sbase)
__device__ __forceinline__ size_t lane_stride_of(const EncDims &d, uint32_t c, size_t rle_stride) { return rans_lane_stride(sym_stride(d, c, rle_stride)); }
__device__ __forceinline__ size_t lane_base(const EncDims &d, uint32_t c, size_t rle_stride) { return d.lbase ? (size_t)d.lbase[c] : (size_t)c * 4 * rans_lane_stride(rle_stride); }

// 16-byte record {xmax, rcp, bias, cmpl | shift << 20}: everything a step needs that does not depend on the state
// is computed here, in parallel (ryg's RansEncSymbolInit, rans_byte.hpp:188-246, restated for 16-bit frequencies):
//   x / freq == mulhi(x, rcp) >> shift  with rcp = ceil(2^(31+s) / freq), s = ceil(log2 freq), shift = s - 1 (x < 2^31);
//   freq == 1: rcp = 2^32 - 1, shift = 0 gives q = x - 1, compensated by bias += 65535.
// (Round 6 tried the reciprocal from a 256 KB table of the 65 536 frequencies instead of the two 32-bit divisions per record -- ~70 vector
// instructions, eight per thread of k_pairs: 133 -> 169 us per launch, profiles/r06_entropy_ab.txt.  The kernel waits for its three
// dependent waves of loads, not for its vector work; the table read is a fourth.  Not kept.)
__device__ __forceinline__ uint4 rans_record(uint32_t lo, uint32_t fr)
{
    uint32_t rcp, shift, bias = lo;
    if (fr >= 2) {
        const int sh = 32 - __clz((int)(fr - 1));
        // ceil(2^(sh+31) / fr) without a 64-bit division: D = fr << (16 - sh) lies in (2^15, 2^16], 2^(sh+31) / fr == 2^47 / D, and
        // 2^47 = 2^16 * (qh * D + rh)  =>  quotient = (qh << 16) + (rh << 16) / D   (rh < D <= 2^16: everything fits 32 bits)
        const uint32_t D = fr << (16 - sh);
        const uint32_t qh = 0x80000000u / D, rh = 0x80000000u - qh * D;
        const uint32_t ql = (rh << 16) / D, rem = (rh << 16) - ql * D;
        rcp = (qh << 16) + ql + (rem ? 1u : 0u);
        shift = (uint32_t)(sh - 1);
    } else {
        rcp = 0xFFFFFFFFu; shift = 0; bias = lo + 65535u;
    }
    return make_uint4(fr << 15, rcp, bias, (65536u - fr) | (shift << 24));
}

// Two symbols per thread (round 5): symbols 2g and 2g + 1 are the pairs 4g .. 4g + 3, i.e. index g of all four lane arrays -- a wave writes
// 1 KiB runs of each -- and the loads of the two symbols (RLE0 symbol, two history entries or a table row each) are in flight together:
// the kernel waited for its loads three quarters of the time with one symbol per thread.
__global__ __launch_bounds__(TB) void k_pairs(const uint16_t *__restrict__ rle, size_t rle_stride, EncDims d, const uint32_t *__restrict__ rlen,
                                             const uint16_t *__restrict__ exph, const uint32_t *__restrict__ mantad,
                                             const uint32_t *__restrict__ ord, const uint32_t *__restrict__ qcdf, uint4 *__restrict__ recs,
                                             uint32_t *__restrict__ pairs_plain)
{
    const uint32_t c = chunk_of(d, blockIdx.y);
    const uint32_t g = blockIdx.x * TB + threadIdx.x, t0 = 2u * g;
    const uint32_t rl = rlen[c];
    const size_t lane_stride = lane_stride_of(d, c, rle_stride), lb = lane_base(d, c, rle_stride);
    uint4 *rc = recs + lb;
    // the chain kernel walks whole rounds of its sixteen-batch ring (256 steps): steps past a chain's last pair get identity records
    // (xmax above every state, q * 0 + x + 0), so that it needs no bounds logic -- at most 271 per chain, written by the threads
    // behind the chunk's last symbol
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t t = t0 + (uint32_t)h;
        if (t >= rl) {
            const uint32_t u = t - rl;
            if (u < 4u * 288u) {
                const uint32_t np = 2u * rl, chain = u / 288u;
                const uint32_t steps_pad = np ? (((np - 1u) >> 2) / 256u + 1u) * 256u : 256u;
                const uint32_t first = (chain < np) ? ((np - 1u - chain) >> 2) + 1u : 0u;
                const uint32_t k = first + u % 288u;
                if (k < steps_pad) rc[(size_t)chain * lane_stride + k] = make_uint4(0x80000000u, 0u, 0u, 0u);
            }
        }
    }
    if (t0 >= rl) return;
    const bool two = t0 + 1u < rl;
    const size_t cs = sym_stride(d, c, rle_stride), o = sym_base(d, c, rle_stride) + t0;
    const uint16_t *hrow = exph + 7 * sym_base(d, c, rle_stride) + t0;
    uint32_t sy[2], l0[2], h0[2], l1[2], f1[2], aux[2];
    int ee[2];
    // first wave of loads: the two RLE0 symbols (one aligned word when the pair is whole)
    sy[0] = rle[(size_t)c * rle_stride + t0];
    sy[1] = two ? rle[(size_t)c * rle_stride + t0 + 1u] : 0u;
#pragma unroll
    for (int h = 0; h < 2; h++) ee[h] = sym_class(sy[h]);
    // second wave: for each symbol its two history entries (entry 0 is 0 and entry 8 is 65536 by definition) and its mantissa word or ordinal
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int e = ee[h];
        const bool live = h == 0 || two;
        l0[h] = (e == 0 || !live) ? 0u : hrow[(size_t)(e - 1) * cs + h];
        h0[h] = (e == 7 || !live) ? 65536u : (uint32_t)hrow[(size_t)e * cs + h];
        aux[h] = !live ? 0u : (e < 2 ? mantad[o + h] : ord[o + h]);
    }
    // third wave: the QuasiModel table rows of the classes >= 2
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int e = ee[h];
        const uint32_t m = sy[h] - (uint32_t)class_base(e);
        if (e < 2) { l1[h] = aux[h] & 0xffffu; f1[h] = aux[h] >> 16; }
        else {
            const int q = qinterval(aux[h]);
            const uint32_t *cdf = qcdf + (((size_t)c * 6 + (e - 2)) * NQ + q) * QSTRIDE;
            l1[h] = cdf[m];
            f1[h] = cdf[m + 1] - l1[h];
        }
    }
    // lane-major: symbol 2g -> lanes 0, 1 and symbol 2g + 1 -> lanes 2, 3, all at index g
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (h == 1 && !two) break;
        rc[(size_t)(2 * h) * lane_stride + g] = rans_record(l0[h], h0[h] - l0[h]);
        rc[(size_t)(2 * h + 1) * lane_stride + g] = rans_record(l1[h], f1[h]);
        if (pairs_plain) {
            uint32_t *out = pairs_plain + (size_t)c * 2 * rle_stride + 2 * (size_t)(t0 + h);
            out[0] = l0[h] | ((h0[h] - l0[h]) << 16);
            out[1] = l1[h] | (f1[h] << 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// rANS (ans.cpp:189-208): four independent sequential chains per chunk (state lane = pair index & 3), last pair
// first.  Each step records the 0..2 renormalisation bytes it emits; their stream positions are a prefix sum.
// ---------------------------------------------------------------------------------------------------------------
// emitted bytes | count << 16 of the step that started from state x with threshold xmax
__device__ __forceinline__ uint32_t emit_word(uint32_t x, uint32_t xmax)
{
    const bool b1 = x >= xmax, b2 = (x >> 8) >= xmax;
    return b2 ? ((x & 0xffffu) | (2u << 16)) : (b1 ? ((x & 0xffu) | (1u << 16)) : 0u);
}

// One encoder step on every lane at once (see k_rans_lanes): the state comes from the lane on the left (DPP row
// rotate), keep = the state this lane had to start from once its turn came (captured when `turn` selects the lane).
// Cost on one wave (tools/issuetest.hip): a VALU instruction issues every ~4.4-5 cycles whatever it does, so the step is kept to
// twelve of them, written as one block (two steps per asm statement) because the order carries the wait states gfx940+ needs and the compiler
// cannot see into asm: two between a VALU write of an SGPR mask and the VALU reading it (cmp b2 .. select u,
// cmp b1 .. select xr) and two between the write of the new state and the next step's DPP read (capture + s_nop).
// After renormalisation xr < freq << 15, so q = xr / freq < 2^15 and 65536 - freq < 2^16: the second product fits the
// 24-bit multiply-add, which also ignores the shift count kept in the top byte of r.w.
// two consecutive steps (turns ta, tb) in one block: the compiler adds a wait state after every asm statement
#define JPK_RANS_STEP_ASM(XIN, XPREV, XOUT, TURN)                                                        \
        "v_mov_b32_dpp " XIN ", " XPREV " row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                       \
        "v_lshrrev_b32 %[x8], 8, " XIN "\n\t"                                                             \
        "v_cmp_ge_u32_e64 %[b2], %[x8], %[xmax]\n\t"    /* b2 = (x >> 8) >= xmax: two bytes leave (implies b1) */ \
        "v_cmp_ge_u32_e64 %[b1], " XIN ", %[xmax]\n\t"  /* b1 = x >= xmax: one byte leaves */              \
        "v_lshrrev_b32 %[x16], 16, " XIN "\n\t"                                                           \
        "v_cndmask_b32_e64 %[u], %[x8], %[x16], %[b2]\n\t"     /* u = b2 ? x >> 16 : x >> 8 */            \
        "v_cndmask_b32_e64 %[xr], " XIN ", %[u], %[b1]\n\t"    /* xr = b1 ? u : x */                      \
        "v_mul_hi_u32 %[q], %[xr], %[rcp]\n\t"                                                            \
        "v_lshrrev_b32_sdwa %[q], %[w], %[q] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t" \
        "v_add_u32 %[t], %[xr], %[bias]\n\t"                                                              \
        "v_mad_u32_u24 " XOUT ", %[q], %[w], %[t]\n\t"  /* == ((xr / freq) << 16) + xr % freq + low */     \
        "v_cndmask_b32_e64 %[keep], %[keep], " XIN ", " TURN "\n\t"  /* my turn: remember the state I started from */ \
        "s_nop 0\n\t"
__device__ __forceinline__ uint32_t rans_step_turn2(uint32_t xprev, const uint4 r, uint32_t &keep, uint64_t ta, uint64_t tb)
{
    uint32_t xin, x8, x16, u, xr, q, t, xm, xn;
    uint64_t b1, b2;
    asm(JPK_RANS_STEP_ASM("%[xin]", "%[xprev]", "%[xm]", "%[ta]")
        JPK_RANS_STEP_ASM("%[xin]", "%[xm]", "%[xn]", "%[tb]")
        : [xin] "=&v"(xin), [x8] "=&v"(x8), [x16] "=&v"(x16), [u] "=&v"(u), [xr] "=&v"(xr), [q] "=&v"(q), [t] "=&v"(t), [xm] "=&v"(xm),
          [xn] "=&v"(xn), [b1] "=&s"(b1), [b2] "=&s"(b2), [keep] "+v"(keep)
        : [xprev] "v"(xprev), [xmax] "v"(r.x), [rcp] "v"(r.y), [bias] "v"(r.z), [w] "v"(r.w), [ta] "s"(ta), [tb] "s"(tb));
    return xn;
}

// One wave per chunk.  The four chains of a chunk (pair j -> chain j & 3, ans.cpp:189-208) own one row of
// 16 lanes each; lane s of a row holds the record of step K - s of the current batch of 16 steps.  The state travels: every step
// all lanes run the same twelve instructions on the state handed over by their left neighbour (DPP row rotate, no readlane), so
// after step t lane t holds the chain's true state and the others compute values nobody uses.  A lane keeps the state it was
// handed on its own turn; per batch the wave stores those 64 states (one coalesced 4-byte store) and nothing else: the emitted
// bytes and their stream positions are a pure function of (state before the step, record) and are computed afterwards by wide
// kernels (k_emit_count / k_emit_prefix / k_put_payload).  This wave is the serial floor of the whole stage -- 13 issue slots per
// step plus ~1 per step of batch overhead.  Steps past a chain's last pair use the identity records k_pairs has written.
//
// Record prefetch, correct by construction (round 4).  Sixteen batches of records are in flight (~6 us of steps: enough to cover a
// record load while suffix-sort kernels of another block saturate the memory system).  Rounds 2-3 kept them in sixteen register
// tuples loaded from inline asm, which the compiler believed to be valid from the asm statement on: a register copy it chose to
// insert (loop back-edge, spill) before the hand-counted s_waitcnt copied stale registers -- timing-dependent wrong bytes, guarded
// only by a regex over the assembly.  Now the records land in an LDS ring (global_load_lds_dwordx4: 1 KiB per wave-instruction,
// lane l -> ring[slot][l]) and NO register holds data in flight: a record becomes a register value through an ordinary ds_read
// that the compiler tracks itself (lgkmcnt), placed behind an asm s_waitcnt with a memory clobber that it cannot be moved across.
// What remains hand-counted is vmcnt, and the count is exact by program text: every vector-memory instruction of the loop (one
// LDS-DMA load and two stores per batch: the states' low halves and, round 6, the emit masks) is issued from asm volatile, loads and stores share vmcnt and retire in order, and
// anything the compiler might add (a spill) is YOUNGER than the load being waited for or older than all of them, which can only make
// the wait stricter.  tests/test_chain_codegen.py checks the generated loop for exactly that instruction census.
constexpr int RANS_RING = 16;                   // batches of records in flight
constexpr int RANS_SLACK = 16 * RANS_RING;      // records in front of the record array: the prefetches of batches that do not exist read there
// One statement per batch for the four instructions that touch vector memory or M0: the LDS-DMA load takes its LDS address from
// M0 and needs one instruction between the SALU write of M0 and itself (the stores are), and M0 is
// compiler-reserved: the kernel uses it nowhere else (tests/test_chain_codegen.py checks), so it is not saved.  Addresses: a
// wave-uniform base in SGPRs + a 32-bit lane offset + an immediate -- inside the sixteen-fold unrolled loop body nothing is computed.
// (The LDS-DMA load carries no immediate: its instruction offset moves the LDS address as well as the global one -- measured the
// hard way -- so the lane offset of the records is stepped by one v_add per batch; the store's offset is an immediate.)
#define JPK_RING_STORE_LOAD(SLOT, SOFF, MOFF)                                                                          \
    asm volatile("s_mov_b32 m0, %[slot]\n\t"                                                                           \
                 "global_store_short %[xoff], %[keep], %[xbase] offset:" #SOFF "\n\t"                                  \
                 "global_store_dword %[moff], %[mw], %[mbase] offset:" #MOFF "\n\t"                                    \
                 "global_load_lds_dwordx4 %[roff], %[rbase]"                                                           \
                 : : [slot] "s"(ring0 + (uint32_t)((SLOT) & (RANS_RING - 1)) * 1024u), [xoff] "v"(xoff), [keep] "v"(keep), [xbase] "s"(xbase), [moff] "v"(moff), [mw] "v"(mw),   \
                     [mbase] "s"(mbase), [roff] "v"(roff), [rbase] "s"(rbase)                                          \
                 : "memory")
__global__ __launch_bounds__(64) void k_rans_lanes(const uint4 *__restrict__ recs, size_t rle_stride, EncDims d, const uint32_t *__restrict__ rlen,
                                                  uint16_t *__restrict__ x16, uint32_t *__restrict__ emask, uint32_t *__restrict__ fstate,
                                                  uint64_t *__restrict__ stamp)
{
    __shared__ uint4 ring[RANS_RING][64];
    const uint64_t t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    // this wave is a long dependent chain: let it win the issue arbitration against the wide kernels of other chunks /
    // blocks that share its SIMD (priority, then age)
    __builtin_amdgcn_s_setprio(3);
    const uint32_t c = chunk_of(d, blockIdx.x);
    const int t = threadIdx.x;
    const uint32_t np = 2 * rlen[c];
    if (np == 0) {
        if (t < 4) fstate[(size_t)c * 4 + t] = RANS_L;
        if (t == 0) { stamp[2 * (size_t)c] = 0; stamp[2 * (size_t)c + 1] = 0; }
        return;
    }
    const int chain = t >> 4, s = t & 15;
    const size_t lane_stride = lane_stride_of(d, c, rle_stride), lb = lane_base(d, c, rle_stride);
    // batches of sixteen steps, rounded up to whole rounds of the ring (round 6: the padding -- identity records, k_pairs -- is walked
    // first, from the start state, and costs at most fifteen batches per chunk; the exit test behind every batch of the unrolled loop,
    // three scalar instructions of 231, is gone)
    const int32_t nbatch = (((int32_t)((np - 1) / 4) / 16 + 1) + RANS_RING - 1) & ~(RANS_RING - 1);
    const int32_t K0 = 16 * nbatch - 1 - s;                        // this lane's step index in the first batch
    uint32_t x = RANS_L;                                           // lane 15 of a row hands the start state to lane 0
    const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&ring[0][0];     // LDS byte address
    // wave-uniform bases (SGPR pairs) and 32-bit lane offsets in bytes.  The record base sits RANS_SLACK records in front of the
    // chunk's first record, so that the offset of a prefetch past the last batch (step index down to -256) stays non-negative: those
    // loads read the slack enc_layout leaves in front of the array (or the chains of the chunk before) and nobody uses them.
    const uint4 *rbase = recs + lb - RANS_SLACK;
    uint16_t *xbase = x16 + lb;
    uint32_t *mbase = emask + (lb >> 4);                           // (lane bases and strides are multiples of 128 records)
    uint32_t roff = (uint32_t)(((size_t)chain * lane_stride + (size_t)(K0 + RANS_SLACK)) * 16u);
    uint32_t xoff = (uint32_t)(((size_t)chain * lane_stride + (size_t)K0) * 2u);
    uint32_t moff = (uint32_t)((((size_t)chain * lane_stride) >> 4) + (size_t)(nbatch - 1)) * 4u;    // the row's mask word of the first batch
    const uint32_t rowsh = 16u * (uint32_t)chain;
    // prologue: the first sixteen batches' records, all landed before the loop starts -- one memory latency per chunk -- so that
    // from here on the steady-state count below holds for every wait
#define JPK_RING_PROLOGUE(SLOT)                                                                                         \
    asm volatile("s_mov_b32 m0, %[slot]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[roff], %[rbase]"                        \
                 : : [slot] "s"(ring0 + (uint32_t)(SLOT) * 1024u), [roff] "v"(roff), [rbase] "s"(rbase) : "memory");      \
    roff -= 256u;
    JPK_RING_PROLOGUE(0)  JPK_RING_PROLOGUE(1)  JPK_RING_PROLOGUE(2)  JPK_RING_PROLOGUE(3)
    JPK_RING_PROLOGUE(4)  JPK_RING_PROLOGUE(5)  JPK_RING_PROLOGUE(6)  JPK_RING_PROLOGUE(7)
    JPK_RING_PROLOGUE(8)  JPK_RING_PROLOGUE(9)  JPK_RING_PROLOGUE(10) JPK_RING_PROLOGUE(11)
    JPK_RING_PROLOGUE(12) JPK_RING_PROLOGUE(13) JPK_RING_PROLOGUE(14) JPK_RING_PROLOGUE(15)
#undef JPK_RING_PROLOGUE
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");            // (roff now points at the lane's record sixteen batches ahead)
    uint4 cur = ring[0][t];
    // One batch: the record of the NEXT batch (slot k + 1) was requested sixteen batches ago minus one; issued after it, in program
    // order: the two stores and the load of each of 14 batches -- vmcnt(42) -- (the first fifteen batches read slots the
    // prologue has drained).  It becomes a register value through an ordinary LDS read behind the wait, used one batch later.
    // Then 16 steps of every chain on `cur`, the store of the kept states, and the request for batch + 16 into the slot `cur` came from.
#define JPK_BATCH(KSLOT, SOFF, MOFF)                                                                                    \
    {                                                                                                                   \
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(3 * (RANS_RING - 2)) : "memory");                                    \
        const uint4 nxt = ring[((KSLOT) + 1) & (RANS_RING - 1)][t];                                                     \
        _Pragma("unroll") for (int st = 0; st < 16; st += 2)                                                            \
            x = rans_step_turn2(x, cur, keep, 0x0001000100010001ull << st, 0x0001000100010001ull << (st + 1));          \
        /* what the step of every lane emitted (round 6): one byte when its start state reached xmax, two when state >> 8 did -- as a     \
           bit per lane, the sixteen lanes of a row = the sixteen steps of the row's chain in this batch (bit s <-> step K - s) */       \
        const uint64_t e1 = __ballot(keep >= cur.x), e2 = __ballot((keep >> 8) >= cur.x);                               \
        const uint32_t mw = ((uint32_t)(e1 >> rowsh) & 0xffffu) | ((uint32_t)(e2 >> rowsh) << 16);                      \
        JPK_RING_STORE_LOAD(KSLOT, SOFF, MOFF);                                                                         \
        roff -= 256u;                                                                                                   \
        cur = nxt;                                                                                                      \
    }
    uint32_t keep = 0;                                             // (every lane's turn comes once per batch and overwrites it)
    for (int32_t left = nbatch; left > 0; left -= RANS_RING) {    // whole rounds: no exit inside
        JPK_BATCH(0, 0, 0)
        JPK_BATCH(1, -32, -4)
        JPK_BATCH(2, -64, -8)
        JPK_BATCH(3, -96, -12)
        JPK_BATCH(4, -128, -16)
        JPK_BATCH(5, -160, -20)
        JPK_BATCH(6, -192, -24)
        JPK_BATCH(7, -224, -28)
        JPK_BATCH(8, -256, -32)
        JPK_BATCH(9, -288, -36)
        JPK_BATCH(10, -320, -40)
        JPK_BATCH(11, -352, -44)
        JPK_BATCH(12, -384, -48)
        JPK_BATCH(13, -416, -52)
        JPK_BATCH(14, -448, -56)
        JPK_BATCH(15, -480, -60)
        xoff -= 16u * 32u;
        moff -= 16u * 4u;
    }
#undef JPK_BATCH
#undef JPK_RING_STORE_LOAD
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");            // the prefetches of batches that do not exist must land before the LDS is released
    if (s == 15) fstate[(size_t)c * 4 + chain] = x;               // the last batch ends at step 0: lane 15 holds each chain's final state
    if (t == 0) {
        // diagnostics: shader cycles and 100 MHz ticks this chunk's chains took (jpk_stats.enc_chain_*)
        stamp[2 * (size_t)c] = __builtin_amdgcn_s_memtime() - t0c;
        stamp[2 * (size_t)c + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
    }
}

// ---- emitted bytes and their positions, in parallel over all pairs ---------------------------------------------------------
// Pair j of a chunk (chain j & 3, step j >> 2) emitted 0..2 bytes; the encoder walks the pairs last to first and the stream grows
// downwards, so the bytes of pair j start  sum_{j' >= j} count(j')  bytes before the end of the chunk's payload.
constexpr int ETILE = 4096;                    // pairs per tile
// Round 6: what the emit kernels read is what the chain left behind for them -- per step the low 16 bits of the state it started
// from (the only bits a step can emit) and, per chain and batch of sixteen steps, one word of emit masks (low half: "one byte at
// least", high half: "two"; bit s <-> step 16 m + 15 - s of the chain: lane s of the chain's row, k_rans_lanes) -- 2.25 bytes per
// pair.  Rounds 2-5 kept the whole 32-bit state and a 2-byte frequency sidecar written by k_pairs and recomputed the comparison
// against x_max twice: 12 bytes read per pair to place 0.17 (profiles/r05_pmc_traffic.json: k_put_payload fetched 418 MB per launch).
// mask words of one chain that are written: one per batch of the chunk
__device__ __forceinline__ uint32_t rans_batches(uint32_t np) { return np ? ((np - 1u) >> 2) / 16u + 1u : 0u; }

__global__ __launch_bounds__(TB) void k_emit_count(const uint32_t *__restrict__ emask, size_t rle_stride, EncDims d,
                                                  const uint32_t *__restrict__ rlen, uint32_t *__restrict__ tsum, uint32_t etpc)
{
    const uint32_t c = chunk_of(d, blockIdx.y), tile = blockIdx.x;
    const uint32_t np = 2 * rlen[c];
    if (tile * ETILE >= np) return;
    const size_t lane_stride = lane_stride_of(d, c, rle_stride), cb = lane_base(d, c, rle_stride);
    // the tile's pairs are the steps [1024 tile, 1024 tile + 1024) of the four chains: 64 mask words each, one per thread
    const uint32_t chain = threadIdx.x >> 6, w = tile * (ETILE / 64) + (threadIdx.x & 63u);
    const uint32_t n = w < rans_batches(np) ? (uint32_t)__popc(emask[((cb + (size_t)chain * lane_stride) >> 4) + w]) : 0u;   // (padding steps carry identity records: no bits)
    __shared__ uint32_t sm[TB / 64 + 1];
    uint32_t tot;
    block_incl_scan<OpSum>(n, sm, &tot);
    if (threadIdx.x == 0) tsum[(size_t)c * etpc + tile] = tot;
}

// one wave per chunk: tsum -> bytes emitted by all LATER tiles (exclusive suffix sum), csize = 16 state bytes + total
__global__ __launch_bounds__(64) void k_emit_prefix(EncDims d, const uint32_t *__restrict__ rlen, uint32_t *__restrict__ tsum, uint32_t etpc,
                                                   uint32_t *__restrict__ csize)
{
    const uint32_t c = chunk_of(d, blockIdx.x);
    const uint32_t np = 2 * rlen[c];
    const int nt = (int)((np + ETILE - 1) / ETILE), l = lane_id();
    uint32_t *ts = tsum + (size_t)c * etpc;
    uint32_t carry = 0;
    for (int hi = nt; hi > 0; hi -= 64) {                 // windows of 64 tiles, last window first
        const int tIdx = hi - 1 - l;                      // lane 0 = the last tile of the window
        const uint32_t v = (tIdx >= 0) ? ts[tIdx] : 0u;
        const uint32_t inc = wave_incl_sum(v);
        if (tIdx >= 0) ts[tIdx] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (l == 0) csize[c] = 16u + carry;
}

constexpr int HDR_MAX = 259 * 5 + 1;   // 1296

// header bytes of one chunk per workgroup (ans.cpp:272-285): thread s encodes Freq[s], the three length fields follow
__global__ __launch_bounds__(256) void k_headers(EncDims d, const int32_t *__restrict__ freq, const uint32_t *__restrict__ csize,
                                                const uint32_t *__restrict__ rlen, uint8_t *__restrict__ hdr, uint32_t *__restrict__ hsize)
{
    __shared__ uint32_t sm[256 / 64 + 1];
    const uint32_t c = blockIdx.x;
    uint8_t *h = hdr + (size_t)c * HDR_MAX;
    uint8_t tmp[5];
    const uint32_t n = (uint32_t)leb_encode((uint32_t)freq[(size_t)c * 256 + threadIdx.x], tmp);
    uint32_t tot;
    const uint32_t inc = block_incl_scan<OpSum>(n, sm, &tot);
    uint8_t *o = h + (inc - n);
    for (uint32_t k = 0; k < n; k++) o[k] = tmp[k];
    if (threadIdx.x == 0) {
        int p = (int)tot;
        p += leb_encode(chunk_len(d, c), h + p);
        p += leb_encode(csize[c], h + p);
        p += leb_encode(rlen[c], h + p);
        hsize[c] = (uint32_t)p;
    }
}

// output offset of every chunk, totals and the chain diagnostics -> mailbox
__global__ __launch_bounds__(64) void k_out_offsets(EncDims d, const uint32_t *__restrict__ csize, const uint32_t *__restrict__ rlen,
                                                   const uint32_t *__restrict__ hsize, uint64_t *__restrict__ outoff, uint32_t *__restrict__ mail,
                                                   const uint64_t *__restrict__ stamp)
{
    if (threadIdx.x != 0) return;
    uint64_t o = 0, rs = 0;
    uint32_t worst = 0;                      // the chunk whose chains ran longest
    for (uint32_t c = 0; c < d.nch; c++) {
        if (d.cblk && c && d.cblk[c] != d.cblk[c - 1]) o = 0;      // group encode: every block's stream starts at 0 of its own buffer
        outoff[c] = o;
        o += (uint64_t)hsize[c] + csize[c];
        rs += rlen[c];
        if (stamp[2 * (size_t)c] > stamp[2 * (size_t)worst]) worst = c;
    }
    outoff[d.nch] = o;
    mail[0] = (uint32_t)o;
    mail[1] = (uint32_t)(o >> 32);
    mail[2] = (uint32_t)rs;
    mail[3] = (uint32_t)(rs >> 32);
    mail[4] = (uint32_t)stamp[2 * (size_t)worst]; mail[5] = (uint32_t)(stamp[2 * (size_t)worst] >> 32);
    mail[6] = (uint32_t)stamp[2 * (size_t)worst + 1]; mail[7] = (uint32_t)(stamp[2 * (size_t)worst + 1] >> 32);
    mail[8] = rlen[worst];
}

__global__ __launch_bounds__(TB) void k_put_headers(EncDims d, const uint8_t *__restrict__ hdr, const uint32_t *__restrict__ hsize,
                                                   const uint64_t *__restrict__ outoff, const uint32_t *__restrict__ fstate, uint8_t *__restrict__ out)
{
    const uint32_t c = blockIdx.x;
    uint8_t *const ob = d.cblk ? d.bout[d.cblk[c]] : out;
    if (!ob) return;                        // group encode: the chunk's block does not fit its buffer (reported JPK_E_CAPACITY): nothing of it is written
    uint8_t *o = ob + outoff[c];
    const uint32_t hs = hsize[c];
    for (uint32_t i = threadIdx.x; i < hs; i += TB) o[i] = hdr[(size_t)c * HDR_MAX + i];
    if (threadIdx.x < 16) {
        // flush order R[3],R[2],R[1],R[0] downwards => forward stream R0 R1 R2 R3, little endian (ans.cpp:203-206)
        uint32_t st = fstate[(size_t)c * 4 + (threadIdx.x >> 2)];
        o[hs + threadIdx.x] = (uint8_t)(st >> (8 * (threadIdx.x & 3)));
    }
}

// payload bytes: every pair places its 0..2 bytes (offset = bytes of all later pairs, from the end of the chunk's payload)
__global__ __launch_bounds__(TB) void k_put_payload(const uint16_t *__restrict__ x16, const uint32_t *__restrict__ emask, size_t rle_stride, EncDims d,
                                                   const uint32_t *__restrict__ rlen, const uint32_t *__restrict__ tsuf, uint32_t etpc,
                                                   const uint32_t *__restrict__ csize, const uint32_t *__restrict__ hsize, const uint64_t *__restrict__ outoff,
                                                   uint8_t *__restrict__ out)
{
    const uint32_t c = blockIdx.y, tile = blockIdx.x;
    const uint32_t np = 2 * rlen[c];
    if (tile * ETILE >= np) return;
    uint8_t *const ob = d.cblk ? d.bout[d.cblk[c]] : out;
    if (!ob) return;                        // (uniform over the workgroup; see k_put_headers)
    const size_t lane_stride = lane_stride_of(d, c, rle_stride), cb = lane_base(d, c, rle_stride);
    // thread t owns the 16 consecutive pairs [j0, j0 + 16) of the tile = the four steps [s0, s0 + 4) of the four chains (pair j: chain
    // j & 3, step j >> 2): per chain one mask word (the four steps share it) and one 8-byte load of their four 16-bit states
    const uint32_t s0 = tile * (ETILE / 4) + threadIdx.x * 4u;
    const bool live = (s0 >> 4) < rans_batches(np);
    uint32_t mw[4];
    uint2 xv[4];
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {        // loads first (a live thread's four steps lie inside the chain's batches: 16 | 4)
        const size_t o = cb + (size_t)ch * lane_stride + (live ? s0 : 0u);
        mw[ch] = live ? emask[o >> 4] : 0u;
        xv[ch] = *reinterpret_cast<const uint2 *>(x16 + o);
    }
    uint32_t e[16];
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int ch = k & 3, q = k >> 2;                           // pair j0 + k
        const uint32_t bit = 15u - ((s0 + (uint32_t)q) & 15u);
        const uint32_t cnt = ((mw[ch] >> bit) & 1u) + ((mw[ch] >> (16u + bit)) & 1u);
        const uint32_t xw = q < 2 ? xv[ch].x : xv[ch].y;
        e[k] = ((q & 1) ? xw >> 16 : xw & 0xffffu) | (cnt << 16);
        n += cnt;
    }
    // bytes emitted by the threads after me = block total - inclusive prefix
    __shared__ uint32_t sm[TB / 64 + 1];
    uint32_t tot;
    const uint32_t inc = block_incl_scan<OpSum>(n, sm, &tot);
    uint32_t after = tsuf[(size_t)c * etpc + tile] + (tot - inc);       // bytes of all pairs after my 16
    uint8_t *end = ob + outoff[c] + hsize[c] + csize[c];
#pragma unroll
    for (int k = 15; k >= 0; k--) {
        const uint32_t cnt = e[k] >> 16;
        after += cnt;
        if (cnt == 1) end[-(int64_t)after] = (uint8_t)e[k];
        else if (cnt == 2) { uint8_t *p = end - after; p[0] = (uint8_t)(e[k] >> 8); p[1] = (uint8_t)e[k]; }   // the later (lower-address) byte of a step comes first
    }
}

inline uint32_t emit_tiles_per_chunk(size_t stride) { return (uint32_t)((2 * stride + ETILE - 1) / ETILE); }

struct EncBufs {
    uint32_t *tilecnt; int32_t *lastpos; int32_t *freq; uint32_t *bstart; uint8_t *ranks;
    uint32_t *lz, *ext, *tcount, *toff, *rlen;
    uint16_t *rle;
    uint32_t *clscnt, *clstotal, *ord, *qhist, *qcdf, *dens, *cmap;
    uint8_t *cls8; uint32_t *clist;
    uint32_t *seg_flag; int32_t *seg_lo, *seg_end, *seg_start; uint16_t *seg_tab;
    uint16_t *exph; uint32_t *mantad, *pairs; uint4 *recs;
    uint16_t *x16; uint32_t *emask, *etsum, *fstate, *csize;
    uint64_t *stamp;
    uint8_t *hdr; uint32_t *hsize; uint64_t *outoff;
};

enum { LAY_RANK = 1, LAY_RLE = 2, LAY_MODEL = 4, LAY_RANS = 8, LAY_PLAIN = 16 };

// `sym_total` / `lane_total`: slots of the symbol-indexed arrays and records of the rANS lanes over all chunks (the compact layout,
// EncDims::sbase; known once the RLE0 stage has run).  0: one slot per byte (the stand-alone probes).
void enc_layout(Arena &a, const EncDims &d, EncBufs &b, int what, size_t sym_total = 0, size_t lane_total = 0)
{
    const size_t tiles = (size_t)d.nch * d.tpc;
    const size_t stride = d.chunk;                 // rle symbols per chunk <= chunk bytes
    const size_t nsym = sym_total ? sym_total : (size_t)d.nch * stride;
    const size_t nlane = lane_total ? lane_total : (size_t)d.nch * 4 * rans_lane_stride(stride);
    const EncBufs keep = b;                        // (a second call adds the model / rANS stage to the rank / RLE0 buffers of the first)
    if (what & LAY_RANK) memset(&b, 0, sizeof b); else b = keep;
    if (what & LAY_RANK) {
        b.tilecnt = a.get<uint32_t>(tiles * 256);
        b.lastpos = a.get<int32_t>(tiles * 256);
        b.freq = a.get<int32_t>((size_t)d.nch * 256);
        b.bstart = a.get<uint32_t>((size_t)d.nch * 256);
        b.ranks = a.get<uint8_t>((size_t)d.nch * stride + 64);
    }
    if (what & LAY_RLE) {
        b.lz = a.get<uint32_t>(tiles);
        b.ext = a.get<uint32_t>(tiles);
        b.tcount = a.get<uint32_t>(tiles);
        b.toff = a.get<uint32_t>(tiles);
        b.rlen = a.get<uint32_t>(d.nch + 1);
        b.rle = a.get<uint16_t>((size_t)d.nch * stride + 64);
    }
    if (what & LAY_MODEL) {
        b.dens = a.get<uint32_t>(d.nch + 64);
        b.cmap = a.get<uint32_t>(d.nch + 64);
        b.clscnt = a.get<uint32_t>(tiles * 8);
        b.clstotal = a.get<uint32_t>((size_t)d.nch * 8);
        b.ord = a.get<uint32_t>(nsym);
        b.qhist = a.get<uint32_t>((size_t)d.nch * 6 * NQ * QSTRIDE);
        b.qcdf = a.get<uint32_t>((size_t)d.nch * 6 * NQ * QSTRIDE);
        b.exph = a.get<uint16_t>(7 * nsym);
        b.mantad = a.get<uint32_t>(nsym);
        b.cls8 = a.get<uint8_t>(nsym + 64);
        b.clist = a.get<uint32_t>(2 * ((nsym + 3) & ~(size_t)3) + 64);
        const size_t segs = (size_t)d.nch * 9 * d.tpc;
        b.seg_flag = a.get<uint32_t>(segs);
        b.seg_lo = a.get<int32_t>(segs);
        b.seg_end = a.get<int32_t>(segs);
        b.seg_start = a.get<int32_t>(segs);
        b.seg_tab = a.get<uint16_t>(segs * 32);
        b.recs = a.get<uint4>(nlane + RANS_SLACK) + (a.planning ? 0 : RANS_SLACK);   // slack in front: k_rans_lanes' prefetches past the last batch
        b.pairs = (what & LAY_PLAIN) ? a.get<uint32_t>((size_t)d.nch * stride * 2) : nullptr;
    }
    if (what & LAY_RANS) {
        b.x16 = a.get<uint16_t>(nlane + 64);       // low half of the state before every step, lane-major like recs
        b.emask = a.get<uint32_t>(nlane / 16 + 64); // emit masks: one word per chain and batch of sixteen steps
        b.etsum = a.get<uint32_t>((size_t)d.nch * emit_tiles_per_chunk(stride));
        b.fstate = a.get<uint32_t>((size_t)d.nch * 4);
        b.csize = a.get<uint32_t>(d.nch);
        b.stamp = a.get<uint64_t>(2 * (size_t)d.nch);
        b.hdr = a.get<uint8_t>((size_t)d.nch * HDR_MAX);
        b.hsize = a.get<uint32_t>(d.nch);
        b.outoff = a.get<uint64_t>(d.nch + 1);
    }
}

EncDims make_dims(uint32_t len, uint32_t chunk)
{
    EncDims d;
    d.len = len;
    d.chunk = chunk;
    d.nch = (len + chunk - 1) / chunk;
    d.tpc = (chunk + ATILE - 1) / ATILE;
    d.ncl = d.nch;
    d.cmap = nullptr;
    d.clen = nullptr;
    d.cblk = nullptr;
    d.bout = nullptr;
    d.sbase = nullptr;
    d.lbase = nullptr;
    return d;
}

// symbol changes per chunk: a cheap proxy for the length of its rANS chain (used to launch the densest chunks first)
__global__ __launch_bounds__(TB) void k_density(const uint8_t *__restrict__ in, EncDims d, uint32_t *__restrict__ dens)
{
    const uint32_t c = blockIdx.y, ts = blockIdx.x * ATILE * 4;
    const uint32_t clen = chunk_len(d, c);
    if (ts >= clen) return;
    const uint32_t te = (ts + ATILE * 4 < clen) ? ts + ATILE * 4 : clen;
    const uint8_t *src = in + (size_t)c * d.chunk;
    uint32_t n = 0;
    for (uint32_t i = ts + threadIdx.x * 16; i < te; i += TB * 16) {
        uint8_t prev = i ? src[i - 1] : 0;
        for (uint32_t k = 0; k < 16 && i + k < te; k++) { uint8_t v = src[i + k]; n += (v != prev); prev = v; }
    }
    n = wave_sum(n);
    if (lane_id() == 0 && n) atomicAdd(&dens[c], n);
}

// launch order of the chunks: densest first, ties by chunk number (rank by counting; one workgroup, nch <= a few thousand)
__global__ __launch_bounds__(256) void k_order(EncDims d, const uint32_t *__restrict__ dens, uint32_t *__restrict__ cmap)
{
    for (uint32_t c = threadIdx.x; c < d.nch; c += blockDim.x) {
        const uint32_t v = dens[c];
        uint32_t r = 0;
        for (uint32_t k = 0; k < d.nch; k++) {
            const uint32_t w = dens[k];
            r += (w > v || (w == v && k < c)) ? 1u : 0u;
        }
        cmap[r] = c;
    }
}

int run_rank(jpk_ctx *ctx, const uint8_t *d_in, const EncDims &d, EncBufs &b)
{
    JPK_LAUNCH(ctx, PROF_ENC_HIST, d.len, k_enc_hist, dim3(d.tpc, d.ncl), dim3(TB), d_in, d, b.tilecnt, b.lastpos);
    JPK_LAUNCH(ctx, PROF_ENC_HIST, d.len, k_enc_prep, dim3(d.ncl), dim3(256), d, b.tilecnt, b.lastpos, b.freq, b.bstart);
    JPK_LAUNCH(ctx, PROF_ENC_MTF, d.len, k_enc_mtf, dim3((d.tpc + 3) / 4, d.ncl), dim3(TB), d_in, d, b.tilecnt, b.lastpos, b.bstart, b.ranks);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

int run_rle(jpk_ctx *ctx, const uint8_t *d_ranks, const EncDims &d, EncBufs &b)
{
    JPK_LAUNCH(ctx, PROF_ENC_RLE, d.len, k_rle_lz, dim3(d.tpc, d.ncl), dim3(TB), d_ranks, d, b.lz);
    JPK_LAUNCH(ctx, PROF_ENC_RLE, d.len, k_rle_ext, dim3(jpk_grid(d.ncl, 64)), dim3(64), d, b.lz, b.ext);
    JPK_LAUNCH(ctx, PROF_ENC_RLE, d.len, (k_rle_tiles<false>), dim3(d.tpc, d.ncl), dim3(TB), d_ranks, d, b.ext, b.tcount, b.toff, b.rle, (size_t)d.chunk);
    JPK_LAUNCH(ctx, PROF_ENC_RLE, d.len, k_tile_prefix, dim3(jpk_grid(d.ncl, 64)), dim3(64), d, b.tcount, b.toff, b.rlen, 1u, (const uint32_t *)nullptr);
    JPK_LAUNCH(ctx, PROF_ENC_RLE, d.len, (k_rle_tiles<true>), dim3(d.tpc, d.ncl), dim3(TB), d_ranks, d, b.ext, b.tcount, b.toff, b.rle, (size_t)d.chunk);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

int run_model(jpk_ctx *ctx, const uint16_t *d_rle, const uint32_t *d_rlen, const EncDims &d, EncBufs &b)
{
    const size_t stride = d.chunk;
    JPK_LAUNCH(ctx, PROF_ENC_CLASS, 0, k_cls_count, dim3(d.tpc, d.ncl), dim3(TB), d_rle, stride, d, d_rlen, b.clscnt);
    JPK_LAUNCH(ctx, PROF_ENC_CLASS, 0, k_cls_prefix, dim3(jpk_grid((size_t)d.ncl * 8, 64)), dim3(64), d, d_rlen, b.clscnt, b.clstotal);
    JPK_LAUNCH(ctx, PROF_ENC_CLASS, 0, k_cls_ord, dim3(d.tpc, d.ncl), dim3(TB), d_rle, stride, d, d_rlen, b.clscnt, b.ord, b.qhist, b.cls8, b.clist);
    JPK_LAUNCH(ctx, PROF_ENC_CLASS, 0, k_quasi_build, dim3(NQ, 6, d.ncl), dim3(64), d, b.clstotal, b.qhist, b.qcdf);
    AdArgs aa;
    aa.cls8 = b.cls8; aa.clist = b.clist; aa.rle_stride = stride; aa.rlen = d_rlen; aa.clstotal = b.clstotal; aa.clsbase = b.clscnt;
    aa.exph = b.exph; aa.mantad = b.mantad;
    static const uint32_t warm = [] {
        const char *e = getenv("JPK_AD_WARM");
        long v = e ? atol(e) : (long)AD_WARM_DEFAULT;
        v = (v + 63) / 64 * 64;
        return (uint32_t)(v < 320 ? 320 : (v > 4096 ? 4096 : v));
    }();
    aa.warm = warm;
    aa.seg_flag = b.seg_flag; aa.seg_lo = b.seg_lo; aa.seg_end = b.seg_end; aa.seg_start = b.seg_start; aa.seg_tab = b.seg_tab;
    JPK_LAUNCH(ctx, PROF_ENC_ADAPTIVE, 0, k_adapt_a, dim3((d.tpc + 63) / 64, 9, d.ncl), dim3(64), d, aa);
    JPK_LAUNCH(ctx, PROF_ENC_ADAPTIVE, 0, k_adapt_ext, dim3((d.tpc + 63) / 64, 9, d.ncl), dim3(64), d, aa);
    JPK_LAUNCH(ctx, PROF_ENC_ADAPTIVE, 0, k_adapt_tab, dim3((d.tpc + 1) / 2, 9, d.ncl), dim3(64), d, aa);
    JPK_LAUNCH(ctx, PROF_ENC_ADAPTIVE, 0, k_adapt_b, dim3(jpk_grid((size_t)d.ncl * 16, 64)), dim3(64), d, aa);
    JPK_LAUNCH(ctx, PROF_ENC_ADAPTIVE, 0, k_adapt_c, dim3((d.tpc + 63) / 64, 9, d.ncl), dim3(64), d, aa);
    JPK_LAUNCH(ctx, PROF_ENC_PAIRS, 0, k_pairs, dim3(jpk_grid(stride, 2 * TB) + 3, d.ncl), dim3(TB), d_rle, stride, d, d_rlen, b.exph, b.mantad, b.ord,
                       b.qcdf, b.recs, b.pairs);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

}  // namespace

namespace {
size_t enc_plan_bytes(const EncDims &d, int nblk, uint32_t sym_per_byte_256);
// Text carries ~0.42 RLE0 symbols per byte, the densest data 1.0: the arena is first sized for 0.55 and grows (the stage starts
// over) the first time a denser block arrives.
constexpr uint32_t ENC_PLAN_SYM_256 = 141;
}  // namespace

// arena bytes of one Ans::Encode of len bytes of text-like data (jpk_ctx_reserve)
size_t jpk_ans_encode_arena_bytes(uint32_t len)
{
    if (len == 0) return 0;
    const EncDims d = make_dims(len, ANS_CHUNK);
    return enc_plan_bytes(d, 0, ENC_PLAN_SYM_256);
}
// the same for the densest input there is (every byte an RLE0 symbol): what the arena grows to when such a block arrives
size_t jpk_ans_encode_arena_bytes_worst(uint32_t len)
{
    if (len == 0) return 0;
    const EncDims d = make_dims(len, ANS_CHUNK);
    return enc_plan_bytes(d, 0, 256);
}

namespace {
// everything of the stage up to the chunks' output offsets, enqueued on ctx->stream (and the group streams): rank coding, RLE0,
// models, the rANS chains, emit counts, headers, offsets.  `d` describes the chunks (one block: make_dims; a group of blocks: the
// per-chunk tables), `inflight_n` the blocks the device is compressing right now (drives the launch grouping).
// tables a group of blocks hands in (host side): the core gives them device copies inside its own arena layout
struct GroupTabs { const uint32_t *clen, *cblk; uint8_t *const *bout; int nblk; };

// slots per chunk of the compact symbol layout (EncDims::sbase): its RLE0 symbol count rounded up to 256, prefix sums, totals -> mailbox
__global__ void k_sym_layout(EncDims d, const uint32_t *__restrict__ rlen, uint32_t *__restrict__ sbase, uint32_t *__restrict__ lbase, uint32_t *__restrict__ mail)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t so = 0, lo = 0;
    for (uint32_t c = 0; c < d.nch; c++) {
        sbase[c] = so; lbase[c] = lo;
        const uint32_t cs = (rlen[c] + 255u) & ~255u;
        so += cs;
        lo += 4u * (uint32_t)rans_lane_stride(cs);
    }
    sbase[d.nch] = so; lbase[d.nch] = lo;
    mail[12] = so; mail[13] = lo;
}

// what the stage plans for `nch` chunks when the input has `sym_per_byte_256` / 256 RLE0 symbols per byte (256 = the worst case)
size_t enc_plan_bytes(const EncDims &d, int nblk, uint32_t sym_per_byte_256)
{
    EncBufs b;
    jpk_ctx dummy;
    Arena plan(&dummy, true);
    enc_layout(plan, d, b, LAY_RANK | LAY_RLE);
    plan.get<uint32_t>((size_t)d.nch + 1);
    plan.get<uint32_t>((size_t)d.nch + 1);
    if (nblk) { plan.get<uint32_t>(d.nch); plan.get<uint32_t>(d.nch); plan.get<uint8_t *>((size_t)nblk); }
    const size_t worst = (size_t)d.nch * d.chunk;
    size_t sym = worst / 256 * sym_per_byte_256 + 256 * (size_t)d.nch;
    if (sym > worst) sym = worst;
    enc_layout(plan, d, b, LAY_MODEL | LAY_RANS, sym, 2 * sym + 4 * 384 * (size_t)d.nch);
    return plan.need;
}
int encode_core(jpk_ctx *ctx, const uint8_t *d_in, EncDims &d, EncBufs &b, int inflight_n, const GroupTabs *gt = nullptr)
{
    hipStream_t st = ctx->stream;
    const size_t stride = d.chunk;
    // ---- rank coding and RLE0 of every chunk, then the compact layout of everything that is indexed by RLE0 symbol ----------
    // The symbol count of a chunk is known on the device only: one 8-byte read back (the stage's first host synchronisation) sizes
    // the model / rANS buffers for what the block HAS -- ~33 bytes of arena per block byte on text instead of the 78 that "every
    // byte a symbol" takes.  When the arena turns out too small it grows and the stage starts over (the first dense block of a
    // context; jpk_ctx_reserve sizes for text).
    JPK_TRY(jpk_arena_ensure(ctx, ctx->arena_base + enc_plan_bytes(d, gt ? gt->nblk : 0, ENC_PLAN_SYM_256)));
    for (int attempt = 0;; attempt++) {
        Arena real(ctx, false);
        enc_layout(real, d, b, LAY_RANK | LAY_RLE);
        uint32_t *sbase = real.get<uint32_t>((size_t)d.nch + 1), *lbase = real.get<uint32_t>((size_t)d.nch + 1);
        if (gt) {
            uint32_t *d_clen = real.get<uint32_t>(d.nch), *d_cblk = real.get<uint32_t>(d.nch);
            uint8_t **d_bout = real.get<uint8_t *>((size_t)gt->nblk);
            if (ctx->arena_off > ctx->arena_cap) return JPK_E_ALLOC;
            JPK_HIP(hipMemcpyAsync(d_clen, gt->clen, sizeof(uint32_t) * d.nch, hipMemcpyHostToDevice, st));
            JPK_HIP(hipMemcpyAsync(d_cblk, gt->cblk, sizeof(uint32_t) * d.nch, hipMemcpyHostToDevice, st));
            JPK_HIP(hipMemcpyAsync(d_bout, gt->bout, sizeof(uint8_t *) * (size_t)gt->nblk, hipMemcpyHostToDevice, st));
            d.clen = d_clen; d.cblk = d_cblk; d.bout = d_bout;
        }
        if (ctx->arena_off > ctx->arena_cap) return JPK_E_ALLOC;
        d.sbase = nullptr; d.lbase = nullptr;
        JPK_TRY(run_rank(ctx, d_in, d, b));
        JPK_TRY(run_rle(ctx, b.ranks, d, b));
        hipLaunchKernelGGL(k_sym_layout, dim3(1), dim3(64), 0, st, d, b.rlen, sbase, lbase, ctx->d_mail);
        uint32_t mail[14];
        JPK_TRY(jpk_read_mail(ctx, mail, 14));
        const size_t sym_total = mail[12], lane_total = mail[13];
        Arena rest(ctx, true);
        EncBufs dummy = b;
        enc_layout(rest, d, dummy, LAY_MODEL | LAY_RANS, sym_total ? sym_total : 256, lane_total ? lane_total : 512);
        if (ctx->arena_off + rest.need > ctx->arena_cap) {
            if (attempt) return JPK_E_ALLOC;
            JPK_TRY(jpk_arena_ensure(ctx, ctx->arena_off + rest.need));      // frees and re-allocates: the buffers above are gone
            continue;
        }
        enc_layout(real, d, b, LAY_MODEL | LAY_RANS, sym_total ? sym_total : 256, lane_total ? lane_total : 512);
        d.sbase = sbase; d.lbase = lbase;
        break;
    }
    // (clearing the histograms inside k_cls_count instead of a memset of its own: k_cls_ord 64 -> 85 us per launch, round 6 -- not kept)
    JPK_HIP(hipMemsetAsync(b.qhist, 0, (size_t)d.nch * 6 * NQ * QSTRIDE * 4, st));
    // The stage is bounded by the longest rANS chain (the densest chunk), a single wave, and several of the parallel
    // kernels in front of it are latency-bound per chunk as well.  The chunks are ordered by a density estimate and cut
    // into up to four groups, densest first, and each group runs the whole stage on its own stream: the densest
    // chains start after a quarter of the parallel work, beside the parallel stages of the other groups.
    auto pre_chain = [&](const EncDims &g) -> int {
        JPK_TRY(run_model(ctx, b.rle, b.rlen, g, b));
        return JPK_OK;
    };
    auto chain = [&](const EncDims &g) -> int {
        JPK_LAUNCH(ctx, PROF_ENC_RANS, 0, k_rans_lanes, dim3(g.ncl), dim3(64), b.recs, stride, g, b.rlen, b.x16, b.emask, b.fstate, b.stamp);
        const uint32_t etpc = emit_tiles_per_chunk(stride);
        JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_emit_count, dim3(etpc, g.ncl), dim3(TB), b.emask, stride, g, b.rlen, b.etsum, etpc);
        JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_emit_prefix, dim3(g.ncl), dim3(64), g, b.rlen, b.etsum, etpc, b.csize);
        return JPK_OK;
    };
    // Graded launch groups shorten ONE block (its longest chains start after a sixteenth of the parallel work) but cost throughput
    // when other blocks fill the machine anyway (more kernels, more streams on the hardware queues): 4 groups alone, 2 beside one
    // other block, 1 beside two or more (default bench, 4 blocks in flight: 3.23 / 3.32 / 3.46 GB/s with 4 / 2 / 1 groups)
    int ngroups = jpk_enc_groups_for(inflight_n, d.nch);
    static const int groups_env = [] { const char *e = getenv("JPK_ENC_GROUPS"); return e ? atoi(e) : 0; }();
    if (groups_env >= 1 && groups_env <= jpk_ctx::ENC_GROUPS && (uint32_t)groups_env <= d.nch) ngroups = groups_env;
    // group streams are created when a block first needs them and then stay with the context (parked while the device is busy
    // with other blocks: an idle stream costs nothing, destroying and re-creating it under fluctuating load costs a synchronise)
    for (int g = 0; g + 1 < ngroups; g++)
        if (!ctx->aux[g] && hipStreamCreateWithFlags(&ctx->aux[g], hipStreamNonBlocking) != hipSuccess) { ctx->aux[g] = nullptr; ngroups = g + 1; break; }
    if (ngroups >= 2) {
        // launch order on the device (no host round trip): density per chunk, rank by counting
        JPK_HIP(hipMemsetAsync(b.dens, 0, (size_t)d.nch * 4, st));
        JPK_LAUNCH(ctx, PROF_ENC_HIST, d.len, k_density, dim3((d.chunk + ATILE * 4 - 1) / (ATILE * 4), d.nch), dim3(TB), d_in, d, b.dens);
        JPK_LAUNCH(ctx, PROF_ENC_HIST, 0, k_order, dim3(1), dim3(256), d, b.dens, b.cmap);
        JPK_HIP(hipEventRecord(ctx->ev_pre[0], st));           // histogram memset + chunk order are in place
        // graded groups, densest first: the first group is small so that the longest chains -- the serial floor of the whole
        // stage -- start after a sixteenth of the parallel work; with four groups: 1/16, 1/8, 1/4 of the chunks and the rest
        uint32_t gsz[jpk_ctx::ENC_GROUPS];
        {
            uint32_t left = d.nch;
            for (int g = 0; g < ngroups; g++) {
                uint32_t n = (g + 1 == ngroups) ? left : d.nch >> (ngroups - g);
                if (n < 2) n = 2;
                if (n > left) n = left;
                gsz[g] = n;
                left -= n;
            }
        }
        uint32_t off = 0;
        int rc = JPK_OK;
        for (int g = 0; g < ngroups && rc == JPK_OK; g++) {
            const uint32_t n = gsz[g];
            if (n == 0) continue;
            EncDims gd = d;
            gd.ncl = n; gd.cmap = b.cmap + off;
            off += n;
            // every buffer of the stage is indexed by chunk, so the groups are independent: each runs on its own stream
            hipStream_t gs = (g + 1 < ngroups) ? ctx->aux[g] : st;
            if (gs != st && hipStreamWaitEvent(gs, ctx->ev_pre[0], 0) != hipSuccess) { rc = JPK_E_DEVICE; break; }
            ctx->stream = gs;
            rc = pre_chain(gd);
            if (rc == JPK_OK) rc = jpk_gate_mark(ctx, gs);        // the wide kernels of this group end here; the chains run beside the next block
            if (rc == JPK_OK) rc = chain(gd);
            ctx->stream = st;
            if (rc == JPK_OK && gs != st && hipEventRecord(ctx->ev_done[g], gs) != hipSuccess) rc = JPK_E_DEVICE;
        }
        jpk_gate_leave(ctx);                                       // everything GPU-saturating of this block is enqueued
        for (int g = 0; g + 1 < ngroups && rc == JPK_OK; g++)
            if (hipStreamWaitEvent(st, ctx->ev_done[g], 0) != hipSuccess) rc = JPK_E_DEVICE;
        if (rc != JPK_OK) {
            // groups already launched keep reading and writing the arena: join them before the caller may reuse it
            for (int g = 0; g + 1 < jpk_ctx::ENC_GROUPS; g++) if (ctx->aux[g]) (void)hipStreamSynchronize(ctx->aux[g]);
            (void)hipStreamSynchronize(st);
            return rc;
        }
    } else {
        int rc = pre_chain(d);
        if (rc == JPK_OK) rc = jpk_gate_mark(ctx, st);
        jpk_gate_leave(ctx);
        JPK_TRY(rc);
        JPK_TRY(chain(d));
    }
    JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_headers, dim3(d.nch), dim3(256), d, b.freq, b.csize, b.rlen, b.hdr, b.hsize);
    JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_out_offsets, dim3(1), dim3(64), d, b.csize, b.rlen, b.hsize, b.outoff, ctx->d_mail, b.stamp);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}
}  // namespace

int jpk_ans_encode_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    const JpkCompressInflight inflight(ctx->device);      // blocks of this device in their forward BWT or entropy encode right now, this one included
    *out_len = 0;
    ctx->stats.ans_chunks = 0;
    ctx->stats.ans_rle_symbols = 0;
    if (len == 0) return JPK_OK;
    hipStream_t st = ctx->stream;
    EncDims d = make_dims((uint32_t)len, ANS_CHUNK);
    EncBufs b;
    memset(&b, 0, sizeof b);
    JPK_TRY(encode_core(ctx, d_in, d, b, inflight.n));
    const size_t stride = d.chunk;
    uint32_t mail[9];
    JPK_TRY(jpk_read_mail(ctx, mail, 9));
    ctx->stats.enc_chain_cycles = (int64_t)(((uint64_t)mail[5] << 32) | mail[4]);
    ctx->stats.enc_chain_ns = (int64_t)(((uint64_t)mail[7] << 32) | mail[6]) * 10;     // s_memrealtime ticks at 100 MHz
    ctx->stats.enc_chain_steps = ((int64_t)mail[8] * 2 + 3) / 4;
    const uint64_t total = ((uint64_t)mail[1] << 32) | mail[0];
    ctx->stats.ans_chunks = d.nch;
    ctx->stats.ans_rle_symbols = (int64_t)(((uint64_t)mail[3] << 32) | mail[2]);
    if (ctx->prof_on) {    // symbol-based kernels: units are only known now (RLE0 symbols / rANS pairs of this call)
        const uint64_t rs = (uint64_t)ctx->stats.ans_rle_symbols;
        ctx->prof_units[PROF_ENC_CLASS] += rs;
        ctx->prof_units[PROF_ENC_ADAPTIVE] += rs;
        ctx->prof_units[PROF_ENC_PAIRS] += rs;
        ctx->prof_units[PROF_ENC_RANS] += 2 * rs;
        ctx->prof_units[PROF_ENC_EMIT] += 2 * rs;
    }
    if (total > (uint64_t)out_cap) return JPK_E_CAPACITY;
    JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_put_headers, dim3(d.nch), dim3(TB), d, b.hdr, b.hsize, b.outoff, b.fstate, d_out);
    JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_put_payload, dim3(emit_tiles_per_chunk(stride), d.nch), dim3(TB), b.x16, b.emask, stride, d, b.rlen, b.etsum,
               emit_tiles_per_chunk(stride), b.csize, b.hsize, b.outoff, d_out);
    JPK_HIP(hipGetLastError());
    JPK_HIP(hipStreamSynchronize(st));
    *out_len = (int32_t)total;
    return JPK_OK;
}

// ---- group encode: the images of several (small) blocks through ONE set of grids ---------------------------------------------
// Chunks are independent (ans.cpp:136-140), so the chunks of all blocks of a group run through the same launches; a block's image
// starts at a chunk boundary of the staging buffer `d_stage` (block b at first_chunk[b] * 1 MiB), every chunk knows its length, its
// block and, through the block, its output buffer; offsets restart at every block.  Host synchronisations per group: the symbol layout (encode_core), the chunk sizes, the end of
// the emit kernels, and one more when a block does not fit its buffer.
size_t jpk_ans_encode_group_arena_bytes(uint32_t nchunks, int nblk)
{
    const EncDims d = make_dims(nchunks * ANS_CHUNK, ANS_CHUNK);
    return enc_plan_bytes(d, nblk, ENC_PLAN_SYM_256);
}

// The arena (from ctx->arena_base on) must hold jpk_ans_encode_group_arena_bytes(); the caller has sized it.
int jpk_ans_encode_group_device(jpk_ctx *ctx, int nblk, const uint8_t *d_stage, const uint32_t *first_chunk, const int32_t *mid_len, uint8_t *const *d_out,
                                const int32_t *out_cap, int32_t *out_len, int32_t *status)
{
    const JpkCompressInflight inflight(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<uint32_t> clen, cblk;
    for (int b = 0; b < nblk; b++) {
        out_len[b] = 0;
        status[b] = JPK_OK;
        const uint32_t ml = (uint32_t)mid_len[b], nc = (ml + ANS_CHUNK - 1) / ANS_CHUNK;
        if (clen.size() != first_chunk[b]) return JPK_E_ARG;                   // images are laid out back to back, in block order
        for (uint32_t k = 0; k < nc; k++) {
            clen.push_back(k + 1 < nc ? (uint32_t)ANS_CHUNK : ml - k * (uint32_t)ANS_CHUNK);
            cblk.push_back((uint32_t)b);
        }
    }
    const uint32_t nch = (uint32_t)clen.size();
    if (nch == 0) return JPK_OK;
    EncDims d = make_dims(nch * ANS_CHUNK, ANS_CHUNK);
    EncBufs b;
    memset(&b, 0, sizeof b);
    const GroupTabs gt = {clen.data(), cblk.data(), d_out, nblk};     // (host vectors: they outlive the synchronisations below)
    JPK_TRY(encode_core(ctx, d_stage, d, b, inflight.n, &gt));
    uint8_t **d_bout = const_cast<uint8_t **>(d.bout);
    // sizes: header + payload of every chunk; one copy, the group's only host synchronisation before the payload is placed
    std::vector<uint32_t> hs(nch), cs(nch);
    JPK_HIP(hipMemcpyAsync(hs.data(), b.hsize, sizeof(uint32_t) * nch, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipMemcpyAsync(cs.data(), b.csize, sizeof(uint32_t) * nch, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> tot((size_t)nblk, 0);
    for (uint32_t c = 0; c < nch; c++) tot[cblk[c]] += (uint64_t)hs[c] + cs[c];
    bool all_fit = true;
    for (int k = 0; k < nblk; k++)
        if (tot[k] > (uint64_t)out_cap[k]) { status[k] = JPK_E_CAPACITY; all_fit = false; }
    if (!all_fit) {
        // a block that does not fit its buffer is not written at all: its entry of the output table becomes null and the two emit
        // kernels skip its chunks; the blocks that fit are emitted as usual (no scratch sink, no whole-group failure)
        std::vector<uint8_t *> outs(d_out, d_out + nblk);
        for (int k = 0; k < nblk; k++) if (status[k] != JPK_OK) outs[k] = nullptr;
        JPK_HIP(hipMemcpyAsync(d_bout, outs.data(), sizeof(uint8_t *) * (size_t)nblk, hipMemcpyHostToDevice, st));
        JPK_HIP(hipStreamSynchronize(st));                                     // (`outs` is a local: the copy must have read it)
    }
    const size_t stride = d.chunk;
    JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_put_headers, dim3(d.nch), dim3(TB), d, b.hdr, b.hsize, b.outoff, b.fstate, (uint8_t *)nullptr);
    JPK_LAUNCH(ctx, PROF_ENC_EMIT, 0, k_put_payload, dim3(emit_tiles_per_chunk(stride), d.nch), dim3(TB), b.x16, b.emask, stride, d, b.rlen, b.etsum,
               emit_tiles_per_chunk(stride), b.csize, b.hsize, b.outoff, (uint8_t *)nullptr);
    JPK_HIP(hipGetLastError());
    JPK_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < nblk; k++) if (status[k] == JPK_OK) out_len[k] = (int32_t)tot[k];
    ctx->stats.ans_chunks = (int32_t)nch;
    return JPK_OK;
}

// Postcoder::Encode (rank.cpp:45-90) for one buffer of any length, in place
int jpk_rank_encode_device(jpk_ctx *ctx, uint8_t *d_t, int32_t *d_freq, int32_t len)
{
    if (len == 0) {
        JPK_HIP(hipMemsetAsync(d_freq, 0, 256 * 4, ctx->stream));
        return JPK_OK;
    }
    const EncDims d = make_dims((uint32_t)len, (uint32_t)len);
    EncBufs b;
    Arena plan(ctx, true);
    enc_layout(plan, d, b, LAY_RANK);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    enc_layout(real, d, b, LAY_RANK);
    JPK_TRY(run_rank(ctx, d_t, d, b));
    JPK_HIP(hipMemcpyAsync(d_t, b.ranks, (size_t)len, hipMemcpyDeviceToDevice, ctx->stream));
    JPK_HIP(hipMemcpyAsync(d_freq, b.freq, 256 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return JPK_OK;
}

// RLE::encode (rle.cpp:22-47) of one chunk
int jpk_rle_encode_device(jpk_ctx *ctx, const uint8_t *d_ranks, int32_t len, uint16_t *d_rle, int32_t *rlen)
{
    *rlen = 0;
    if (len == 0) return JPK_OK;
    const EncDims d = make_dims((uint32_t)len, (uint32_t)len);
    EncBufs b;
    Arena plan(ctx, true);
    enc_layout(plan, d, b, LAY_RLE);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    enc_layout(real, d, b, LAY_RLE);
    JPK_TRY(run_rle(ctx, d_ranks, d, b));
    JPK_HIP(hipMemcpyAsync(ctx->d_mail, b.rlen, 4, hipMemcpyDeviceToDevice, ctx->stream));
    uint32_t n = 0;
    JPK_TRY(jpk_read_mail(ctx, &n, 1));
    JPK_HIP(hipMemcpyAsync(d_rle, b.rle, (size_t)n * 2, hipMemcpyDeviceToDevice, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    *rlen = (int32_t)n;
    return JPK_OK;
}

// model pass of one chunk: rle symbols -> packed pairs
int jpk_model_pairs_device(jpk_ctx *ctx, const uint16_t *d_rle, int32_t rlen, uint32_t *d_pairs)
{
    if (rlen == 0) return JPK_OK;
    const EncDims d = make_dims((uint32_t)rlen, (uint32_t)rlen);
    EncBufs b;
    Arena plan(ctx, true);
    enc_layout(plan, d, b, LAY_RLE | LAY_MODEL | LAY_PLAIN);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    enc_layout(real, d, b, LAY_RLE | LAY_MODEL | LAY_PLAIN);
    uint32_t n = (uint32_t)rlen;
    JPK_HIP(hipMemcpyAsync(b.rlen, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    JPK_HIP(hipMemsetAsync(b.qhist, 0, (size_t)d.nch * 6 * NQ * QSTRIDE * 4, ctx->stream));
    JPK_TRY(run_model(ctx, d_rle, b.rlen, d, b));
    JPK_HIP(hipMemcpyAsync(d_pairs, b.pairs, (size_t)rlen * 8, hipMemcpyDeviceToDevice, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    return JPK_OK;
}
