// bwt_fwd.hip -- forward BWT on gfx950: GPU suffix-array construction (replaces divsufsort, divsufsort.cpp:1721)
// followed by the BWT gather and the 120 sampled ranks of BlockSort::Bwt::ForwardBwt (bwt.cpp:22-65).
//
// Suffix array = prefix doubling (Larsson-Sadakane ranks) with compaction of resolved suffixes:
//   round 0   key = first 7 bytes (big-endian, zero padded): one LSD radix sort of all n suffixes, 7 passes (radix.hip),
//             fed in descending text position so that a short suffix -- a proper prefix of anything it ties with on
//             the padded bytes -- comes first: plain suffix order even when the text contains 0x00.
//   round h   (h = 7, 14, 28, ...) active suffixes only.  The active list keeps groups of equal h-rank contiguous
//             and in SA order, so a group is sorted by key2 = rank[sa + h] + 1 (0 past the end) independently:
//               * groups of <= 1024 suffixes: k_seg_round -- one workgroup owns the groups that start in its
//                 1024-element window, stages (sa, key2, group id) in LDS, LDS radix sort, re-ranks, writes ISA,
//                 drops singletons into SA.  One read + one write of the active list per round.
//               * larger groups: compacted and sent through the global radix sort on (group rank << 32 | key2).
//             Ranks are double-buffered (read ISA_cur, write ISA_nxt) so that the gathers of a round never see
//             ranks written by the same round.
// Every array stays in HBM (T n, ISA 2 x 4n, SA 4n, sort ping-pong 24n, active lists 16n, temps 16n).
#include "common.hpp"
#include "prims.hpp"

using namespace jpk;

namespace {

constexpr int TB = 256;
constexpr uint32_t DONE = 0x80000000u;

// ---- round 0 ----------------------------------------------------------------------------------------
// key = first 7 bytes, big-endian in bits 63..8, zero padded past the end of the text.  Slot j holds suffix n-1-j:
// the LSD sort is stable, so suffixes that tie on the padded bytes come out in DESCENDING text position, i.e. a
// short suffix (a proper prefix of everything it ties with) lands in front -- plain suffix order even when the text
// contains 0x00 -- and the low byte of the key needs no sort pass.
__global__ __launch_bounds__(TB) void k_init_keys(const uint8_t *__restrict__ T, uint32_t n, uint64_t *__restrict__ keys,
                                                 uint32_t *__restrict__ vals)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = n - 1 - j;
    uint32_t left = n - i;
    uint64_t k = 0;
#pragma unroll
    for (int b = 0; b < 7; b++) {
        uint64_t c = (b < (int)left) ? T[i + b] : 0;
        k |= c << (56 - 8 * b);
    }
    keys[j] = k;
    vals[j] = i;
}

// head flags of equal-key runs -> hv[i] = head ? i : 0  (input of an inclusive max scan); a suffix with fewer than
// 7 bytes is always a group of its own
__global__ __launch_bounds__(TB) void k_heads_u64(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sa, uint32_t m, uint32_t n,
                                                 uint32_t *__restrict__ hv)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    bool head = (j == 0) || (keys[j] != keys[j - 1]) || (sa[j] + 7u > n) || (sa[j - 1] + 7u > n);
    hv[j] = head ? j : 0u;
}

// round 0: grp[] (= index of the run head) -> both rank buffers, singletons -> SA, keep flags for compaction
__global__ __launch_bounds__(TB) void k_round0_finish(const uint32_t *__restrict__ grp, const uint32_t *__restrict__ sa, uint32_t n,
                                                     uint32_t *__restrict__ ISA0, uint32_t *__restrict__ ISA1, uint32_t *__restrict__ SA,
                                                     uint32_t *__restrict__ keep)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= n) return;
    uint32_t g = grp[j], s = sa[j];
    ISA0[s] = g;
    bool head = (g == j);
    bool next_head = (j + 1 == n) || (grp[j + 1] == j + 1);
    bool single = head && next_head;
    if (single) { SA[j] = s; ISA1[s] = g; }      // finished: final rank in both buffers; the rest is rewritten by round 1
    keep[j] = single ? 0u : 1u;
}

// stream compaction of the survivors: (sa, grp) -> active lists
__global__ __launch_bounds__(TB) void k_compact(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ sa,
                                               const uint32_t *__restrict__ grp, uint32_t m, uint32_t *__restrict__ a_sa,
                                               uint32_t *__restrict__ a_grp)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    if (keep[j]) {
        uint32_t p = pos[j];
        a_sa[p] = sa[j];
        a_grp[p] = grp[j];
    }
}

// ---- doubling rounds: window bookkeeping -----------------------------------------------------------------
constexpr int SEG_TILE = 1024;             // a workgroup owns the groups that START in its SEG_TILE window
constexpr int SEG_SPAN = 2 * SEG_TILE;     // ... and therefore sees at most this many elements
constexpr int SEG_ITEMS = SEG_SPAN / TB;   // 8
constexpr int SEG_DBITS = 9;               // digit width of the LDS sort: (26-bit rank, 10-bit local group) = 36 bits = 4 passes
constexpr int SEG_DIGITS = 1 << SEG_DBITS;
static_assert(SEG_DIGITS == 2 * TB, "two digits per thread in the digit scan");

// lasthead[w] = 1 + (largest group-head index inside window w), 0 if the window has no head
__global__ __launch_bounds__(TB) void k_win_heads(const uint32_t *__restrict__ a_grp, uint32_t m, uint32_t *__restrict__ lasthead)
{
    __shared__ uint32_t sm[TB / 64 + 1];
    const uint32_t base = blockIdx.x * SEG_TILE;
    uint32_t best = 0;
#pragma unroll
    for (int k = 0; k < SEG_TILE / TB; k++) {
        uint32_t j = base + k * TB + threadIdx.x;
        if (j < m) {
            bool head = (j == 0) || (a_grp[j] != a_grp[j - 1]);
            if (head) best = j + 1;
        }
    }
    uint32_t tot;
    block_incl_scan<OpMax>(best, sm, &tot);
    if (threadIdx.x == 0) lasthead[blockIdx.x] = tot;
}

// the sort / re-rank of all groups of <= SEG_TILE elements, one window per workgroup
__global__ __launch_bounds__(TB) void k_seg_round(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, uint32_t m, uint32_t n,
                                                 uint32_t h, int key_bits, const uint32_t *__restrict__ winscan /* inclusive max-scan of lasthead */,
                                                 const uint32_t *__restrict__ ISA_cur, uint32_t *__restrict__ ISA_nxt, uint32_t *__restrict__ SA,
                                                 uint32_t *__restrict__ b_sa, uint32_t *__restrict__ b_grp, uint32_t *__restrict__ lflag, uint32_t *__restrict__ keep,
                                                 uint32_t *__restrict__ large_count)
{
    __shared__ uint32_t g[SEG_SPAN];          // group rank (abs SA position of the group head) per loaded element
    __shared__ uint32_t sv[SEG_SPAN];         // suffix index of the owned elements
    __shared__ uint32_t k2[SEG_SPAN];         // sort key of the owned elements
    __shared__ uint16_t gsl[SEG_SPAN];        // group start (local position) per loaded element, 0xFFFF = spill-in
    __shared__ uint16_t lgid[SEG_SPAN];       // local group id of the owned elements
    __shared__ uint16_t idxA[SEG_SPAN], idxB[SEG_SPAN];
    __shared__ uint32_t cnt[TB / 64][SEG_DIGITS];
    __shared__ uint32_t dbase[SEG_DIGITS];
    __shared__ uint32_t sm[TB / 64 + 1];
    __shared__ uint32_t s_fo, s_oe;

    const uint32_t base = blockIdx.x * SEG_TILE;
    const uint32_t avail = (m - base < (uint32_t)SEG_SPAN) ? m - base : (uint32_t)SEG_SPAN;
    const bool list_ends = (base + avail == m);
    const uint32_t wlen = avail < (uint32_t)SEG_TILE ? avail : (uint32_t)SEG_TILE;    // elements of my own window
    const int tid = threadIdx.x;

    for (uint32_t p = tid; p < avail; p += TB) g[p] = a_grp[base + p];
    const uint32_t gprev = (base > 0) ? a_grp[base - 1] : 0xFFFFFFFFu;
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
    __shared__ uint32_t fz[TB], rz[TB];
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
    if (tid == 0) { s_fo = 0xFFFFFFFFu; s_oe = 0; }
    __syncthreads();
    // ---- classify: size of the group of each position; owned = starts in my window and size <= SEG_TILE ----
    const uint32_t ph = (blockIdx.x > 0) ? winscan[blockIdx.x - 1] : 0u;          // 1 + last head before my window
    const uint32_t spill_start = ph > 0 ? ph - 1 : 0u;                              // global index
    uint32_t my_fo = 0xFFFFFFFFu, my_oe = 0, my_large = 0;
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
            if (p < wlen) { lflag[base + p] = large ? 1u : 0u; my_large += large ? 1u : 0u; }
            if (gs != 0xFFFFu && gs < (uint32_t)SEG_TILE && !large) {
                if (p < my_fo) my_fo = p;
                if (p + 1 > my_oe) my_oe = p + 1;
            }
        }
    }
    if (my_fo != 0xFFFFFFFFu) { atomicMin(&s_fo, my_fo); atomicMax(&s_oe, my_oe); }
    my_large = wave_sum(my_large);
    if (lane_id() == 0 && my_large) atomicAdd(large_count, my_large);
    __syncthreads();
    const uint32_t fo = s_fo, oe = s_oe;
    if (fo == 0xFFFFFFFFu) return;             // nothing owned
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
    for (uint32_t q = tid; q < no; q += TB) {
        const uint32_t s = a_sa[base + fo + q];
        sv[q] = s;
        const uint64_t s2 = (uint64_t)s + h;
        k2[q] = (s2 < n) ? ISA_cur[s2] + 1u : 0u;
        idxA[q] = (uint16_t)q;
    }
    __syncthreads();
    const uint32_t ngroups = (uint32_t)lgid[no - 1] + 1u;

    // ---- LSD radix sort of the index permutation by the composite key (lgid << key_bits) | key2, 9 bits per pass ----
    uint16_t *src = idxA, *dst = idxB;
    const int w = tid >> 6, l = tid & 63;
    const uint64_t lt = lanemask_lt();
    const int gbits = (ngroups > 1u) ? 32 - __clz((int)(ngroups - 1u)) : 0;
    const int npass = key_bits > 0 ? (key_bits + gbits + SEG_DBITS - 1) / SEG_DBITS : 0;
    // each wave ranks a contiguous quarter of the owned range: only ceil(no / 256) iterations of 64 are live
    const int nit = (int)((no + TB - 1) / TB);
    const uint32_t wspan = (uint32_t)nit * 64u;
    for (int pass = 0; pass < npass; pass++) {
        const int shift = SEG_DBITS * pass;
        const int part = (shift + SEG_DBITS <= key_bits) ? 0 : (shift >= key_bits ? 2 : 1);   // digit from key2 / both / group id
        for (int i = tid; i < (TB / 64) * SEG_DIGITS; i += TB) (&cnt[0][0])[i] = 0;
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
            else if (part == 2) d = (uint32_t)lgid[id] >> (shift - key_bits);
            else d = (k2[id] >> shift) | ((uint32_t)lgid[id] << (key_bits - shift));
            d = valid ? (d & (uint32_t)(SEG_DIGITS - 1)) : 0u;
            dg[it] = d | (id << SEG_DBITS);
            const uint64_t mm = match_any<SEG_DBITS>(d, valid);
            const uint32_t below = (uint32_t)__popcll(mm & lt);
            const uint32_t c = valid ? cnt[w][d] : 0u;
            rk[it] = c + below;
            if (valid && below == 0) cnt[w][d] = c + (uint32_t)__popcll(mm);
        }
        __syncthreads();
        {   // per digit: exclusive over waves, then exclusive over digits (two digits per thread, in digit order)
            uint32_t s2[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int d = 2 * tid + e;
                uint32_t s = 0;
#pragma unroll
                for (int k = 0; k < TB / 64; k++) { uint32_t t = cnt[k][d]; cnt[k][d] = s; s += t; }
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
#pragma unroll
    for (int k = 0; k < SEG_ITEMS; k++) {
        const uint32_t q = p0 + k;
        ap[k] = 0; nh[k] = 0;
        if (q < no) {
            const uint32_t id = src[q];
            const uint32_t pos = fo + q;                        // groups keep their positions through the sort
            ap[k] = g[pos] + (pos - (uint32_t)gsl[pos]);
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
    __shared__ uint8_t firstflag[TB + 1];
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
            const uint32_t s = sv[src[q]];
            ISA_nxt[s] = run;
            if (single) SA[ap[k]] = s;
            b_sa[base + fo + q] = s;
            b_grp[base + fo + q] = run | (single ? DONE : 0u);
            keep[base + fo + q] = single ? 0u : 1u;
        }
    }
}

// ---- large groups: compaction into a side list, global sort, write back ---------------------------------
__global__ __launch_bounds__(TB) void k_large_gather(const uint32_t *__restrict__ lflag, const uint32_t *__restrict__ lpos_scan,
                                                    const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, uint32_t m,
                                                    uint32_t *__restrict__ l_sa, uint32_t *__restrict__ l_grp, uint32_t *__restrict__ l_pos)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m || !lflag[j]) return;
    uint32_t p = lpos_scan[j];
    l_sa[p] = a_sa[j];
    l_grp[p] = a_grp[j];
    l_pos[p] = j;
}

// old-group head positions in the large list (input of the max scan that gives every element its group's head)
__global__ __launch_bounds__(TB) void k_large_heads(const uint32_t *__restrict__ a_grp, uint32_t m, uint32_t *__restrict__ hv)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    bool head = (j == 0) || (a_grp[j - 1] != a_grp[j]);
    hv[j] = head ? j : 0u;
}

// Sort key of a large-group element: (group id << 32) | rank of suffix + h.  Every group on this path has more than
// 1024 elements, so its head position in the list divided by 1024 is a dense, order-preserving id: the group digits
// need bits(list length / 1024) instead of bits(n) -- two radix passes instead of four on a 64 MiB block.
__global__ __launch_bounds__(TB) void k_make_keys(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ jhead, uint32_t m, uint32_t n,
                                                 uint32_t h, const uint32_t *__restrict__ ISA, uint64_t *__restrict__ keys,
                                                 uint32_t *__restrict__ vals)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    uint32_t s = a_sa[j];
    uint64_t s2 = (uint64_t)s + h;
    uint32_t k2 = (s2 < n) ? ISA[s2] + 1u : 0u;
    keys[j] = ((uint64_t)(jhead[j] >> 10) << 32) | k2;
    vals[j] = s;
}

// abs position of element j after the sort = group rank + offset inside the (old) group (groups keep their index
// ranges through the sort, so the pre-sort group array still applies); new head flag from the full 64-bit key;
// nh[j] = newhead ? abspos : 0 (input of the max scan).  grp and nh may alias.
__global__ __launch_bounds__(TB) void k_abspos(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ jhead, const uint32_t *grp, uint32_t m,
                                              uint32_t *__restrict__ abspos, uint32_t *nh)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    uint64_t k = keys[j];
    uint32_t ap = grp[j] + (j - jhead[j]);
    abspos[j] = ap;
    bool head = (j == 0) || (keys[j - 1] != k);
    nh[j] = head ? ap : 0u;
}

// new ranks -> ISA_nxt; singletons -> SA; sorted elements back to their slots of the active list
__global__ __launch_bounds__(TB) void k_large_finish(const uint32_t *__restrict__ abspos, const uint32_t *__restrict__ newrank,
                                                    const uint32_t *__restrict__ vals, const uint32_t *__restrict__ l_pos, uint32_t m,
                                                    uint32_t *__restrict__ ISA_nxt, uint32_t *__restrict__ SA, uint32_t *__restrict__ b_sa,
                                                    uint32_t *__restrict__ b_grp, uint32_t *__restrict__ keep)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    uint32_t ap = abspos[j], r = newrank[j], s = vals[j];
    ISA_nxt[s] = r;
    bool head = (r == ap);
    bool next_head = (j + 1 == m) || (newrank[j + 1] == abspos[j + 1]);
    bool single = head && next_head;
    if (single) SA[ap] = s;
    uint32_t p = l_pos[j];
    b_sa[p] = s;
    b_grp[p] = r | (single ? DONE : 0u);
    keep[p] = single ? 0u : 1u;
}

// compaction of a round's output; finished suffixes also get their final rank in the buffer that was READ this
// round (it becomes the write buffer of the next round and is never rewritten for them)
__global__ __launch_bounds__(TB) void k_compact_round(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos,
                                                     const uint32_t *__restrict__ b_sa, const uint32_t *__restrict__ b_grp, uint32_t m,
                                                     uint32_t *__restrict__ ISA_cur, uint32_t *__restrict__ a_sa, uint32_t *__restrict__ a_grp)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    const uint32_t gv = b_grp[j], s = b_sa[j];
    if (keep[j]) {
        const uint32_t p = pos[j];
        a_sa[p] = s;
        a_grp[p] = gv;
    } else ISA_cur[s] = gv & ~DONE;
}

// ---- BWT emission (bwt.cpp:44-61) -------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_bwt_gather(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA, const uint32_t *__restrict__ ISA,
                                                  uint32_t n, uint8_t *__restrict__ out)
{
    uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t idx = ISA[0];
    uint32_t s = SA[i];
    if (i == 0) out[0] = T[n - 1];
    if (i == idx) return;                       // the row whose predecessor is the sentinel is dropped
    uint32_t o = (i < idx) ? i + 1 : i;
    out[o] = T[s - 1];
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

struct SaBufs {
    uint64_t *keysA, *keysB;
    uint32_t *valsA, *valsB, *ISA0, *ISA1, *SA, *a_sa, *a_grp, *b_sa, *b_grp, *t1, *t2, *t3, *keep, *l_pos, *win, *scratch;
};

void sa_layout(Arena &a, size_t n, SaBufs &b, bool need_sa_buf)
{
    b.keysA = a.get<uint64_t>(n);
    b.keysB = a.get<uint64_t>(n);
    b.valsA = a.get<uint32_t>(n);
    b.valsB = a.get<uint32_t>(n);
    b.ISA0 = a.get<uint32_t>(n);
    b.ISA1 = a.get<uint32_t>(n);
    b.SA = need_sa_buf ? a.get<uint32_t>(n) : nullptr;
    b.a_sa = a.get<uint32_t>(n);
    b.a_grp = a.get<uint32_t>(n);
    b.b_sa = a.get<uint32_t>(n);
    b.b_grp = a.get<uint32_t>(n);
    b.t1 = a.get<uint32_t>(n);
    b.t2 = a.get<uint32_t>(n);
    b.t3 = a.get<uint32_t>(n);
    b.keep = a.get<uint32_t>(n);
    b.l_pos = a.get<uint32_t>(n);
    b.win = a.get<uint32_t>(n / SEG_TILE + 64);
    size_t sw = jpk_radix_scratch_words(n);
    size_t sc = jpk_scan_scratch_words(n);
    b.scratch = a.get<uint32_t>(sw > sc ? sw : sc);
}

// builds SA (uint32) for T[0..n); on return *isa_final points at the buffer holding the complete inverse SA
int build_sa(jpk_ctx *ctx, const uint8_t *T, uint32_t n, SaBufs &b, uint32_t **isa_final)
{
    hipStream_t st = ctx->stream;
    const unsigned g_n = jpk_grid(n, TB);
    ctx->stats.sa_rounds = 0;
    ctx->stats.sa_sorted_elems = 0;
    memset(ctx->stats.sa_round_active, 0, sizeof ctx->stats.sa_round_active);
    memset(ctx->stats.sa_round_large, 0, sizeof ctx->stats.sa_round_large);
    ctx->stats.sa_round_active[0] = (int32_t)n;

    // round 0: sort by the first 7 bytes (7 passes; ties keep descending text position)
    JPK_LAUNCH(ctx, PROF_SA_KEYS, n, k_init_keys, dim3(g_n), dim3(TB), T, n, b.keysA, b.valsA);
    {
        const int shifts[7] = {8, 16, 24, 32, 40, 48, 56};
        JPK_TRY(jpk_radix_sort_pairs_u64(ctx, b.keysA, b.valsA, b.keysB, b.valsB, n, shifts, 7, b.scratch));
        ctx->stats.sa_sorted_elems += n;
    }
    JPK_LAUNCH(ctx, PROF_SA_RERANK, n, k_heads_u64, dim3(g_n), dim3(TB), b.keysA, b.valsA, n, n, b.t1);
    JPK_TRY(jpk_inclusive_max_u32(ctx, b.t1, b.t2, n, b.scratch));                     // t2 = grp
    JPK_LAUNCH(ctx, PROF_SA_RERANK, n, k_round0_finish, dim3(g_n), dim3(TB), b.t2, b.valsA, n, b.ISA0, b.ISA1, b.SA, b.t1);  // t1 = keep
    JPK_TRY(jpk_exclusive_sum_u32(ctx, b.t1, b.t3, n, b.scratch, ctx->d_mail));       // t3 = pos
    JPK_LAUNCH(ctx, PROF_SA_RERANK, n, k_compact, dim3(g_n), dim3(TB), b.t1, b.t3, b.valsA, b.t2, n, b.a_sa, b.a_grp);
    uint32_t m = 0;
    JPK_TRY(jpk_read_mail(ctx, &m, 1));
    ctx->stats.sa_rounds = 1;

    const int kbits = jpk_bits_for(n);     // key2 <= n, group rank < n
    const int key_passes = getenv("JPK_DBG_NOSORT") ? 0 : (kbits + 7) / 8;   // debug: time k_seg_round without its LDS sort
    uint32_t *isa_cur = b.ISA0, *isa_nxt = b.ISA1;
    uint64_t h = 7;
    while (m > 0) {
        if (h >= n) return JPK_E_DEVICE;   // cannot happen: every suffix is unique once h >= n
        const unsigned g_m = jpk_grid(m, TB);
        const unsigned nwin = jpk_grid(m, SEG_TILE);
        // window bookkeeping: start of the group that spills into each window
        JPK_LAUNCH(ctx, PROF_SA_KEYS, m, k_win_heads, dim3(nwin), dim3(TB), b.a_grp, m, b.win);
        JPK_TRY(jpk_inclusive_max_u32(ctx, b.win, b.win, nwin, b.scratch));
        JPK_HIP(hipMemsetAsync(ctx->d_mail + 4, 0, 4, st));
        JPK_LAUNCH(ctx, PROF_SA_SEG, m, k_seg_round, dim3(nwin), dim3(TB), b.a_sa, b.a_grp, m, n, (uint32_t)h, key_passes ? kbits : 0, b.win, isa_cur, isa_nxt,
                   b.SA, b.b_sa, b.b_grp, b.t1, b.keep, ctx->d_mail + 4);              // t1 = lflag
        uint32_t mailw[5];
        JPK_TRY(jpk_read_mail(ctx, mailw, 5));
        const uint32_t lc = mailw[4];
        if (ctx->stats.sa_rounds < JPK_SA_MAX_ROUNDS) {
            ctx->stats.sa_round_active[ctx->stats.sa_rounds] = (int32_t)m;
            ctx->stats.sa_round_large[ctx->stats.sa_rounds] = (int32_t)lc;
        }
        if (lc > 0) {
            JPK_TRY(jpk_exclusive_sum_u32(ctx, b.t1, b.t2, m, b.scratch, nullptr));      // t2 = position in the large list
            const unsigned g_l = jpk_grid(lc, TB);
            uint32_t *l_sa = b.valsB, *l_grp = b.t3;
            JPK_LAUNCH(ctx, PROF_SA_KEYS, lc, k_large_gather, dim3(g_m), dim3(TB), b.t1, b.t2, b.a_sa, b.a_grp, m, l_sa, l_grp, b.l_pos);
            JPK_LAUNCH(ctx, PROF_SA_KEYS, lc, k_large_heads, dim3(g_l), dim3(TB), l_grp, lc, b.t1);
            JPK_TRY(jpk_inclusive_max_u32(ctx, b.t1, b.t2, lc, b.scratch));               // t2 = jhead (old groups)
            JPK_LAUNCH(ctx, PROF_SA_KEYS, lc, k_make_keys, dim3(g_l), dim3(TB), l_sa, b.t2, lc, n, (uint32_t)h, isa_cur, b.keysA, b.valsA);
            int lshifts[8];
            int lns = 0;
            for (int s = 0; s < kbits && lns < key_passes; s += 8) lshifts[lns++] = s;
            const int gbits = jpk_bits_for(lc >> 10);
            for (int s = 0; s < gbits; s += 8) lshifts[lns++] = 32 + s;
            JPK_TRY(jpk_radix_sort_pairs_u64(ctx, b.keysA, b.valsA, b.keysB, b.valsB, lc, lshifts, lns, b.scratch));
            ctx->stats.sa_sorted_elems += lc;
            JPK_LAUNCH(ctx, PROF_SA_RERANK, lc, k_abspos, dim3(g_l), dim3(TB), b.keysA, b.t2, l_grp, lc, b.t1, b.t3);  // t1 = abspos, t3 = nh (over l_grp)
            JPK_TRY(jpk_inclusive_max_u32(ctx, b.t3, b.t2, lc, b.scratch));               // t2 = newrank
            JPK_LAUNCH(ctx, PROF_SA_RERANK, lc, k_large_finish, dim3(g_l), dim3(TB), b.t1, b.t2, b.valsA, b.l_pos, lc, isa_nxt, b.SA, b.b_sa, b.b_grp,
                       b.keep);
        }
        JPK_TRY(jpk_exclusive_sum_u32(ctx, b.keep, b.t3, m, b.scratch, ctx->d_mail));     // t3 = pos
        JPK_LAUNCH(ctx, PROF_SA_RERANK, m, k_compact_round, dim3(g_m), dim3(TB), b.keep, b.t3, b.b_sa, b.b_grp, m, isa_cur, b.a_sa, b.a_grp);
        uint32_t m2 = 0;
        JPK_TRY(jpk_read_mail(ctx, &m2, 1));
        m = m2;
        h <<= 1;
        uint32_t *t = isa_cur; isa_cur = isa_nxt; isa_nxt = t;
        ctx->stats.sa_rounds++;
    }
    JPK_HIP(hipGetLastError());
    *isa_final = isa_cur;                  // after the last swap isa_cur holds the ranks written by the last round
    return JPK_OK;
}

}  // namespace

int jpk_suffix_array_device(jpk_ctx *ctx, const uint8_t *d_t, int32_t n, int32_t *d_sa)
{
    if (n <= 0) return JPK_OK;
    SaBufs b;
    Arena plan(ctx, true);
    sa_layout(plan, (size_t)n, b, false);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    sa_layout(real, (size_t)n, b, false);
    b.SA = reinterpret_cast<uint32_t *>(d_sa);
    uint32_t *isa;
    return build_sa(ctx, d_t, (uint32_t)n, b, &isa);
}

int jpk_fwd_bwt_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out)
{
    const int32_t rem = len % JPK_BWT_UNITS, nlen = len - rem;
    if (nlen <= 0) {
        // bwt.cpp:29-35: only the raw tail is produced; the 480 trailer bytes are left untouched
        if (rem > 0) JPK_HIP(hipMemcpyAsync(d_out, d_in, (size_t)rem, hipMemcpyDeviceToDevice, ctx->stream));
        return JPK_OK;
    }
    SaBufs b;
    Arena plan(ctx, true);
    sa_layout(plan, (size_t)nlen, b, true);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    sa_layout(real, (size_t)nlen, b, true);
    uint32_t *isa;
    JPK_TRY(build_sa(ctx, d_in, (uint32_t)nlen, b, &isa));
    JPK_LAUNCH(ctx, PROF_BWT_GATHER, nlen, k_bwt_gather, dim3(jpk_grid(nlen, TB)), dim3(TB), d_in, b.SA, isa, (uint32_t)nlen, d_out);
    hipLaunchKernelGGL(k_bwt_trailer, dim3(1), dim3(128), 0, ctx->stream, d_in, isa, (uint32_t)nlen, (uint32_t)len, d_out);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}
