// bwt_inv.hip -- inverse BWT on gfx950, replacing BlockSort::Bwt::InverseBwt (bwt.cpp:72-282) and its
// CUDAInverse<<<40,3>>> kernel (bwt.cpp:8-19), which walks only 120 chains.
//
// The reference builds Map[] by a stable counting sort of the BWT image and then follows
//     p = Map[p-1];  T[i] = Bwt[p - (p >= idx)]            (idx = trailer[0])
// from p = idx.  Two facts make this massively parallel without changing a byte of T:
//   * T[i] is the symbol whose bucket [cum[c], cum[c+1]) contains j = p_old - 1, so the walk needs ONE random
//     4-byte gather per output byte (nxt[j] = Map[j] - 1) and no gather from the BWT image at all;
//   * the chain j -> nxt[j] is a single linked list over all n positions, so it can be list-ranked: ~n/64
//     pseudo-random splitters each walk to the next splitter writing their bytes to a private scratch slot,
//     the slots are ranked by pointer jumping, and a copy pass places every slot at its text offset.
// Kernels: tile histograms (LDS 256 bins) -> scan -> stable scatter (wave match-any) -> walk -> rank -> copy.
#include "common.hpp"
#include "prims.hpp"

using namespace jpk;

namespace {

constexpr int TB = 256;
constexpr int WAVES = TB / 64;
constexpr int ITEMS = 32;                  // bytes per thread in the Map-build tiles
constexpr int TILE = TB * ITEMS;           // 8192 bytes
constexpr int STRIDE = 64;                 // one splitter per 64 positions of j-space
constexpr int CAP = 256;                   // scratch bytes per slot (4x the mean sub-list length)
constexpr uint32_t NIL = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t splitter_of(uint32_t k) { return k * STRIDE + (mix32(k) & (STRIDE - 1)); }
__device__ __forceinline__ bool is_splitter(uint32_t j) { return (j & (STRIDE - 1)) == (mix32(j / STRIDE) & (STRIDE - 1)); }

// Device-resident bookkeeping of one inverse BWT: nothing of it is needed on the host before the call's last kernel, so the
// whole transform is enqueued without a host round trip (a batch of blocks: one status read at the very end).
struct InvState {
    uint32_t status;        // 0 or JPK_E_CORRUPT (as uint32)
    uint32_t I;             // trailer[0] = ISA[0] + 1 (bwt.cpp:57-61 / :134); 0 = invalid: every later kernel leaves at once
    uint32_t novf;          // overflow slots the walk has chained
    uint32_t head_dist;     // bytes the chain from the head covers (must be n)
};

__device__ __forceinline__ void inv_prep_body(const uint8_t *__restrict__ trailer, uint32_t n, InvState *__restrict__ st)
{
    const uint32_t I = (uint32_t)trailer[0] | ((uint32_t)trailer[1] << 8) | ((uint32_t)trailer[2] << 16) | ((uint32_t)trailer[3] << 24);
    const bool ok = I >= 1u && I <= n;
    st->status = ok ? 0u : (uint32_t)JPK_E_CORRUPT;
    st->I = ok ? I : 0u;
    st->novf = 0;
    st->head_dist = 0;
}
__global__ void k_inv_prep(const uint8_t *__restrict__ trailer, uint32_t n, InvState *__restrict__ st)
{
    if (threadIdx.x == 0) inv_prep_body(trailer, n, st);
}

// ---- Map build ---------------------------------------------------------------------------------------
__device__ __forceinline__ void hist_body(const uint8_t *__restrict__ B, uint32_t n, uint32_t *__restrict__ tilehist, uint32_t ntiles, uint32_t bx)
{
    __shared__ uint32_t h[WAVES][256];
    for (int i = threadIdx.x; i < WAVES * 256; i += TB) (&h[0][0])[i] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t base = (size_t)bx * TILE + (size_t)w * (64 * ITEMS) + l;
    // loads first, all in flight (slots past the end re-read the last byte and are masked): a load inside `if (i < n)` makes the
    // compiler wait for each one in turn
    uint8_t sy[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const size_t i = base + (size_t)it * 64;
        sy[it] = B[i < n ? i : n - 1];
    }
    // counted by wave match, not LDS atomics: a BWT image is runs of equal bytes, i.e. most lanes of a wave hit one bin
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const bool valid = base + (size_t)it * 64 < n;
        const uint64_t m = match_any8(sy[it], valid);
        if (valid && (m & lt) == 0ull) h[w][sy[it]] += (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += TB)
        tilehist[(size_t)d * ntiles + bx] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
}
__global__ __launch_bounds__(TB) void k_hist(const uint8_t *__restrict__ B, uint32_t n, uint32_t *__restrict__ tilehist, uint32_t ntiles)
{
    hist_body(B, n, tilehist, ntiles, blockIdx.x);
}

// cum[c] = first F-position of symbol c (= scanned table entry of tile 0), cum[256] = n
// (pos_base: what the scanned table holds in front of this block's entries -- 0 for one block, the bytes of the blocks before it in a batch)
__device__ __forceinline__ void cum_body(const uint32_t *__restrict__ tileoff, uint32_t ntiles, uint32_t n, uint32_t *__restrict__ cum, uint32_t pos_base)
{
    uint32_t c = threadIdx.x;
    if (c < 256) cum[c] = tileoff[(size_t)c * ntiles] - pos_base;
    if (c == 0) cum[256] = n;
}
__global__ void k_cum(const uint32_t *__restrict__ tileoff, uint32_t ntiles, uint32_t n, uint32_t *__restrict__ cum)
{
    cum_body(tileoff, ntiles, n, cum, 0u);
}

// nxt[F-position] = (i < I ? i : i + 1) - 1     (bwt.cpp:171-174, minus one so that -1 terminates the chain)
template <bool DEV>
__device__ __forceinline__ void build_nxt_body(const uint8_t *__restrict__ B, uint32_t n, uint32_t I_host, const InvState *__restrict__ st,
                                               const uint32_t *__restrict__ tileoff, uint32_t ntiles, int32_t *__restrict__ nxt, uint32_t bx, uint32_t pos_base)
{
    __shared__ uint32_t cnt[WAVES][256];
    __shared__ uint32_t gbase[256];
    const uint32_t I = DEV ? st->I : I_host;
    if (DEV && I == 0u) return;
    for (int i = threadIdx.x; i < WAVES * 256; i += TB) (&cnt[0][0])[i] = 0;
    for (int d = threadIdx.x; d < 256; d += TB) gbase[d] = tileoff[(size_t)d * ntiles + bx] - pos_base;
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t base = (size_t)bx * TILE + (size_t)w * (64 * ITEMS) + l;
    const uint64_t lt = lanemask_lt();
    uint8_t sym[ITEMS];
    uint16_t rnk[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {              // all loads in flight before the first match
        const size_t i = base + (size_t)it * 64;
        sym[it] = B[i < n ? i : n - 1];
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        size_t i = base + (size_t)it * 64;
        bool valid = i < n;
        uint32_t d = valid ? sym[it] : 0;
        sym[it] = (uint8_t)d;
        uint64_t m = match_any8(d, valid);
        uint32_t below = (uint32_t)__popcll(m & lt);
        uint32_t c = valid ? cnt[w][d] : 0;
        rnk[it] = (uint16_t)(c + below);
        if (valid && below == 0) cnt[w][d] = c + (uint32_t)__popcll(m);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += TB) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < WAVES; k++) { uint32_t t = cnt[k][d]; cnt[k][d] = s; s += t; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        size_t i = base + (size_t)it * 64;
        if (i < n) {
            uint32_t d = sym[it];
            size_t dst = (size_t)gbase[d] + cnt[w][d] + rnk[it];
            uint32_t ii = (uint32_t)i;
            nxt[dst] = (ii < I) ? (int32_t)ii - 1 : (int32_t)ii;
        }
    }
}
template <bool DEV>
__global__ __launch_bounds__(TB) void k_build_nxt(const uint8_t *__restrict__ B, uint32_t n, uint32_t I_host, const InvState *__restrict__ st,
                                                 const uint32_t *__restrict__ tileoff, uint32_t ntiles, int32_t *__restrict__ nxt)
{
    build_nxt_body<DEV>(B, n, I_host, st, tileoff, ntiles, nxt, blockIdx.x, 0u);
}

// ---- walk ----------------------------------------------------------------------------------------------
// One lane per splitter (+ one for the chain head j0 = I-1).  Lane k walks from its splitter to the next
// one, turning every visited j into its symbol via cum[] (LDS, with a 4096-entry coarse LUT) and writing
// the symbols 16 bytes at a time to scratch[slot*CAP ...].  A sub-list longer than CAP continues in a
// freshly allocated overflow slot, so every slot holds <= CAP bytes.
constexpr int LUT = 4096;

__device__ __forceinline__ void walk_body(const int32_t *__restrict__ nxt, const uint32_t *__restrict__ cum_g, uint32_t n, InvState *__restrict__ st,
                                          uint32_t nsplit, uint32_t lut_shift, uint8_t *__restrict__ scratch,
                                          uint32_t *__restrict__ slot_len, uint32_t *__restrict__ slot_next, uint32_t max_slots, uint32_t bx)
{
    __shared__ uint32_t cum[257];
    __shared__ uint8_t lut[LUT];
    const uint32_t I = st->I;
    if (I == 0u) return;
    uint32_t *const ovf_counter = &st->novf;
    for (int i = threadIdx.x; i < 257; i += TB) cum[i] = cum_g[i];
    __syncthreads();
    for (int q = threadIdx.x; q < LUT; q += TB) {
        uint64_t pos = (uint64_t)q << lut_shift;
        int lo = 0, hi = 256;                 // largest c with cum[c] <= pos
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if ((uint64_t)cum[mid] <= pos) lo = mid; else hi = mid;
        }
        lut[q] = (uint8_t)lo;
    }
    __syncthreads();

    const uint32_t k = bx * TB + threadIdx.x;
    if (k > nsplit) return;
    const uint32_t j0 = I - 1;
    uint32_t j;
    if (k == nsplit) j = j0;
    else {
        j = splitter_of(k);
        if (j >= n || j == j0) {              // splitter outside the block, or the head (walked by lane nsplit)
            slot_len[k] = 0;
            slot_next[k] = NIL;
            return;
        }
    }
    uint32_t slot = k, len = 0, steps = 0;
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    uint8_t *dst = scratch + (size_t)slot * CAP;
    for (;;) {
        uint32_t c = lut[j >> lut_shift];
        while (cum[c + 1] <= j) c++;
        const uint32_t sh = (len & 3u) * 8u, wi = (len >> 2) & 3u;
        const uint32_t cv = c << sh;
        if (wi == 0) w0 |= cv; else if (wi == 1) w1 |= cv; else if (wi == 2) w2 |= cv; else w3 |= cv;
        len++;
        if ((len & 15u) == 0) {
            *reinterpret_cast<uint4 *>(dst + len - 16) = make_uint4(w0, w1, w2, w3);
            w0 = w1 = w2 = w3 = 0;
        }
        const int32_t nx = nxt[j];
        uint32_t link = NIL;
        bool stop = false;
        if (nx < 0 || ++steps > n) stop = true;      // steps > n: only on a corrupt (non-injective) map
        else if (is_splitter((uint32_t)nx)) { stop = true; link = (uint32_t)nx / STRIDE; }
        if (stop) {
            if (len & 15u) *reinterpret_cast<uint4 *>(dst + (len & ~15u)) = make_uint4(w0, w1, w2, w3);
            slot_len[slot] = len;
            slot_next[slot] = link;
            return;
        }
        if (len == CAP) {                       // overflow: chain a new slot
            uint32_t ns = nsplit + 1 + atomicAdd(ovf_counter, 1u);
            if (ns >= max_slots) { slot_len[slot] = len; slot_next[slot] = NIL; return; }   // corrupt input guard
            slot_len[slot] = len;
            slot_next[slot] = ns;
            slot = ns; len = 0;
            dst = scratch + (size_t)slot * CAP;
        }
        j = (uint32_t)nx;
    }
}
__global__ __launch_bounds__(TB) void k_walk(const int32_t *__restrict__ nxt, const uint32_t *__restrict__ cum_g, uint32_t n, InvState *__restrict__ st,
                                            uint32_t nsplit, uint32_t lut_shift, uint8_t *__restrict__ scratch,
                                            uint32_t *__restrict__ slot_len, uint32_t *__restrict__ slot_next, uint32_t max_slots)
{
    walk_body(nxt, cum_g, n, st, nsplit, lut_shift, scratch, slot_len, slot_next, max_slots, blockIdx.x);
}

// ---- list ranking of the slots (Wyllie pointer jumping, ping-pong buffers) ----------------------------
// dist[s] = bytes from the start of slot s to the end of the text.
// slots in use: the splitters, the head and the overflow slots the walk chained (known on the device only; grids are launched
// for the upper bound max_slots)
__device__ __forceinline__ uint32_t slots_in_use(const InvState *st, uint32_t nsplit, uint32_t max_slots)
{
    if (st->I == 0u) return 0u;
    const uint32_t ns = nsplit + 1u + st->novf;
    return ns < max_slots ? ns : max_slots;
}

__device__ __forceinline__ void rank_init_body(const uint32_t *__restrict__ slot_len, const uint32_t *__restrict__ slot_next, const InvState *__restrict__ st,
                                               uint32_t nsplit, uint32_t max_slots, uint32_t *__restrict__ dist, uint32_t *__restrict__ link, uint32_t bx)
{
    const uint32_t nslots = slots_in_use(st, nsplit, max_slots);
    uint32_t s = bx * TB + threadIdx.x;
    if (s >= nslots) return;
    dist[s] = slot_len[s];
    link[s] = slot_next[s];
}
__device__ __forceinline__ void rank_jump_body(const uint32_t *__restrict__ dist_in, const uint32_t *__restrict__ link_in, const InvState *__restrict__ st,
                                               uint32_t nsplit, uint32_t max_slots, uint32_t *__restrict__ dist_out, uint32_t *__restrict__ link_out, uint32_t bx)
{
    const uint32_t nslots = slots_in_use(st, nsplit, max_slots);
    uint32_t s = bx * TB + threadIdx.x;
    if (s >= nslots) return;
    uint32_t d = dist_in[s], l = link_in[s];
    if (l != NIL) { d += dist_in[l]; l = link_in[l]; }
    dist_out[s] = d;
    link_out[s] = l;
}
__global__ __launch_bounds__(TB) void k_rank_init(const uint32_t *__restrict__ slot_len, const uint32_t *__restrict__ slot_next, const InvState *__restrict__ st,
                                                 uint32_t nsplit, uint32_t max_slots, uint32_t *__restrict__ dist, uint32_t *__restrict__ link)
{
    rank_init_body(slot_len, slot_next, st, nsplit, max_slots, dist, link, blockIdx.x);
}
__global__ __launch_bounds__(TB) void k_rank_jump(const uint32_t *__restrict__ dist_in, const uint32_t *__restrict__ link_in, const InvState *__restrict__ st,
                                                 uint32_t nsplit, uint32_t max_slots, uint32_t *__restrict__ dist_out, uint32_t *__restrict__ link_out)
{
    rank_jump_body(dist_in, link_in, st, nsplit, max_slots, dist_out, link_out, blockIdx.x);
}

// one wave per 4 slots (16 lanes each): T[n - dist[s] ...] = scratch[s*CAP ... + len)
__device__ __forceinline__ void copy_out_body(const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ slot_len,
                                              const uint32_t *__restrict__ dist, const InvState *__restrict__ st, uint32_t nsplit, uint32_t max_slots,
                                              uint32_t n, uint8_t *__restrict__ T, uint32_t bx)
{
    const uint32_t nslots = slots_in_use(st, nsplit, max_slots);
    const uint32_t gid = bx * TB + threadIdx.x;
    const uint32_t s = gid >> 4, sub = gid & 15u;
    if (s >= nslots) return;
    const uint32_t len = slot_len[s];
    if (len == 0) return;
    const uint32_t d = dist[s];
    if (d > n || d < len) return;              // corrupt chain; reported by the host-side check of the head slot
    const uint8_t *src = scratch + (size_t)s * CAP;
    uint8_t *dst = T + (n - d);
    for (uint32_t b = sub; b < len; b += 16) dst[b] = src[b];
}
__global__ __launch_bounds__(TB) void k_copy_out(const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ slot_len,
                                                const uint32_t *__restrict__ dist, const InvState *__restrict__ st, uint32_t nsplit, uint32_t max_slots,
                                                uint32_t n, uint8_t *__restrict__ T)
{
    copy_out_body(scratch, slot_len, dist, st, nsplit, max_slots, n, T, blockIdx.x);
}

__global__ void k_inv_tail(const uint8_t *__restrict__ B, uint32_t n, uint32_t len, uint8_t *__restrict__ T)
{
    uint32_t t = threadIdx.x;
    if (t < len - n) T[n + t] = B[n + t];
}

// the chain from trailer[0] must cover the whole block; more overflow slots than a valid block can need = corrupt as well
__global__ void k_head_check(const uint32_t *__restrict__ dist, uint32_t head_slot, uint32_t n, uint32_t max_slots, InvState *__restrict__ st)
{
    if (st->I == 0u) return;
    const uint32_t d = dist[head_slot];
    st->head_dist = d;
    if (d != n || head_slot + 1u + st->novf > max_slots) st->status = (uint32_t)JPK_E_CORRUPT;
}

// ---- comparator: the reference's own GPU kernel shape (CUDAInverse<<<40,3>>>, bwt.cpp:8-19, 226-229) ----------------
// 120 threads, one per stored index, each following p = Map[p-1]; T[k*step + i] = Bwt[p - (p >= idx)] for step = n/120
// dependent iterations.  Not used by any product path: it is the measured baseline next to k_walk (bench.py extras), the
// number behind "replaced, not hipified".
__global__ void k_chase120(const int32_t *__restrict__ nxt, const uint8_t *__restrict__ B, uint32_t n, uint32_t len, uint32_t idx, uint8_t *__restrict__ T)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= JPK_BWT_UNITS) return;
    const uint32_t step = n / JPK_BWT_UNITS;
    const uint8_t *tr = B + len + 4 * k;
    uint32_t p = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    uint8_t *dst = T + (size_t)k * step;
    for (uint32_t i = 0; i < step; i++) {
        if (p < 1 || p > n) return;                      // corrupt index: stop (the probe is only meaningful on valid images)
        p = (uint32_t)nxt[p - 1] + 1u;                   // Map[p-1]; 0 on the very last step of the last chain (row 0 = T[n-1])
        if (p > n) return;
        dst[i] = B[p - (p >= idx ? 1u : 0u)];
    }
}

struct InvBufs {
    uint32_t *tilehist, *scan_scratch, *cum, *slot_len, *slot_next, *distA, *distB, *linkA, *linkB;
    int32_t *nxt;
    uint8_t *scratch;
    InvState *state;
};

void inv_layout(Arena &a, size_t n, InvBufs &b, size_t &ntiles, size_t &nsplit, size_t &max_slots)
{
    ntiles = (n + TILE - 1) / TILE;
    nsplit = (n + STRIDE - 1) / STRIDE;
    max_slots = nsplit + 1 + n / CAP + 2;
    b.tilehist = a.get<uint32_t>(256 * ntiles);
    b.scan_scratch = a.get<uint32_t>(jpk_scan_scratch_words(256 * ntiles));
    b.cum = a.get<uint32_t>(260);
    b.nxt = a.get<int32_t>(n);
    b.slot_len = a.get<uint32_t>(max_slots);
    b.slot_next = a.get<uint32_t>(max_slots);
    b.distA = a.get<uint32_t>(max_slots);
    b.distB = a.get<uint32_t>(max_slots);
    b.linkA = a.get<uint32_t>(max_slots);
    b.linkB = a.get<uint32_t>(max_slots);
    b.scratch = a.get<uint8_t>(max_slots * CAP);
    b.state = a.get<InvState>(1);
}

// ---- the same kernels over MANY blocks at once (jpk_inv_bwt_batch_enqueue) ------------------------------------------------------------
// A stream of the reference's smallest blocks is decompressed 256 blocks to a call, and an inverse BWT is ~30 launches whatever its size:
// 7 700 launches, 40 ms of host time, for kernels that together run for 5.  Here every launch covers all blocks: blockIdx.y = the job, a
// table of per-job pointers and sizes, grids as wide as the largest job needs (blocks beyond a job's own extent leave at once).  The tile
// histograms of all jobs are ONE array and ONE scan; what the scan leaves in front of a job's entries is the sum of the blocks before it
// (pos_base, known on the host).  Each job has its own InvState, so a corrupt block stops only itself.
struct InvJob {
    const uint8_t *B;
    uint8_t *T;
    uint32_t *verdict;
    uint32_t n, len, ntiles, nsplit, max_slots, lut_shift, pos_base, pad;
    uint32_t *tilehist, *cum, *slot_len, *slot_next, *distA, *distB, *linkA, *linkB;
    int32_t *nxt;
    uint8_t *scratch;
    InvState *state;
};
__global__ void k_b_prep(const InvJob *__restrict__ jobs, uint32_t njobs)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < njobs) inv_prep_body(jobs[q].B + jobs[q].len, jobs[q].n, jobs[q].state);
}
__global__ __launch_bounds__(TB) void k_b_hist(const InvJob *__restrict__ jobs)
{
    const InvJob &j = jobs[blockIdx.y];
    if (blockIdx.x >= j.ntiles) return;
    hist_body(j.B, j.n, j.tilehist, j.ntiles, blockIdx.x);
}
__global__ void k_b_cum(const InvJob *__restrict__ jobs)
{
    const InvJob &j = jobs[blockIdx.x];
    cum_body(j.tilehist, j.ntiles, j.n, j.cum, j.pos_base);
}
__global__ __launch_bounds__(TB) void k_b_build_nxt(const InvJob *__restrict__ jobs)
{
    const InvJob &j = jobs[blockIdx.y];
    if (blockIdx.x >= j.ntiles) return;
    build_nxt_body<true>(j.B, j.n, 0u, j.state, j.tilehist, j.ntiles, j.nxt, blockIdx.x, j.pos_base);
}
__global__ __launch_bounds__(TB) void k_b_walk(const InvJob *__restrict__ jobs)
{
    const InvJob &j = jobs[blockIdx.y];
    if (blockIdx.x * TB > j.nsplit) return;
    walk_body(j.nxt, j.cum, j.n, j.state, j.nsplit, j.lut_shift, j.scratch, j.slot_len, j.slot_next, j.max_slots, blockIdx.x);
}
__global__ __launch_bounds__(TB) void k_b_rank_init(const InvJob *__restrict__ jobs)
{
    const InvJob &j = jobs[blockIdx.y];
    if (blockIdx.x * TB >= j.max_slots) return;
    rank_init_body(j.slot_len, j.slot_next, j.state, j.nsplit, j.max_slots, j.distA, j.linkA, blockIdx.x);
}
__global__ __launch_bounds__(TB) void k_b_rank_jump(const InvJob *__restrict__ jobs, int flip)
{
    const InvJob &j = jobs[blockIdx.y];
    if (blockIdx.x * TB >= j.max_slots) return;
    if (flip) rank_jump_body(j.distB, j.linkB, j.state, j.nsplit, j.max_slots, j.distA, j.linkA, blockIdx.x);
    else rank_jump_body(j.distA, j.linkA, j.state, j.nsplit, j.max_slots, j.distB, j.linkB, blockIdx.x);
}
// head check, the raw tail of the image (len % 120 bytes) and the verdict: one small workgroup per job
__global__ void k_b_finish(const InvJob *__restrict__ jobs, int final_in_b)
{
    const InvJob &j = jobs[blockIdx.x];
    const uint32_t *dist = final_in_b ? j.distB : j.distA;
    InvState *st = j.state;
    if (threadIdx.x < j.len - j.n) j.T[j.n + threadIdx.x] = j.B[j.n + threadIdx.x];
    if (threadIdx.x != 0) return;
    if (st->I != 0u) {
        const uint32_t d = dist[j.nsplit];
        st->head_dist = d;
        if (d != j.n || j.nsplit + 1u + st->novf > j.max_slots) st->status = (uint32_t)JPK_E_CORRUPT;
    }
    j.verdict[0] = st->status; j.verdict[1] = st->I; j.verdict[2] = st->novf; j.verdict[3] = st->head_dist;
}
__global__ __launch_bounds__(TB) void k_b_copy_out(const InvJob *__restrict__ jobs, int final_in_b)
{
    const InvJob &j = jobs[blockIdx.y];
    if (blockIdx.x * (TB / 16) >= j.max_slots) return;
    copy_out_body(j.scratch, j.slot_len, final_in_b ? j.distB : j.distA, j.state, j.nsplit, j.max_slots, j.n, j.T, blockIdx.x);
}

// what one job takes from the arena (everything of inv_layout but the tile table and the scan scratch, which the batch shares)
void inv_job_layout(Arena &a, size_t n, InvBufs &b, size_t &ntiles, size_t &nsplit, size_t &max_slots)
{
    ntiles = (n + TILE - 1) / TILE;
    nsplit = (n + STRIDE - 1) / STRIDE;
    max_slots = nsplit + 1 + n / CAP + 2;
    b.cum = a.get<uint32_t>(260);
    b.nxt = a.get<int32_t>(n);
    b.slot_len = a.get<uint32_t>(max_slots);
    b.slot_next = a.get<uint32_t>(max_slots);
    b.distA = a.get<uint32_t>(max_slots);
    b.distB = a.get<uint32_t>(max_slots);
    b.linkA = a.get<uint32_t>(max_slots);
    b.linkB = a.get<uint32_t>(max_slots);
    b.scratch = a.get<uint8_t>(max_slots * CAP);
    b.state = a.get<InvState>(1);
}

}  // namespace

// arena bytes one inverse BWT of n sorted bytes takes from the start of the arena (the batch decoder parks its own buffers behind)
size_t jpk_inv_bwt_arena_bytes(uint32_t n)
{
    InvBufs b;
    size_t ntiles, nsplit, max_slots;
    jpk_ctx dummy;
    Arena plan(&dummy, true);
    inv_layout(plan, n ? n : 1, b, ntiles, nsplit, max_slots);
    return plan.need + (1u << 20);
}

// Enqueues one inverse BWT on ctx->stream without any host round trip: the trailer index is read and validated on the device,
// grids are launched for the upper bound of the slot count, and the verdict (InvState) is copied to d_verdict[0..4) -- device
// memory of the caller that outlives the arena scratch -- by the last kernel.  The scratch starts at ctx->arena_base (0 outside
// a batch), so consecutive calls on one stream share it (in-order stream); a batch that runs several inverse BWTs side by side
// gives each lane its own base and its own stream (jpk_dev_blocks_decompress).
__global__ void k_inv_verdict(const InvState *__restrict__ st, uint32_t *__restrict__ out)
{
    if (threadIdx.x == 0) { out[0] = st->status; out[1] = st->I; out[2] = st->novf; out[3] = st->head_dist; }
}

int jpk_inv_bwt_enqueue(jpk_ctx *ctx, const uint8_t *d_in, int32_t len_with_trailer, uint8_t *d_out, uint32_t *d_verdict)
{
    hipStream_t st = ctx->stream;
    const int32_t len = len_with_trailer - JPK_TRAILER_BYTES;
    if (len < 0) return JPK_E_CORRUPT;
    const int32_t rem = len % JPK_BWT_UNITS;
    const uint32_t n = (uint32_t)(len - rem);
    if (n == 0) {
        if (rem > 0) JPK_HIP(hipMemcpyAsync(d_out, d_in, (size_t)rem, hipMemcpyDeviceToDevice, st));
        JPK_HIP(hipMemsetAsync(d_verdict, 0, 16, st));
        return JPK_OK;
    }
    InvBufs b;
    size_t ntiles, nsplit, max_slots;
    Arena plan(ctx, true);
    inv_layout(plan, n, b, ntiles, nsplit, max_slots);
    if (!jpk_arena_fits(ctx, ctx->arena_base + plan.need)) {
        if (ctx->arena_base) return JPK_E_ALLOC;                 // a batch has sized the arena and keeps buffers in it: it must not move
        JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    }
    Arena real(ctx, false);
    inv_layout(real, n, b, ntiles, nsplit, max_slots);

    hipLaunchKernelGGL(k_inv_prep, dim3(1), dim3(64), 0, st, d_in + len, n, b.state);
    JPK_LAUNCH(ctx, PROF_INV_HIST, n, k_hist, dim3((unsigned)ntiles), dim3(TB), d_in, n, b.tilehist, (uint32_t)ntiles);
    JPK_TRY(jpk_exclusive_sum_u32(ctx, b.tilehist, b.tilehist, 256 * ntiles, b.scan_scratch, nullptr));
    JPK_LAUNCH(ctx, PROF_INV_HIST, n, k_cum, dim3(1), dim3(256), b.tilehist, (uint32_t)ntiles, n, b.cum);
    JPK_LAUNCH(ctx, PROF_INV_BUILD, n, (k_build_nxt<true>), dim3((unsigned)ntiles), dim3(TB), d_in, n, 0u, b.state, b.tilehist, (uint32_t)ntiles, b.nxt);

    int lut_shift = 0;
    while (((uint64_t)(n - 1) >> lut_shift) >= LUT) lut_shift++;
    JPK_LAUNCH(ctx, PROF_INV_WALK, n, k_walk, dim3(jpk_grid(nsplit + 1, TB)), dim3(TB), b.nxt, b.cum, n, b.state, (uint32_t)nsplit, (uint32_t)lut_shift,
                       b.scratch, b.slot_len, b.slot_next, (uint32_t)max_slots);
    const unsigned g_s = jpk_grid(max_slots, TB);
    JPK_LAUNCH(ctx, PROF_INV_RANK, 0, k_rank_init, dim3(g_s), dim3(TB), b.slot_len, b.slot_next, b.state, (uint32_t)nsplit, (uint32_t)max_slots, b.distA, b.linkA);
    uint32_t *di = b.distA, *li = b.linkA, *dout = b.distB, *lo = b.linkB;
    const int rounds = jpk_bits_for((uint32_t)max_slots) + 1;
    for (int r = 0; r < rounds; r++) {
        JPK_LAUNCH(ctx, PROF_INV_RANK, 0, k_rank_jump, dim3(g_s), dim3(TB), di, li, b.state, (uint32_t)nsplit, (uint32_t)max_slots, dout, lo);
        uint32_t *t = di; di = dout; dout = t;
        t = li; li = lo; lo = t;
    }
    JPK_LAUNCH(ctx, PROF_INV_RANK, 0, k_head_check, dim3(1), dim3(1), di, (uint32_t)nsplit, n, (uint32_t)max_slots, b.state);
    JPK_LAUNCH(ctx, PROF_INV_COPY, n, k_copy_out, dim3(jpk_grid(max_slots * 16, TB)), dim3(TB), b.scratch, b.slot_len, di, b.state, (uint32_t)nsplit,
               (uint32_t)max_slots, n, d_out);
    if (rem > 0) JPK_LAUNCH(ctx, PROF_INV_COPY, n, k_inv_tail, dim3(1), dim3(128), d_in, n, (uint32_t)len, d_out);
    hipLaunchKernelGGL(k_inv_verdict, dim3(1), dim3(64), 0, st, b.state, d_verdict);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

// the same, synchronous: one host round trip, at the end (the caller wants the status anyway)
int jpk_inv_bwt_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len_with_trailer, uint8_t *d_out)
{
    uint32_t *verdict = ctx->d_mail + 16;
    JPK_TRY(jpk_inv_bwt_enqueue(ctx, d_in, len_with_trailer, d_out, verdict));
    JPK_HIP(hipMemcpyAsync(ctx->h_mail + 16, verdict, 16, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->prof_on) jpk_prof_resolve(ctx);
    const uint32_t *v = ctx->h_mail + 16;
    const int32_t len = len_with_trailer - JPK_TRAILER_BYTES;
    const uint32_t n = (uint32_t)(len - len % JPK_BWT_UNITS);
    if (n) {
        ctx->stats.inv_splitters = (int64_t)((n + STRIDE - 1) / STRIDE) + 1;
        ctx->stats.inv_overflow_slots = v[2];
    }
    return v[0] ? JPK_E_CORRUPT : JPK_OK;
}

// the 120-chain comparator: same Map build, then the literal chase.  *chase_ms = time of the chase kernel alone.
int jpk_inv_bwt_chains120_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len_with_trailer, uint8_t *d_out, float *chase_ms)
{
    hipStream_t st = ctx->stream;
    const int32_t len = len_with_trailer - JPK_TRAILER_BYTES;
    if (len < 0) return JPK_E_CORRUPT;
    const int32_t rem = len % JPK_BWT_UNITS;
    const uint32_t n = (uint32_t)(len - rem);
    if (chase_ms) *chase_ms = 0.f;
    if (n == 0) {
        if (rem > 0) JPK_HIP(hipMemcpyAsync(d_out, d_in, (size_t)rem, hipMemcpyDeviceToDevice, st));
        return JPK_OK;
    }
    JPK_HIP(hipMemcpyAsync(ctx->h_mail, d_in + len, 4, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipStreamSynchronize(st));
    const uint32_t I = ctx->h_mail[0];
    if (I < 1 || I > n) return JPK_E_CORRUPT;
    InvBufs b;
    size_t ntiles, nsplit, max_slots;
    Arena plan(ctx, true);
    inv_layout(plan, n, b, ntiles, nsplit, max_slots);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    inv_layout(real, n, b, ntiles, nsplit, max_slots);
    hipLaunchKernelGGL(k_hist, dim3((unsigned)ntiles), dim3(TB), 0, st, d_in, n, b.tilehist, (uint32_t)ntiles);
    JPK_TRY(jpk_exclusive_sum_u32(ctx, b.tilehist, b.tilehist, 256 * ntiles, b.scan_scratch, nullptr));
    hipLaunchKernelGGL(k_cum, dim3(1), dim3(256), 0, st, b.tilehist, (uint32_t)ntiles, n, b.cum);
    hipLaunchKernelGGL((k_build_nxt<false>), dim3((unsigned)ntiles), dim3(TB), 0, st, d_in, n, I, (const InvState *)nullptr, b.tilehist, (uint32_t)ntiles, b.nxt);
    hipEvent_t e0, e1;
    JPK_HIP(hipEventCreate(&e0));
    JPK_HIP(hipEventCreate(&e1));
    JPK_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_chase120, dim3(40), dim3(3), 0, st, b.nxt, d_in, n, (uint32_t)len, I, d_out);     // <<<40,3>>>, bwt.cpp:226-229
    JPK_HIP(hipEventRecord(e1, st));
    if (rem > 0) hipLaunchKernelGGL(k_inv_tail, dim3(1), dim3(128), 0, st, d_in, n, (uint32_t)len, d_out);
    JPK_HIP(hipGetLastError());
    JPK_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (chase_ms) *chase_ms = ms;
    return JPK_OK;
}

// ---- many inverse BWTs through one set of launches ---------------------------------------------------------------------------------------
static uint32_t inv_sorted_len(int32_t len_with_trailer) { const int32_t len = len_with_trailer - JPK_TRAILER_BYTES; return (uint32_t)(len - len % JPK_BWT_UNITS); }

// arena bytes of a batch (from ctx->arena_base): the job table, the shared tile table and its scan scratch, every job's own buffers
// (incrementally: a job adds its own buffers and its tiles; the tables depend on the job count and the tile total -- the group
// planner of jpk_dev_blocks_decompress extends a group block by block without re-planning the prefix)
void jpk_inv_bwt_batch_plan_add(int32_t len_with_trailer, size_t *job_bytes, size_t *tiles)
{
    jpk_ctx dummy;
    Arena plan(&dummy, true);
    const uint32_t n = inv_sorted_len(len_with_trailer);
    InvBufs b;
    size_t ntiles, nsplit, max_slots;
    inv_job_layout(plan, n ? n : 1, b, ntiles, nsplit, max_slots);
    *job_bytes += plan.need;
    *tiles += ntiles;
}
size_t jpk_inv_bwt_batch_plan_total(int njobs, size_t job_bytes, size_t tiles)
{
    jpk_ctx dummy;
    Arena plan(&dummy, true);
    plan.get<InvJob>((size_t)njobs);
    plan.get<uint32_t>(256 * tiles);
    plan.get<uint32_t>(jpk_scan_scratch_words(256 * tiles));
    return job_bytes + plan.need + (1u << 20);
}
size_t jpk_inv_bwt_batch_arena_bytes(int njobs, const int32_t *len_with_trailer)
{
    size_t job_bytes = 0, tiles = 0;
    for (int q = 0; q < njobs; q++) jpk_inv_bwt_batch_plan_add(len_with_trailer[q], &job_bytes, &tiles);
    return jpk_inv_bwt_batch_plan_total(njobs, job_bytes, tiles);
}

// Enqueues the inverse BWTs of `njobs` images (each with >= 120 sorted bytes: len - 480 >= 120) on ctx->stream, no host round trip;
// d_verdict[4 * slot ..] (slot = verdict_slot[q], or q) receives job q's {status, trailer index, overflow slots, head distance}.  The arena must already hold
// jpk_inv_bwt_batch_arena_bytes() from ctx->arena_base on (the caller keeps other buffers in it: it may not move here).
// `host_jobs` is the caller's: it must outlive the stream work (the table is copied from it asynchronously).
int jpk_inv_bwt_batch_enqueue(jpk_ctx *ctx, int njobs, const uint8_t *const *d_in, const int32_t *len_with_trailer, uint8_t *const *d_out, uint32_t *d_verdict,
                              const int *verdict_slot, std::vector<uint8_t> &host_jobs)
{
    if (njobs <= 0) return JPK_OK;
    if (njobs > JPK_INV_BATCH_MAX_JOBS) return JPK_E_ARG;      // blockIdx.y = the job: the grid's y extent is 65 535 (the caller cuts its groups there)
    hipStream_t st = ctx->stream;
    if (!jpk_arena_fits(ctx, ctx->arena_base + jpk_inv_bwt_batch_arena_bytes(njobs, len_with_trailer))) return JPK_E_ALLOC;
    Arena real(ctx, false);
    host_jobs.assign((size_t)njobs * sizeof(InvJob), 0);
    InvJob *hj = reinterpret_cast<InvJob *>(host_jobs.data());
    size_t tiles = 0, max_tiles = 0, max_split = 0, max_slots_all = 0;
    uint64_t pos = 0;
    for (int q = 0; q < njobs; q++) {
        const int32_t len = len_with_trailer[q] - JPK_TRAILER_BYTES;
        const uint32_t n = inv_sorted_len(len_with_trailer[q]);
        if (len < 0 || n == 0) return JPK_E_ARG;
        InvBufs b;
        size_t ntiles, nsplit, max_slots;
        inv_job_layout(real, n, b, ntiles, nsplit, max_slots);
        InvJob &j = hj[q];
        j.B = d_in[q]; j.T = d_out[q]; j.verdict = d_verdict + 4 * (size_t)(verdict_slot ? verdict_slot[q] : q);
        j.n = n; j.len = (uint32_t)len; j.ntiles = (uint32_t)ntiles; j.nsplit = (uint32_t)nsplit; j.max_slots = (uint32_t)max_slots;
        int lut_shift = 0;
        while (((uint64_t)(n - 1) >> lut_shift) >= LUT) lut_shift++;
        j.lut_shift = (uint32_t)lut_shift;
        if (pos > 0xFFFFFFFFull) return JPK_E_ARG;              // (the shared scan is 32-bit: a batch of small blocks stays far below)
        j.pos_base = (uint32_t)pos;
        pos += n;
        j.cum = b.cum; j.nxt = b.nxt; j.slot_len = b.slot_len; j.slot_next = b.slot_next;
        j.distA = b.distA; j.distB = b.distB; j.linkA = b.linkA; j.linkB = b.linkB; j.scratch = b.scratch; j.state = b.state;
        tiles += ntiles;
        if (ntiles > max_tiles) max_tiles = ntiles;
        if (nsplit > max_split) max_split = nsplit;
        if (max_slots > max_slots_all) max_slots_all = max_slots;
    }
    if (pos > 0xFFFFFFFFull) return JPK_E_ARG;
    InvJob *d_jobs = real.get<InvJob>((size_t)njobs);
    uint32_t *tilehist = real.get<uint32_t>(256 * tiles);
    uint32_t *scan_scratch = real.get<uint32_t>(jpk_scan_scratch_words(256 * tiles));
    {
        size_t off = 0;
        for (int q = 0; q < njobs; q++) { hj[q].tilehist = tilehist + off; off += 256 * (size_t)hj[q].ntiles; }
    }
    JPK_HIP(hipMemcpyAsync(d_jobs, hj, (size_t)njobs * sizeof(InvJob), hipMemcpyHostToDevice, st));
    const unsigned nj = (unsigned)njobs;
    JPK_LAUNCH(ctx, PROF_INV_HIST, 0, k_b_prep, dim3(jpk_grid(nj, 64)), dim3(64), d_jobs, nj);
    JPK_LAUNCH(ctx, PROF_INV_HIST, pos, k_b_hist, dim3((unsigned)max_tiles, nj), dim3(TB), d_jobs);
    JPK_TRY(jpk_exclusive_sum_u32(ctx, tilehist, tilehist, 256 * tiles, scan_scratch, nullptr));
    JPK_LAUNCH(ctx, PROF_INV_HIST, 0, k_b_cum, dim3(nj), dim3(256), d_jobs);
    JPK_LAUNCH(ctx, PROF_INV_BUILD, pos, k_b_build_nxt, dim3((unsigned)max_tiles, nj), dim3(TB), d_jobs);
    JPK_LAUNCH(ctx, PROF_INV_WALK, pos, k_b_walk, dim3(jpk_grid(max_split + 1, TB), nj), dim3(TB), d_jobs);
    const unsigned g_s = jpk_grid(max_slots_all, TB);
    JPK_LAUNCH(ctx, PROF_INV_RANK, 0, k_b_rank_init, dim3(g_s, nj), dim3(TB), d_jobs);
    const int rounds = jpk_bits_for((uint32_t)max_slots_all) + 1;
    for (int r = 0; r < rounds; r++) JPK_LAUNCH(ctx, PROF_INV_RANK, 0, k_b_rank_jump, dim3(g_s, nj), dim3(TB), d_jobs, r & 1);
    const int final_in_b = rounds & 1;                        // round r reads A when r is even and leaves its result in B
    JPK_LAUNCH(ctx, PROF_INV_COPY, pos, k_b_copy_out, dim3(jpk_grid(max_slots_all * 16, TB), nj), dim3(TB), d_jobs, final_in_b);
    JPK_LAUNCH(ctx, PROF_INV_COPY, 0, k_b_finish, dim3(nj), dim3(128), d_jobs, final_in_b);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}
