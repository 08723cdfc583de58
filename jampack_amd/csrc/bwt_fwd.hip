// bwt_fwd.hip -- forward BWT on gfx950: GPU suffix-array construction (replaces divsufsort, divsufsort.cpp:1721)
// followed by the BWT gather and the 120 sampled ranks of BlockSort::Bwt::ForwardBwt (bwt.cpp:22-65).
//
// Suffix array = prefix doubling (Larsson-Sadakane ranks) over radix sorts, with compaction of resolved
// suffixes:
//   round 0   key = first 7 bytes (big-endian) | min(7, bytes left): one 64-bit radix sort of all n suffixes.
//             A short suffix is a proper prefix of anything it ties with on the padded bytes, and its smaller
//             length code puts it first -- plain suffix order even when the text contains 0x00.
//   round h   (h = 7, 14, 28, ...) active suffixes only: key = (rank of the group head << 32) | rank[sa + h] + 1
//             (0 past the end); sort, re-rank inside groups, write ISA, drop singletons into SA.
// Every array stays in HBM (T n, ISA 4n, SA 4n, sort ping-pong 24n, active lists 8n, temps 12n).
#include "common.hpp"
#include "prims.hpp"

using namespace jpk;

namespace {

constexpr int TB = 256;

// ---- round 0 ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_init_keys(const uint8_t *__restrict__ T, uint32_t n, uint64_t *__restrict__ keys,
                                                 uint32_t *__restrict__ vals)
{
    uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    uint32_t left = n - i;
    uint64_t k = 0;
#pragma unroll
    for (int b = 0; b < 7; b++) {
        uint64_t c = (b < (int)left) ? T[i + b] : 0;
        k |= c << (56 - 8 * b);
    }
    k |= (left < 7u) ? left : 7u;
    keys[i] = k;
    vals[i] = i;
}

// head flags of equal-key runs -> hv[i] = head ? i : 0  (input of an inclusive max scan)
__global__ __launch_bounds__(TB) void k_heads_u64(const uint64_t *__restrict__ keys, uint32_t m, uint32_t *__restrict__ hv)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    bool head = (j == 0) || (keys[j] != keys[j - 1]);
    hv[j] = head ? j : 0u;
}

// round 0: grp[] (= index of the run head) -> ISA, singletons -> SA, keep flags for compaction
__global__ __launch_bounds__(TB) void k_round0_finish(const uint32_t *__restrict__ grp, const uint32_t *__restrict__ sa, uint32_t n,
                                                     uint32_t *__restrict__ ISA, uint32_t *__restrict__ SA, uint32_t *__restrict__ keep)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= n) return;
    uint32_t g = grp[j], s = sa[j];
    ISA[s] = g;
    bool head = (g == j);
    bool next_head = (j + 1 == n) || (grp[j + 1] == j + 1);
    bool single = head && next_head;
    if (single) SA[j] = s;
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

// ---- doubling rounds --------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_make_keys(const uint32_t *__restrict__ a_sa, const uint32_t *__restrict__ a_grp, uint32_t m, uint32_t n,
                                                 uint32_t h, const uint32_t *__restrict__ ISA, uint64_t *__restrict__ keys,
                                                 uint32_t *__restrict__ vals, uint32_t *__restrict__ hv)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    uint32_t s = a_sa[j], g = a_grp[j];
    uint64_t s2 = (uint64_t)s + h;
    uint32_t k2 = (s2 < n) ? ISA[s2] + 1u : 0u;
    keys[j] = ((uint64_t)g << 32) | k2;
    vals[j] = s;
    // old-group head positions in the active list (groups stay contiguous through the sort)
    bool head = (j == 0) || (a_grp[j - 1] != g);
    hv[j] = head ? j : 0u;
}

// abs position of element j after the sort = group rank + offset inside the (old) group;
// new head flag from the full 64-bit key; nh[j] = newhead ? abspos : 0 (input of the max scan)
__global__ __launch_bounds__(TB) void k_abspos(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ jhead, uint32_t m,
                                              uint32_t *__restrict__ abspos, uint32_t *__restrict__ nh)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    uint64_t k = keys[j];
    uint32_t ap = (uint32_t)(k >> 32) + (j - jhead[j]);
    abspos[j] = ap;
    bool head = (j == 0) || (keys[j - 1] != k);
    nh[j] = head ? ap : 0u;
}

// new ranks -> ISA; singletons -> SA; keep flags
__global__ __launch_bounds__(TB) void k_round_finish(const uint32_t *__restrict__ abspos, const uint32_t *__restrict__ newrank,
                                                    const uint32_t *__restrict__ vals, uint32_t m, uint32_t *__restrict__ ISA,
                                                    uint32_t *__restrict__ SA, uint32_t *__restrict__ keep)
{
    uint32_t j = blockIdx.x * TB + threadIdx.x;
    if (j >= m) return;
    uint32_t ap = abspos[j], r = newrank[j], s = vals[j];
    ISA[s] = r;
    bool head = (r == ap);
    bool next_head = (j + 1 == m) || (newrank[j + 1] == abspos[j + 1]);
    bool single = head && next_head;
    if (single) SA[ap] = s;
    keep[j] = single ? 0u : 1u;
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

__global__ __launch_bounds__(TB) void k_copy_u32_as_i32(const uint32_t *__restrict__ a, int32_t *__restrict__ b, uint32_t n)
{
    uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i < n) b[i] = (int32_t)a[i];
}

struct SaBufs {
    uint64_t *keysA, *keysB;
    uint32_t *valsA, *valsB, *ISA, *SA, *a_sa, *a_grp, *t1, *t2, *t3, *scratch;
};

void sa_layout(Arena &a, size_t n, SaBufs &b, bool need_sa_buf)
{
    b.keysA = a.get<uint64_t>(n);
    b.keysB = a.get<uint64_t>(n);
    b.valsA = a.get<uint32_t>(n);
    b.valsB = a.get<uint32_t>(n);
    b.ISA = a.get<uint32_t>(n);
    b.SA = need_sa_buf ? a.get<uint32_t>(n) : nullptr;
    b.a_sa = a.get<uint32_t>(n);
    b.a_grp = a.get<uint32_t>(n);
    b.t1 = a.get<uint32_t>(n);
    b.t2 = a.get<uint32_t>(n);
    b.t3 = a.get<uint32_t>(n);
    size_t sw = jpk_radix_scratch_words(n);
    size_t sc = jpk_scan_scratch_words(n);
    b.scratch = a.get<uint32_t>(sw > sc ? sw : sc);
}

// builds SA (uint32) and ISA for T[0..n); T must be readable; returns rounds via ctx->stats
int build_sa(jpk_ctx *ctx, const uint8_t *T, uint32_t n, SaBufs &b)
{
    hipStream_t st = ctx->stream;
    const unsigned g_n = jpk_grid(n, TB);
    ctx->stats.sa_rounds = 0;
    ctx->stats.sa_sorted_elems = 0;

    // round 0: sort by 7 bytes + length code (bits 0..2 and 8..63; the digit at bits 0..7 holds only the code)
    hipLaunchKernelGGL(k_init_keys, dim3(g_n), dim3(TB), 0, st, T, n, b.keysA, b.valsA);
    {
        const int shifts[8] = {0, 8, 16, 24, 32, 40, 48, 56};
        JPK_TRY(jpk_radix_sort_pairs_u64(ctx, b.keysA, b.valsA, b.keysB, b.valsB, n, shifts, 8, b.scratch));
        ctx->stats.sa_sorted_elems += n;
    }
    hipLaunchKernelGGL(k_heads_u64, dim3(g_n), dim3(TB), 0, st, b.keysA, n, b.t1);
    JPK_TRY(jpk_inclusive_max_u32(ctx, b.t1, b.t2, n, b.scratch));                     // t2 = grp
    hipLaunchKernelGGL(k_round0_finish, dim3(g_n), dim3(TB), 0, st, b.t2, b.valsA, n, b.ISA, b.SA, b.t1);  // t1 = keep
    JPK_TRY(jpk_exclusive_sum_u32(ctx, b.t1, b.t3, n, b.scratch, ctx->d_mail));       // t3 = pos
    hipLaunchKernelGGL(k_compact, dim3(g_n), dim3(TB), 0, st, b.t1, b.t3, b.valsA, b.t2, n, b.a_sa, b.a_grp);
    uint32_t m = 0;
    JPK_TRY(jpk_read_mail(ctx, &m, 1));
    ctx->stats.sa_rounds = 1;

    const int kbits = jpk_bits_for(n);     // key2 <= n, group rank < n
    int shifts[8];
    int ns = 0;
    for (int s = 0; s < kbits; s += 8) shifts[ns++] = s;
    for (int s = 0; s < kbits; s += 8) shifts[ns++] = 32 + s;

    uint64_t h = 7;
    while (m > 0) {
        if (h >= n) return JPK_E_DEVICE;   // cannot happen: every suffix is unique once h >= n
        const unsigned g_m = jpk_grid(m, TB);
        hipLaunchKernelGGL(k_make_keys, dim3(g_m), dim3(TB), 0, st, b.a_sa, b.a_grp, m, n, (uint32_t)h, b.ISA, b.keysA, b.valsA, b.t1);
        JPK_TRY(jpk_inclusive_max_u32(ctx, b.t1, b.t2, m, b.scratch));                 // t2 = jhead (old groups)
        JPK_TRY(jpk_radix_sort_pairs_u64(ctx, b.keysA, b.valsA, b.keysB, b.valsB, m, shifts, ns, b.scratch));
        ctx->stats.sa_sorted_elems += m;
        hipLaunchKernelGGL(k_abspos, dim3(g_m), dim3(TB), 0, st, b.keysA, b.t2, m, b.t1, b.t3);  // t1 = abspos, t3 = nh
        JPK_TRY(jpk_inclusive_max_u32(ctx, b.t3, b.t2, m, b.scratch));                 // t2 = newrank
        hipLaunchKernelGGL(k_round_finish, dim3(g_m), dim3(TB), 0, st, b.t1, b.t2, b.valsA, m, b.ISA, b.SA, b.t3);  // t3 = keep
        JPK_TRY(jpk_exclusive_sum_u32(ctx, b.t3, b.t1, m, b.scratch, ctx->d_mail));   // t1 = pos
        hipLaunchKernelGGL(k_compact, dim3(g_m), dim3(TB), 0, st, b.t3, b.t1, b.valsA, b.t2, m, b.a_sa, b.a_grp);
        uint32_t m2 = 0;
        JPK_TRY(jpk_read_mail(ctx, &m2, 1));
        m = m2;
        h <<= 1;
        ctx->stats.sa_rounds++;
    }
    JPK_HIP(hipGetLastError());
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
    return build_sa(ctx, d_t, (uint32_t)n, b);
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
    JPK_TRY(build_sa(ctx, d_in, (uint32_t)nlen, b));
    hipLaunchKernelGGL(k_bwt_gather, dim3(jpk_grid(nlen, TB)), dim3(TB), 0, ctx->stream, d_in, b.SA, b.ISA, (uint32_t)nlen, d_out);
    hipLaunchKernelGGL(k_bwt_trailer, dim3(1), dim3(128), 0, ctx->stream, d_in, b.ISA, (uint32_t)nlen, (uint32_t)len, d_out);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}
