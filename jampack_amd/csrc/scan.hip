// scan.hip -- device-wide prefix scans over uint32 (reduce / scan-partials / downsweep).
// HBM-bound: 2 reads + 1 write of the array per scan.
#include "common.hpp"
#include "prims.hpp"

using namespace jpk;

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t *__restrict__ in, size_t n, uint32_t *__restrict__ partial)
{
    __shared__ uint32_t sm[SCAN_THREADS / 64 + 1];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t acc = Op::id();
    if (base + SCAN_ITEMS <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS / 4; k++) {
            uint4 v = p[k];
            acc = Op::f(acc, Op::f(Op::f(v.x, v.y), Op::f(v.z, v.w)));
        }
    } else {
        for (int k = 0; k < SCAN_ITEMS; k++)
            if (base + k < n) acc = Op::f(acc, in[base + k]);
    }
    uint32_t tot;
    block_incl_scan<Op>(acc, sm, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// one workgroup walks all partials in batches of 1024 with a running carry; partial[] becomes the
// exclusive prefix per tile, total (if requested) the grand reduction.
template <class Op>
__global__ __launch_bounds__(1024) void k_scan_partials(uint32_t *__restrict__ partial, size_t nb, uint32_t *__restrict__ total)
{
    __shared__ uint32_t sm[1024 / 64 + 1];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = Op::id();
    __syncthreads();
    for (size_t b0 = 0; b0 < nb; b0 += 1024) {
        size_t i = b0 + threadIdx.x;
        uint32_t v = (i < nb) ? partial[i] : Op::id();
        uint32_t tot;
        uint32_t inc = block_incl_scan<Op>(v, sm, &tot);
        uint32_t carry = carry_s;
        // exclusive prefix = carry (op) inclusive-of-previous
        uint32_t prev = __shfl_up(inc, 1, 64);
        if (lane_id() == 0) prev = (threadIdx.x == 0) ? Op::id() : sm[(threadIdx.x >> 6) - 1];
        if (i < nb) partial[i] = Op::f(carry, prev);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = Op::f(carry, tot);
        __syncthreads();
    }
    if (total && threadIdx.x == 0) *total = carry_s;
}

template <class Op, bool EXCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_down(const uint32_t *in, uint32_t *out, size_t n,
                                                           const uint32_t *__restrict__ partial)
{
    __shared__ uint32_t sm[SCAN_THREADS / 64 + 1];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    if (base + SCAN_ITEMS <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS / 4; k++) {
            uint4 q = p[k];
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (base + k < n) ? in[base + k] : Op::id();
    }
    uint32_t acc = Op::id();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) acc = Op::f(acc, v[k]);
    uint32_t inc = block_incl_scan<Op>(acc, sm, nullptr);
    // exclusive prefix of this thread
    uint32_t prev = __shfl_up(inc, 1, 64);
    if (lane_id() == 0) prev = (threadIdx.x == 0) ? Op::id() : sm[(threadIdx.x >> 6) - 1];
    uint32_t run = Op::f(partial[blockIdx.x], prev);
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (EXCLUSIVE) { o[k] = run; run = Op::f(run, v[k]); }
        else { run = Op::f(run, v[k]); o[k] = run; }
    }
    if (base + SCAN_ITEMS <= n) {
        uint4 *p = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS / 4; k++) p[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++)
            if (base + k < n) out[base + k] = o[k];
    }
}

template <class Op, bool EXCLUSIVE>
int run_scan(jpk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch, uint32_t *d_total)
{
    if (n == 0) {
        if (d_total) JPK_HIP(hipMemsetAsync(d_total, 0, 4, ctx->stream));
        return JPK_OK;
    }
    size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    JPK_LAUNCH(ctx, PROF_SCAN, 0, (k_scan_reduce<Op>), dim3((unsigned)nb), dim3(SCAN_THREADS), in, n, scratch);
    JPK_LAUNCH(ctx, PROF_SCAN, 0, (k_scan_partials<Op>), dim3(1), dim3(1024), scratch, nb, d_total);
    JPK_LAUNCH(ctx, PROF_SCAN, n, (k_scan_down<Op, EXCLUSIVE>), dim3((unsigned)nb), dim3(SCAN_THREADS), in, out, n, scratch);
    JPK_HIP(hipGetLastError());
    return JPK_OK;
}

}  // namespace

size_t jpk_scan_scratch_words(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 64; }

int jpk_exclusive_sum_u32(jpk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch, uint32_t *d_total)
{
    return run_scan<OpSum, true>(ctx, in, out, n, scratch, d_total);
}

int jpk_inclusive_max_u32(jpk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch)
{
    return run_scan<OpMax, false>(ctx, in, out, n, scratch, nullptr);
}
