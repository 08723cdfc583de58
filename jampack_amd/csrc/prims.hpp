// prims.hpp -- wave64 / workgroup primitives for gfx950 (CDNA4).  Wavefront width is hard-coded to 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jpk {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive wave scans -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d, WAVE);
        if (l >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d, WAVE);
        if (l >= d) v = v > o ? v : o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_min(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d, WAVE);
        if (l >= d) v = v < o ? v : o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

struct OpSum { static __device__ __forceinline__ uint32_t id() { return 0u; }
               static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a + b; }
               static __device__ __forceinline__ uint32_t wscan(uint32_t v) { return wave_incl_sum(v); } };
struct OpMax { static __device__ __forceinline__ uint32_t id() { return 0u; }
               static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; }
               static __device__ __forceinline__ uint32_t wscan(uint32_t v) { return wave_incl_max(v); } };
struct OpMin { static __device__ __forceinline__ uint32_t id() { return 0xFFFFFFFFu; }
               static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a < b ? a : b; }
               static __device__ __forceinline__ uint32_t wscan(uint32_t v) { return wave_incl_min(v); } };

// Workgroup inclusive scan of one value per thread.  `sm` holds >= blockDim/64 + 1 words.  Returns the
// inclusive result; *block_total receives the reduction over the whole workgroup (all threads).
template <class Op>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t *sm, uint32_t *block_total)
{
    const int l = lane_id(), w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc = Op::wscan(v);
    __syncthreads();                       // protect sm from a previous use
    if (l == WAVE - 1) sm[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint32_t t = (l < nw) ? sm[l] : Op::id();
        uint32_t ti = Op::wscan(t);
        if (l < nw) sm[l] = ti;            // inclusive wave totals
    }
    __syncthreads();
    uint32_t prefix = (w > 0) ? sm[w - 1] : Op::id();
    if (block_total) *block_total = sm[nw - 1];
    return Op::f(prefix, inc);
}

// lanes of this wave whose BITS-bit digit equals mine (valid lanes only)
// m = AND_b (bit_b ? ballot_b : ~ballot_b) = ~ OR_b (ballot_b ^ D_b) with D_b = bit b of my digit replicated over a word (one
// v_bfe_i32, which also feeds the ballot's compare): five vector instructions per bit on the two halves of the lane mask
// instead of the seven or eight of the select form -- the match is the bulk of every radix kernel's VALU work.
// (the bit index is a template parameter, so the asm immediate is a constant expression whatever the optimisation level does with loops)
template <int B, int BITS>
struct MatchBits {
    static __device__ __forceinline__ void run(uint32_t d, uint32_t &lo, uint32_t &hi)
    {
        uint32_t D;                                                              // 0 or ~0; asm: the compiler would expand the bit-field extract into two shifts
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(D) : "v"(d), "n"(B));
        const uint64_t bal = __ballot(D != 0u);
        lo |= (uint32_t)bal ^ D;
        hi |= (uint32_t)(bal >> 32) ^ D;
        MatchBits<B + 1, BITS>::run(d, lo, hi);
    }
};
template <int BITS>
struct MatchBits<BITS, BITS> {
    static __device__ __forceinline__ void run(uint32_t, uint32_t &, uint32_t &) {}
};
template <int BITS>
__device__ __forceinline__ uint64_t match_any(uint32_t d, bool valid)
{
    uint32_t lo = 0, hi = 0;
    MatchBits<0, BITS>::run(d, lo, hi);
    const uint64_t v = __ballot(valid);
    return (((uint64_t)~hi << 32) | (uint64_t)~lo) & v;
}
__device__ __forceinline__ uint64_t match_any8(uint32_t d, bool valid) { return match_any<8>(d, valid); }

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace jpk
