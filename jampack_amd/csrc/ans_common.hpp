// ans_common.hpp -- constants and small device helpers shared by the entropy encoder and decoder.
#pragma once
#include <stdint.h>

#include "../../include/jampack_abi.h"
#include "prims.hpp"

namespace jpk {

constexpr int ANS_CHUNK = JPK_ANS_CHUNK;     // ans.hpp:21
constexpr int ATILE = 4096;                  // tile (bytes or symbols) of the per-chunk passes
constexpr int NQ = 32;                       // quasi-model rebuild intervals that fit in one chunk
constexpr int QSTRIDE = 132;                 // padded alphabet (129 + cdf end)
constexpr uint32_t RANS_L = 1u << 23;        // rans_byte.hpp:50

// tables.hpp:10 Exponent[] ; class e covers symbols [EXPO[e], EXPO[e+1])
__device__ __constant__ const int d_EXPO[9] = {0, 2, 4, 8, 16, 32, 64, 128, 257};

// tables.hpp:12-20 Log[]: 0,0,1,1,2x4,3x8,4x16,5x32,6x64,7x129
__device__ __forceinline__ int sym_class(uint32_t s)
{
    if (s < 2) return 0;
    if (s >= 128) return 7;
    return 31 - __clz((int)s);
}
__device__ __forceinline__ int class_base(int e) { return e == 0 ? 0 : (1 << e); }
__device__ __forceinline__ int class_alpha(int e) { return e == 7 ? 129 : (e == 0 ? 2 : (1 << e)); }

// QuasiModel rebuild schedule (model.cpp:160-204): the model is rebuilt after EXP+1 symbols, EXP = 8,16,..,65536.
// Interval q of a class covers class-ordinals [qbound(q), qbound(q+1)) and is coded with the CDF built from the
// histogram of interval q-1 (interval 0: uniform CDF).
__host__ __device__ __forceinline__ uint32_t qbound(int q)
{
    // q <= 13: sum_{k<q} (8*2^k + 1) = 8*(2^q - 1) + q ; afterwards +65537 each
    if (q <= 13) return 8u * ((1u << q) - 1u) + (uint32_t)q;
    return 8u * ((1u << 13) - 1u) + 13u + (uint32_t)(q - 13) * 65537u;
}
__device__ __forceinline__ int qinterval(uint32_t k)
{
    int q = 0;
    while (q + 1 < NQ && qbound(q + 1) <= k) q++;
    return q;
}

// uniform CDF entry i of an alphabet of A symbols (model.cpp:85-95)
__host__ __device__ __forceinline__ uint32_t uniform_cdf(int A, int i)
{
    if (i <= 0) return 0;
    uint32_t scale = 65536u / (uint32_t)A;
    return (uint32_t)i * scale + (65536u - scale * (uint32_t)A);
}

// one step of the adaptive CDF recurrence for entry i given the coded symbol (model.cpp:60-77, closed-form mix row)
__device__ __forceinline__ int32_t adapt_step(int32_t x, int i, int sym, int A)
{
    int32_t mix = (i <= sym) ? i : i + 65536 - A;
    return x + ((mix - x) >> 5);
}

// LEB128 "with carry" (utils.cpp:22-68)
__device__ __forceinline__ int leb_encode(uint32_t v, uint8_t *b)
{
    const uint32_t C0 = 127u, C1 = 16510u, C2 = 2113661u, C3 = 270549116u;
    int n;
    if (v < C0) n = 1;
    else if (v < C1) { n = 2; v -= C0; }
    else if (v < C2) { n = 3; v -= C1; }
    else if (v < C3) { n = 4; v -= C2; }
    else { n = 5; v -= C3; }
    for (int k = 0; k < n; k++) b[k] = (uint8_t)((v >> (7 * (n - 1 - k))) & 0x7f);
    b[n - 1] |= 0x80;
    return n;
}
// returns bytes consumed or -1
__device__ __forceinline__ int leb_decode(uint32_t *v, const uint8_t *b, int64_t avail)
{
    const uint32_t C[4] = {127u, 16510u, 2113661u, 270549116u};
    int d = 0;
    uint32_t x = 0;
    while (d < avail && !(b[d] & 0x80)) {
        if (d >= 4) return -1;
        x = (x << 7) | b[d++];
    }
    if (d >= avail) return -1;
    x = (x << 7) | (b[d] & 0x7fu);
    if (d > 0) x += C[d - 1];
    *v = x;
    return d + 1;
}

}  // namespace jpk
