// ans_dec.hip -- Ans::Decode (ans.cpp:236-270) on gfx950.
//   header walk (ans.cpp:254-261, ReadHeader :287-302)           one lane, chunk chain is serial by format
//   rANS + model decode (Threaded_Decode, ans.cpp:30-92)          one wave per chunk; lanes = CDF entries
//   RLE0 decode (rle.cpp:52-74)                                   one workgroup per chunk, scan based
//   sorted-rank decode (rank.cpp:96-151)                          one wave per chunk; the 256-entry list is one
//                                                                 byte-vector register (4 positions per lane)
// The entropy decoder is sequential inside a chunk by construction of the format (shared byte pointer of the
// four rANS states, adaptive models, bucket hopping of the rank decoder); parallelism comes from the chunks.
#include "ans_common.hpp"
#include "common.hpp"

using namespace jpk;

namespace {

constexpr int TB = 256;

struct ChunkInfo {          // one per chunk, written by the header walk
    uint64_t in_off;        // payload offset in the input
    uint32_t clen, olen, rlen;
    uint32_t pad;
    uint64_t out_off;       // offset in the decoded output
    uint64_t rle_off;       // offset in the packed rle buffer
};

// mail: [0]=status, [1]=nch, [2..3]=total out, [4..5]=total rle
__global__ void k_dec_headers(const uint8_t *__restrict__ in, uint32_t len, uint64_t out_cap, uint32_t max_chunks, ChunkInfo *__restrict__ info,
                              int32_t *__restrict__ freq, uint32_t *__restrict__ mail)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t ip = 0, op = 0, rp = 0;
    uint32_t nch = 0;
    int status = 0;
    while (ip < len) {
        if (nch >= max_chunks) { status = JPK_E_CORRUPT; break; }
        uint32_t v = 0;
        int64_t total = 0;
        bool bad = false;
        for (int s = 0; s < 256; s++) {
            int n = leb_decode(&v, in + ip, (int64_t)len - (int64_t)ip);
            if (n < 0) { bad = true; break; }
            ip += n;
            if (v > (uint32_t)ANS_CHUNK) { bad = true; break; }
            freq[(size_t)nch * 256 + s] = (int32_t)v;
            total += v;
        }
        uint32_t olen = 0, clen = 0, rlen = 0;
        int n = 0;
        if (!bad) { n = leb_decode(&olen, in + ip, (int64_t)len - (int64_t)ip); if (n < 0) bad = true; else ip += n; }
        if (!bad) { n = leb_decode(&clen, in + ip, (int64_t)len - (int64_t)ip); if (n < 0) bad = true; else ip += n; }
        if (!bad) { n = leb_decode(&rlen, in + ip, (int64_t)len - (int64_t)ip); if (n < 0) bad = true; else ip += n; }
        if (bad || olen > (uint32_t)ANS_CHUNK || rlen > (uint32_t)ANS_CHUNK || (uint64_t)clen > len - ip || clen < 16 ||
            total != (int64_t)olen) {                                       // ans.cpp:297-298, rank.cpp:104-108
            status = JPK_E_CORRUPT;
            break;
        }
        if (op + olen > out_cap) { status = JPK_E_CAPACITY; break; }
        ChunkInfo ci;
        ci.in_off = ip; ci.clen = clen; ci.olen = olen; ci.rlen = rlen; ci.pad = 0; ci.out_off = op; ci.rle_off = rp;
        info[nch] = ci;
        ip += clen; op += olen; rp += rlen;
        nch++;
    }
    mail[0] = (uint32_t)status;
    mail[1] = nch;
    mail[2] = (uint32_t)op; mail[3] = (uint32_t)(op >> 32);
    mail[4] = (uint32_t)rp; mail[5] = (uint32_t)(rp >> 32);
}

// ---------------------------------------------------------------------------------------------------------------
// rANS + model decode: one wave per chunk (ans.cpp:30-92)
// ---------------------------------------------------------------------------------------------------------------
struct QuasiLds {
    uint32_t cdf[6][QSTRIDE];
    uint32_t f[6][QSTRIDE];
    uint32_t seen[6], expn[6];
};

__device__ __forceinline__ void quasi_rebuild(QuasiLds &q, int k, int A, int l)
{
    uint32_t F[3];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) { int i = l + 64 * j; F[j] = (i < A) ? q.f[k][i] : 0u; tot += F[j]; }
    tot = wave_sum(tot);
    int lg = 0;
    while ((tot >> lg) + (uint32_t)A > 65536u) lg++;
    uint32_t t2 = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) { int i = l + 64 * j; F[j] = (i < A) ? (F[j] >> lg) + 1u : 0u; t2 += F[j]; }
    t2 = wave_sum(t2);
    uint32_t t3 = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) { F[j] = (65536u * F[j]) / t2; t3 += F[j]; }
    t3 = wave_sum(t3);
    if (l == 0) F[0] += 65536u - t3;
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        int i = l + 64 * j;
        uint32_t inc = wave_incl_sum(F[j]);
        if (i < A) { q.cdf[k][i] = carry + inc - F[j]; q.f[k][i] = 0; }
        carry += __shfl(inc, 63, 64);
    }
    if (l == 0) { q.cdf[k][A] = 65536u; q.seen[k] = 0; q.expn[k] = (q.expn[k] < 65536u) ? q.expn[k] << 1 : 65536u; }
}

// 256-byte input window held across the wave (lane l = dword l), next window prefetched: the payload byte a
// renormalisation needs is a v_readlane away instead of a dependent global load.
struct ByteWindow {
    const uint32_t *pw;       // dword-aligned base (payload start rounded down)
    uint32_t a;               // payload start - aligned base (0..3)
    int64_t gbase;            // byte offset of pw inside the whole input buffer
    int64_t in_len;
    const uint8_t *in;
    uint32_t cur, nxt;        // my dword of the current / next window
    uint32_t wi;              // index of the current window
    int l;
    __device__ __forceinline__ uint32_t load(uint32_t w) const
    {
        const int64_t g = gbase + ((int64_t)w * 64 + l) * 4;     // byte offset in the input buffer
        if (g + 4 <= in_len) return pw[(size_t)w * 64 + l];
        uint32_t v = 0;
        for (int k = 0; k < 4; k++)
            if (g + k < in_len) v |= (uint32_t)in[g + k] << (8 * k);
        return v;
    }
    __device__ __forceinline__ void init(const uint8_t *input, int64_t input_len, const uint8_t *p, int lane)
    {
        in = input; in_len = input_len; l = lane;
        a = (uint32_t)((uintptr_t)p & 3u);
        pw = reinterpret_cast<const uint32_t *>(p - a);
        gbase = (p - a) - input;
        wi = 0;
        cur = load(0);
        nxt = load(1);
    }
    // byte at payload offset ptr (uniform); windows only move forward
    __device__ __forceinline__ uint32_t get(uint32_t ptr)
    {
        const uint32_t o = ptr + a;
        if ((o >> 8) != wi) {             // uniform
            wi++;
            cur = nxt;
            nxt = load(wi + 1);
        }
        const uint32_t d = __builtin_amdgcn_readlane(cur, (o & 255u) >> 2);
        return (d >> ((o & 3u) * 8u)) & 0xffu;
    }
};

__global__ __launch_bounds__(64) void k_dec_rans(const uint8_t *__restrict__ in, int64_t in_len, const ChunkInfo *__restrict__ info,
                                                uint16_t *__restrict__ rle, uint32_t *__restrict__ status)
{
    __shared__ QuasiLds q;
    __builtin_amdgcn_s_setprio(3);         // serial chain: win the issue arbitration on a shared SIMD
    const uint32_t c = blockIdx.x;
    const int l = lane_id();
    const ChunkInfo ci = info[c];
    const uint8_t *p = in + ci.in_off;
    const uint32_t clen = ci.clen, rlen = ci.rlen;
    uint16_t *out = rle + ci.rle_off;

    for (int k = 0; k < 6; k++) {
        const int A = class_alpha(k + 2);
        for (int i = l; i < QSTRIDE; i += 64) { q.cdf[k][i] = (i <= A) ? uniform_cdf(A, i) : 65536u; q.f[k][i] = 0; }
        if (l == 0) { q.seen[k] = 0; q.expn[k] = 8; }
    }
    __syncthreads();
    ByteWindow bw;
    bw.init(in, in_len, p, l);
    // exponent model: lane i holds cdf[i] (i <= 8); other lanes hold 65536 so they never count
    int32_t ex = (l <= 8) ? (int32_t)uniform_cdf(8, l) : 65536;
    int32_t a0 = 32768, a1 = 32768;          // cdf[1] of the two alphabet-2 mantissa models
    uint32_t R0, R1, R2, R3;
    {
        uint32_t b[16];
        for (int k = 0; k < 16; k++) b[k] = bw.get(k);
        R0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        R1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        R2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        R3 = b[12] | (b[13] << 8) | (b[14] << 16) | (b[15] << 24);
    }
    uint32_t ptr = 16;
    bool bad = false;
    uint32_t mysym = 0;
    for (uint32_t t = 0; t < rlen; t++) {
        // ---- exponent ----
        uint32_t x = R0;
        uint32_t range = x & 0xffffu;
        const int e = __popcll(__ballot(l >= 1 && l <= 7 && (uint32_t)ex <= range));
        uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(ex, e);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readlane(ex, e + 1);
        if (l >= 1 && l <= 7) ex = adapt_step(ex, l, e, 8);
        x = (hi - lo) * (x >> 16) + range - lo;
        while (x < RANS_L) {
            if (ptr >= clen) { bad = true; break; }
            x = (x << 8) | bw.get(ptr);
            ptr++;
        }
        R0 = R1; R1 = R2; R2 = R3; R3 = x;
        // ---- mantissa ----
        x = R0;
        range = x & 0xffffu;
        uint32_t m;
        if (e < 2) {
            const int32_t a = (e == 0) ? a0 : a1;
            m = (range >= (uint32_t)a) ? 1u : 0u;
            lo = m ? (uint32_t)a : 0u;
            hi = m ? 65536u : (uint32_t)a;
            const int32_t na = adapt_step(a, 1, (int)m, 2);
            if (e == 0) a0 = na; else a1 = na;
        } else {
            const int k = e - 2, A = class_alpha(e);
            const uint32_t c1 = q.cdf[k][l + 1];                                   // entries 1..64
            m = (uint32_t)__popcll(__ballot((l + 1 < A) && c1 <= range));
            if (A > 65) {                                                         // class 7 only: entries 65..128
                const uint32_t c2 = q.cdf[k][l + 65];
                m += (uint32_t)__popcll(__ballot((l + 65 < A) && c2 <= range));
            }
            lo = q.cdf[k][m];
            hi = q.cdf[k][m + 1];
            // QuasiModel::Update (model.cpp:160-204); a single wave: LDS operations complete in program order
            const uint32_t seen = q.seen[k] + 1, expn = q.expn[k];
            if (l == 0) { q.f[k][m] += 16u; q.seen[k] = seen; }
            if (seen > expn) { __syncthreads(); quasi_rebuild(q, k, A, l); __syncthreads(); }
        }
        x = (hi - lo) * (x >> 16) + range - lo;
        while (x < RANS_L) {
            if (ptr >= clen) { bad = true; break; }
            x = (x << 8) | bw.get(ptr);
            ptr++;
        }
        R0 = R1; R1 = R2; R2 = R3; R3 = x;
        if (bad) break;
        const uint32_t sym = (uint32_t)class_base(e) + m;
        if ((t & 63u) == (uint32_t)l) mysym = sym;
        if ((t & 63u) == 63u) out[t - 63 + l] = (uint16_t)mysym;
    }
    if (!bad && (rlen & 63u)) {
        const uint32_t base = rlen & ~63u;
        if (base + l < rlen) out[base + l] = (uint16_t)mysym;
    }
    if (R0 != RANS_L || R1 != RANS_L || R2 != RANS_L || R3 != RANS_L) bad = true;     // ans.cpp:91-92
    if (bad && l == 0) atomicOr(status, 1u);
}

// ---------------------------------------------------------------------------------------------------------------
// RLE0 decode: one workgroup per chunk.  Output is pre-zeroed, so only symbols > 1 are written; a digit group
// contributes value-1 zeros at the position of its last digit (rle.cpp:52-74).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_dec_rle(const uint16_t *__restrict__ rle, const ChunkInfo *__restrict__ info, uint8_t *__restrict__ out,
                                                 uint32_t *__restrict__ status)
{
    const uint32_t c = blockIdx.x;
    const ChunkInfo ci = info[c];
    const uint16_t *src = rle + ci.rle_off;
    uint8_t *dst = out + ci.out_off;
    const uint32_t rlen = ci.rlen, olen = ci.olen;
    __shared__ uint32_t sm[1024 / 64 + 1];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t bad_s;
    if (threadIdx.x == 0) { carry_s = 0; bad_s = 0; }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < rlen; b0 += 1024 * 4) {
        const uint32_t p0 = b0 + threadIdx.x * 4;
        uint32_t cnt[4], val[4];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t = p0 + k;
            cnt[k] = 0; val[k] = 0;
            if (t < rlen) {
                const uint32_t sy = src[t];
                if (sy > 256u) atomicOr(&bad_s, 1u);
                if (sy > 1u) { cnt[k] = 1; val[k] = sy - 1u; }
                else if (t + 1 >= rlen || src[t + 1] > 1u) {     // last digit of a group: gather the group backwards
                    uint32_t bits = 0, nb = 0;
                    int64_t u = t;
                    while (u >= 0 && src[u] <= 1u && nb <= 21) { bits |= (uint32_t)src[u] << nb; nb++; u--; }
                    if (nb > 20) atomicOr(&bad_s, 1u);
                    else cnt[k] = ((1u << nb) | bits) - 1u;
                }
            }
            s += cnt[k];
        }
        uint32_t tot;
        uint32_t inc = block_incl_scan<OpSum>(s, sm, &tot);
        uint32_t run = carry_s + inc - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (val[k]) { if (run < olen) dst[run] = (uint8_t)val[k]; else atomicOr(&bad_s, 1u); }
            run += cnt[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && (carry_s != olen || bad_s)) atomicOr(status, 2u);     // rle.cpp:72 "rle mismatch!"
}

// ---------------------------------------------------------------------------------------------------------------
// sorted-rank decode (rank.cpp:96-151): one wave per chunk.
// list: 256 byte positions held as one dword per lane (lane l = positions 4l..4l+3, little endian).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v)        // lane i <- lane i+1 (lane 63 keeps its value): one DPP move
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t list_shift_insert(uint32_t v, int l, uint32_t r, uint32_t sym)
{
    // positions < r take the value of position+1, position r takes sym, positions > r unchanged
    const uint32_t nextv = wave_shl1(v);
    const uint32_t shifted = (v >> 8) | (nextv << 24);
    const int nb = (int)r - 4 * l;                      // bytes of this lane below position r
    uint32_t mask = nb <= 0 ? 0u : (nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u));
    uint32_t res = (shifted & mask) | (v & ~mask);
    if (nb >= 0 && nb < 4) res = (res & ~(0xFFu << (8 * nb))) | (sym << (8 * nb));
    return res;
}

__global__ __launch_bounds__(64) void k_dec_rank(uint8_t *__restrict__ data, const ChunkInfo *__restrict__ info, const int32_t *__restrict__ freq,
                                                uint8_t *__restrict__ tmp, uint32_t *__restrict__ status)
{
    __builtin_amdgcn_s_setprio(3);         // serial chain: win the issue arbitration on a shared SIMD
    const uint32_t c = blockIdx.x;
    const int l = lane_id();
    const ChunkInfo ci = info[c];
    const uint32_t len = ci.olen;
    if (len == 0) return;
    const uint8_t *R = data + ci.out_off;        // rank array
    uint8_t *T = tmp + ci.out_off;               // decoded symbols (copied back by the caller)
    __shared__ uint32_t sf[256];
    __shared__ uint32_t bpos[256], bend[256];
    __shared__ uint32_t lst[64];
    const int32_t *fq = freq + (size_t)c * 256;
    for (int s = l; s < 256; s += 64) sf[s] = (uint32_t)fq[s];
    for (int i = l; i < 64; i += 64) lst[i] = 0;
    __syncthreads();
    uint32_t uniq = 0;
    // bucket layout in GenerateSortedMap order; list[R[bucket start]] = symbol (rank.cpp:114-123)
    for (int s = l; s < 256; s += 64) {
        const uint32_t f = sf[s];
        uint32_t b = 0;
        for (int k = 0; k < 256; k++) { uint32_t fk = sf[k]; if (fk > f || (fk == f && k < s)) b += fk; }
        bpos[s] = b + 1;
        bend[s] = b + f;
        if (f > 0) {
            const uint32_t r0 = R[b];
            atomicOr(&lst[r0 >> 2], (uint32_t)s << (8 * (r0 & 3u)));
        }
        uniq += (f > 0);
    }
    uniq = wave_sum(uniq);
    __syncthreads();
    uint32_t v = lst[l];
    uint32_t sym = rfl(v) & 0xffu;
    // One iteration = one run of the current symbol: the wave peeks the next 64 ranks of the symbol's bucket, z leading
    // zero ranks mean z more copies of the symbol (rank 0 = "same symbol again"), which are written by z+1 lanes at once;
    // the first non-zero rank (or the end of the bucket) then moves the symbol inside / out of the list.
    uint32_t i = 0;
    while (i < len) {
        const uint32_t bp = bpos[sym], be = bend[sym];
        const uint32_t rl = (bp + l < be) ? (uint32_t)R[bp + l] : 0xFFu;    // lanes past the bucket end count as non-zero
        const uint64_t nz = __ballot(rl != 0);
        const uint32_t z = nz ? (uint32_t)__builtin_ctzll(nz) : 64u;
        uint32_t cnt = (z < 64u) ? z + 1u : 64u;
        if (cnt > len - i) cnt = len - i;
        if ((uint32_t)l < cnt) T[i + l] = (uint8_t)sym;
        i += cnt;
        if (z >= 64u) {                                  // 64 zero ranks consumed, the same symbol goes on
            if (l == 0) bpos[sym] = bp + 64u;
            continue;
        }
        if (bp + z < be) {
            const uint32_t r = __builtin_amdgcn_readlane(rl, z);             // the non-zero rank that ends the run
            if (l == 0) bpos[sym] = bp + z + 1u;
            v = list_shift_insert(v, l, r, sym);
            sym = rfl(v) & 0xffu;
        } else {
            if (l == 0) bpos[sym] = be;
            if (uniq > 0) {
                uniq--;
                // drop the front: positions < uniq shift down by one (rank.cpp:140-147; executes at least once)
                const uint32_t lim = uniq > 0 ? uniq : 1u;
                const uint32_t nextv = wave_shl1(v);
                const uint32_t shifted = (v >> 8) | (nextv << 24);
                const int nb = (int)lim - 4 * l;
                const uint32_t mask = nb <= 0 ? 0u : (nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u));
                v = (shifted & mask) | (v & ~mask);
                sym = rfl(v) & 0xffu;
            }
        }
    }
}

}  // namespace

int jpk_ans_decode_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    *out_len = 0;
    if (len == 0) return JPK_OK;
    hipStream_t st = ctx->stream;
    const uint32_t max_chunks = (uint32_t)len / 275u + 2u;   // a chunk is >= 259 header bytes + 16 state bytes
    // pass 1: headers
    ChunkInfo *info;
    int32_t *freq;
    {
        Arena plan(ctx, true);
        plan.get<ChunkInfo>(max_chunks);
        plan.get<int32_t>((size_t)max_chunks * 256);
        JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    }
    size_t head_bytes;
    {
        Arena real(ctx, false);
        info = real.get<ChunkInfo>(max_chunks);
        freq = real.get<int32_t>((size_t)max_chunks * 256);
        head_bytes = ctx->arena_off;
    }
    JPK_LAUNCH(ctx, PROF_DEC_HEADERS, 0, k_dec_headers, dim3(1), dim3(64), d_in, (uint32_t)len, (uint64_t)out_cap, max_chunks, info, freq, ctx->d_mail);
    JPK_HIP(hipGetLastError());
    uint32_t mail[6];
    JPK_TRY(jpk_read_mail(ctx, mail, 6));
    if ((int32_t)mail[0] != 0) return (int32_t)mail[0];
    const uint32_t nch = mail[1];
    const uint64_t total_out = ((uint64_t)mail[3] << 32) | mail[2];
    const uint64_t total_rle = ((uint64_t)mail[5] << 32) | mail[4];
    ctx->stats.ans_chunks = nch;
    ctx->stats.ans_rle_symbols = (int64_t)total_rle;
    if (nch == 0) return JPK_OK;

    // pass 2 buffers: the arena may move when it grows, so the header tables are re-created if it does
    const size_t need = head_bytes + jpk_align(total_rle * 2 + 64) + jpk_align(total_out + 64) + 4096;
    if (need > ctx->arena_cap) {
        JPK_TRY(jpk_arena_ensure(ctx, need));
        Arena real(ctx, false);
        info = real.get<ChunkInfo>(max_chunks);
        freq = real.get<int32_t>((size_t)max_chunks * 256);
        JPK_LAUNCH(ctx, PROF_DEC_HEADERS, 0, k_dec_headers, dim3(1), dim3(64), d_in, (uint32_t)len, (uint64_t)out_cap, max_chunks, info, freq, ctx->d_mail);
    }
    uint16_t *rle = (uint16_t *)(ctx->arena + head_bytes);
    uint8_t *tmp = ctx->arena + head_bytes + jpk_align(total_rle * 2 + 64);
    uint32_t *status = ctx->d_mail + 8;
    JPK_HIP(hipMemsetAsync(status, 0, 4, st));
    JPK_HIP(hipMemsetAsync(d_out, 0, total_out, st));
    JPK_LAUNCH(ctx, PROF_DEC_RANS, 2 * total_rle, k_dec_rans, dim3(nch), dim3(64), d_in, (int64_t)len, info, rle, status);
    JPK_LAUNCH(ctx, PROF_DEC_RLE, total_rle, k_dec_rle, dim3(nch), dim3(1024), rle, info, d_out, status);
    JPK_LAUNCH(ctx, PROF_DEC_RANK, total_out, k_dec_rank, dim3(nch), dim3(64), d_out, info, freq, tmp, status);
    JPK_HIP(hipGetLastError());
    JPK_HIP(hipMemcpyAsync(d_out, tmp, total_out, hipMemcpyDeviceToDevice, st));
    JPK_HIP(hipMemcpyAsync(ctx->d_mail, status, 4, hipMemcpyDeviceToDevice, st));
    uint32_t stw = 0;
    JPK_TRY(jpk_read_mail(ctx, &stw, 1));
    if (stw) return JPK_E_CORRUPT;
    *out_len = (int32_t)total_out;
    return JPK_OK;
}

// Postcoder::Decode (rank.cpp:96-151) for one buffer (len <= 2^31), in place
int jpk_rank_decode_device(jpk_ctx *ctx, uint8_t *d_r, const int32_t *d_freq, int32_t len)
{
    if (len == 0) return JPK_OK;
    hipStream_t st = ctx->stream;
    Arena plan(ctx, true);
    plan.get<ChunkInfo>(1);
    plan.get<uint8_t>((size_t)len + 64);
    plan.get<int32_t>(256);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    ChunkInfo *info = real.get<ChunkInfo>(1);
    uint8_t *tmp = real.get<uint8_t>((size_t)len + 64);
    int32_t *hf = real.get<int32_t>(256);
    // validate sum(freq) == len on the host (rank.cpp:104-108)
    int32_t f[256];
    JPK_HIP(hipMemcpyAsync(f, d_freq, sizeof f, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipStreamSynchronize(st));
    int64_t tot = 0;
    for (int s = 0; s < 256; s++) { if (f[s] < 0) return JPK_E_CORRUPT; tot += f[s]; }
    if (tot != len) return JPK_E_CORRUPT;
    JPK_HIP(hipMemcpyAsync(hf, d_freq, sizeof f, hipMemcpyDeviceToDevice, st));
    ChunkInfo ci;
    memset(&ci, 0, sizeof ci);
    ci.olen = (uint32_t)len;
    JPK_HIP(hipMemcpyAsync(info, &ci, sizeof ci, hipMemcpyHostToDevice, st));
    JPK_HIP(hipStreamSynchronize(st));
    uint32_t *status = ctx->d_mail + 8;
    JPK_LAUNCH(ctx, PROF_DEC_RANK, 0, k_dec_rank, dim3(1), dim3(64), d_r, info, hf, tmp, status);
    JPK_HIP(hipGetLastError());
    JPK_HIP(hipMemcpyAsync(d_r, tmp, (size_t)len, hipMemcpyDeviceToDevice, st));
    JPK_HIP(hipStreamSynchronize(st));
    return JPK_OK;
}
