// ans_dec.hip -- Ans::Decode (ans.cpp:236-270) on gfx950.
//   header walk (ans.cpp:254-261, ReadHeader :287-302)           one wave per block (terminator ballot), chunk chain serial by format
//   rANS + model decode (Threaded_Decode, ans.cpp:30-92)          one wave per chunk; lanes = CDF entries
//   RLE0 decode (rle.cpp:52-74)                                   one workgroup per chunk, scan based
//   sorted-rank decode (rank.cpp:96-151)                          one wave per chunk; list positions 0..63 one per lane,
//                                                                 64..255 packed four to a lane
// Batches of blocks run ONE grid per serial kernel over the chunks of all blocks (longest chains first beyond 1024).
// The entropy decoder is sequential inside a chunk by construction of the format (shared byte pointer of the
// four rANS states, adaptive models, bucket hopping of the rank decoder); parallelism comes from the chunks.
#include <vector>

#include "ans_common.hpp"
#include "common.hpp"

using namespace jpk;

// v_writelane with a scalar value AND a scalar lane select needs the lane select in M0 (one SGPR per VALU instruction on gfx9);
// the asm statements that do so name M0 as clobbered, which is what keeps the compiler's own M0 users (the LDS-DMA top-ups) correct
#pragma clang diagnostic ignored "-Winline-asm"

namespace {


struct ChunkInfo {          // one per chunk, written by the header walk
    uint64_t in_off;        // payload offset in the block's input
    uint32_t clen, olen, rlen;
    uint32_t blk;           // block of the batch this chunk belongs to
    uint64_t out_off;       // offset in the block's decoded output
    uint64_t rle_off;       // offset in the block's packed rle buffer
};

// one block of a batch (jpk_dev_blocks_*): every serial kernel runs ONE grid over the chunks of all blocks, so the number of
// chains in flight does not depend on how many kernels the hardware queues run side by side
struct DecBlock {
    const uint8_t *in;      // Ans stream
    int64_t in_len;
    uint64_t out_cap;
    uint8_t *ranks;         // RLE0 decode writes the rank array here, the rank decoder reads it
    uint8_t *out;           // decoded bytes (the rank decoder's output)
    uint16_t *rle;          // packed RLE0 symbols
    uint32_t cbase;         // first global chunk index of the block
    uint32_t max_chunks;
};

// mail: [0]=status, [1]=nch, [2..3]=total out, [4..5]=total rle
// One wave.  The chunk chain is serial by format (chunk c+1 starts behind chunk c's payload, ans.cpp:254-261), but a header is
// not: it is 259 LEB128 values (256 frequencies, olen, clen, rlen; ReadHeader ans.cpp:287-302) and every value ENDS in the one
// byte with bit 7 set (utils.cpp:70-90).  The wave walks a header 64 bytes at a time: one ballot marks the terminators, their
// running count numbers the values, and the lane that holds a terminator decodes that value from the up to four bytes in
// front of it.
// One wave per block.  WRITE = false: count the chunks and total the sizes only (mail); WRITE = true: also fill info / freq at
// the block's global chunk base.  mail of block b: mail[8 b + ...].
template <bool WRITE>
__global__ __launch_bounds__(64) void k_dec_headers(const DecBlock *__restrict__ blocks, ChunkInfo *__restrict__ info_all, int32_t *__restrict__ freq_all,
                                                   uint32_t *__restrict__ mail_all)
{
    const DecBlock B = blocks[blockIdx.x];
    const uint8_t *__restrict__ in = B.in;
    const uint32_t len = (uint32_t)B.in_len, max_chunks = B.max_chunks;
    const uint64_t out_cap = B.out_cap;
    ChunkInfo *info = WRITE ? info_all + B.cbase : nullptr;
    int32_t *freq = WRITE ? freq_all + (size_t)B.cbase * 256 : nullptr;
    uint32_t *mail = mail_all + 8 * blockIdx.x;
    const int l = lane_id();
    const uint32_t C[4] = {127u, 16510u, 2113661u, 270549116u};
    uint64_t ip = 0, op = 0, rp = 0;
    uint32_t nch = 0;
    int status = 0;
    while (ip < len && status == 0) {
        if (nch >= max_chunks) { status = JPK_E_CORRUPT; break; }
        uint32_t nval = 0;                       // values decoded so far (wave-uniform)
        uint64_t pos = ip;                       // first byte of the current 64-byte window
        uint32_t olen = 0, clen = 0, rlen = 0;
        uint32_t fsum = 0;                       // per-lane partial sum of the frequencies
        bool bad = false;
        uint32_t since = 0;                      // bytes since the last terminator at the start of the window (a value has <= 5 bytes)
        while (nval < 259u) {
            if (pos >= len) { bad = true; break; }
            const uint64_t p = pos + (uint64_t)l;
            const uint32_t byte = (p < len) ? in[p] : 0u;
            const uint64_t term = __ballot(p < len && (byte & 0x80u));
            const uint32_t before = (uint32_t)__popcll(term & ((1ull << l) - 1ull));
            const uint32_t vidx = nval + before;                         // index of the value that ends at my byte
            if (((term >> l) & 1ull) && vidx < 259u) {
                // bytes of my value: from the byte after the previous terminator to me (at most 5)
                const uint64_t prevmask = term & ((1ull << l) - 1ull);
                const uint32_t nlead = prevmask ? (uint32_t)l - (63u - (uint32_t)__clzll((long long)prevmask)) - 1u : (uint32_t)l + since;
                if (nlead > 4u) bad = true;
                else {
                    uint32_t x = 0;
                    for (uint32_t k = nlead; k > 0; k--) x = (x << 7) | in[p - k];
                    x = (x << 7) | (byte & 0x7fu);
                    if (nlead > 0) x += C[nlead - 1];
                    if (vidx < 256u) {
                        if (x > (uint32_t)ANS_CHUNK) bad = true;
                        if (WRITE) freq[(size_t)nch * 256 + vidx] = (int32_t)x;
                        fsum += x;
                    } else if (vidx == 256u) olen = x;
                    else if (vidx == 257u) clen = x;
                    else rlen = x;
                }
            }
            const uint32_t nt = (uint32_t)__popcll(term);
            if (nval + nt >= 259u) {
                // the header ends at the terminator of value 258: position of the (259 - nval)-th terminator of this window
                uint64_t t = term;
                for (uint32_t k = 259u - nval - 1u; k > 0; k--) t &= t - 1ull;
                pos += (uint64_t)__builtin_ctzll(t) + 1ull;
                nval = 259u;
            } else {
                since = nt ? (uint32_t)__clzll((long long)term) : since + 64u;      // continuation bytes behind the window's last terminator
                if (since > 4u) { bad = true; break; }            // a value has at most four of them
                nval += nt;
                pos += 64;
            }
        }
        bad = __ballot(bad) != 0ull;
        // the three length fields were decoded by whichever lanes held their terminators
        olen = wave_sum(olen); clen = wave_sum(clen); rlen = wave_sum(rlen);
        const uint64_t total = (uint64_t)wave_sum(fsum);
        ip = pos;
        // rlen > olen: RLE0 never emits more symbols than bytes (a run of k binary digits stands for >= k zeros, every other symbol
        // for one byte), so RLE::decode would end in "rle mismatch!" (rle.cpp:73) -- and the buffers below are sized by olen
        if (bad || olen > (uint32_t)ANS_CHUNK || rlen > olen || ip > len || (uint64_t)clen > len - ip || clen < 16 ||
            total != (uint64_t)olen) {                                       // ans.cpp:297-298, rank.cpp:104-108
            status = JPK_E_CORRUPT;
            break;
        }
        if (op + olen > out_cap) { status = JPK_E_CAPACITY; break; }
        if (WRITE && l == 0) {
            ChunkInfo ci;
            ci.in_off = ip; ci.clen = clen; ci.olen = olen; ci.rlen = rlen; ci.blk = blockIdx.x; ci.out_off = op; ci.rle_off = rp;
            info[nch] = ci;
        }
        ip += clen; op += olen; rp += rlen;
        nch++;
    }
    if (l == 0) {
        mail[0] = (uint32_t)status;
        mail[1] = nch;
        mail[2] = (uint32_t)op; mail[3] = (uint32_t)(op >> 32);
        mail[4] = (uint32_t)rp; mail[5] = (uint32_t)(rp >> 32);
    }
}

// Launch order of the chains of a large batch: longest first (by RLE0 symbols, what both serial kernels' time follows), so that
// the chains that set the length of the stage are not the ones that had to wait for a free slot.  One pass of counting: 4096
// buckets of 256 symbols, longest bucket first; the order inside a bucket is arrival order (any order is a valid launch order:
// it only decides which chain starts first, never a byte).  One workgroup, any number of chunks.
constexpr int ORD_BUCKETS = 4096, ORD_TB = 1024;
__global__ __launch_bounds__(ORD_TB) void k_dec_order(const ChunkInfo *__restrict__ info, uint32_t n, uint32_t *__restrict__ order)
{
    __shared__ uint32_t h[ORD_BUCKETS];
    __shared__ uint32_t sm[ORD_TB / 64 + 1];
    for (int i = threadIdx.x; i < ORD_BUCKETS; i += ORD_TB) h[i] = 0;
    __syncthreads();
    auto bucket = [](uint32_t rlen) { const uint32_t b = rlen >> 8; return (uint32_t)(ORD_BUCKETS - 1) - (b < (uint32_t)ORD_BUCKETS ? b : (uint32_t)(ORD_BUCKETS - 1)); };
    for (uint32_t t = threadIdx.x; t < n; t += ORD_TB) atomicAdd(&h[bucket(info[t].rlen)], 1u);
    __syncthreads();
    // exclusive scan of the bucket counts, four consecutive buckets per thread
    uint32_t v[ORD_BUCKETS / ORD_TB], acc = 0;
#pragma unroll
    for (int k = 0; k < ORD_BUCKETS / ORD_TB; k++) { v[k] = h[threadIdx.x * (ORD_BUCKETS / ORD_TB) + k]; acc += v[k]; }
    const uint32_t inc = block_incl_scan<OpSum>(acc, sm, nullptr);
    uint32_t run = inc - acc;
#pragma unroll
    for (int k = 0; k < ORD_BUCKETS / ORD_TB; k++) { h[threadIdx.x * (ORD_BUCKETS / ORD_TB) + k] = run; run += v[k]; }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n; t += ORD_TB) order[atomicAdd(&h[bucket(info[t].rlen)], 1u)] = t;
}

// ---------------------------------------------------------------------------------------------------------------
// rANS + model decode: one wave per chunk (ans.cpp:30-92).
//
// The loop is one long dependent chain, so it is written for the issue costs of a single gfx950 wave (measured with
// tools/issuetest.hip): a VALU or SALU instruction issues every ~5 cycles (4.4 ns-ticks at 2.4 GHz: 5.1 independent, 5-6
// in a dependent chain), a SALU instruction that reads an SGPR a VALU instruction has just written stalls ~17 cycles
// more, an LDS round trip is 52-68 cycles, a branch ~14 (not taken) to ~24 (taken) cycles, and every value that is live
// across diverging arms costs a copy at the join.  Hence:
//   * the four rANS states, the byte queue and the two alphabet-2 models are wave-uniform and live in SGPRs;
//   * the 8-symbol exponent model and the mantissa models of classes 2..5 are register resident with "lane = symbol":
//     lane j holds LO_j = cdf[j], HI_j = cdf[j+1], FR_j = HI_j - LO_j of its symbol (class e owns lanes
//     [2^e, 2^(e+1)), which is exactly the RLE0 symbol numbering, tables.hpp:10).  All lanes form their candidate
//     next state FR_j * (x >> 16) + (x & 0xffff) - LO_j at once; one v_cmp (HI_j > x & 0xffff), one s_ff1 and one
//     v_readlane pick the coded symbol and its next state: no LDS, no search loop;
//   * the rebuild countdowns of classes 2..5 live in lanes 0..3 of one VGPR (lanes 4, 5 hold a permanent 1 for classes 6
//     and 7), the output tile is collected with v_writelane and stored every 64 symbols, and ONE scalar test per symbol
//     covers "an event is due" and "a state needs bytes";
//   * payload bytes come from a 64-bit SGPR queue refilled one dword at a time from a 256-byte window held across
//     the wave (lane l = dword l, next window prefetched).
// Classes 6 and 7 (64 and 129 symbols, rare outside incompressible data) keep their CDFs in one / three more registers.
// ---------------------------------------------------------------------------------------------------------------
// QuasiModel rebuild (model.cpp:160-204) of a register-resident model of A <= 64 * NR symbols: symbol i = lane l of
// register j (i = l + 64 j).  hi/lo/fr become the bounds and width of every symbol, the counts are cleared.
template <int NR>
__device__ __forceinline__ void quasi_rebuild_regs(int A, int l, uint32_t (&hi)[NR], uint32_t (&lo)[NR], uint32_t (&fr)[NR], uint32_t (&f)[NR],
                                                   uint32_t &expn)
{
    uint32_t F[NR];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < NR; j++) { F[j] = (l + 64 * j < A) ? f[j] : 0u; tot += F[j]; }
    tot = wave_sum(tot);
    int lg = 0;
    while ((tot >> lg) + (uint32_t)A > 65536u) lg++;
    uint32_t t2 = 0;
#pragma unroll
    for (int j = 0; j < NR; j++) { F[j] = (l + 64 * j < A) ? (F[j] >> lg) + 1u : 0u; t2 += F[j]; }
    t2 = wave_sum(t2);
    uint32_t t3 = 0;
#pragma unroll
    for (int j = 0; j < NR; j++) { F[j] = (65536u * F[j]) / t2; t3 += F[j]; }
    t3 = wave_sum(t3);
    if (l == 0) F[0] += 65536u - t3;
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const uint32_t inc = wave_incl_sum(F[j]);
        hi[j] = carry + inc;                         // symbols past the alphabet end at 65536 with width 0
        lo[j] = carry + inc - F[j];
        fr[j] = F[j];
        f[j] = 0;
        carry += __shfl(inc, 63, 64);
    }
    expn = (expn < 65536u) ? expn << 1 : 65536u;
}

// 256-byte input window held across the wave (lane l = dword l), next window prefetched, drained through a 64-bit
// scalar queue: a renormalisation byte is two scalar instructions away.
struct ByteQueue {
    const uint32_t *pw;       // dword-aligned base (payload start rounded down)
    int64_t gbase;            // byte offset of pw inside the whole input buffer
    int64_t in_len;
    const uint8_t *in;
    uint32_t cur, nxt;        // my dword of the current / next window
    uint32_t di;              // next dword of the stream to enter the queue
    uint64_t buf;             // queued bytes, next one in bits 7:0
    uint32_t nb;              // bytes in the queue
    uint32_t skip;            // bytes in front of the payload in the first dword (alignment)
    int l;
    __device__ __forceinline__ uint32_t load(uint32_t w) const
    {
        const int64_t g = gbase + ((int64_t)w * 64 + l) * 4;     // byte offset in the input buffer
        if (g >= 0 && g + 4 <= in_len) return pw[(size_t)w * 64 + l];
        uint32_t v = 0;
        for (int k = 0; k < 4; k++)
            if (g + k >= 0 && g + k < in_len) v |= (uint32_t)in[g + k] << (8 * k);
        return v;
    }
    __device__ __forceinline__ void refill()
    {
        if ((di & 63u) == 0u && di != 0u) { cur = nxt; nxt = load((di >> 6) + 1u); }      // uniform
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(di & 63u));
        buf |= (uint64_t)d << (8u * nb);
        nb += 4u;
        di++;
    }
    __device__ __forceinline__ void init(const uint8_t *input, int64_t input_len, const uint8_t *p, int lane)
    {
        in = input; in_len = input_len; l = lane;
        const uint32_t a = (uint32_t)((uintptr_t)p & 3u);
        pw = reinterpret_cast<const uint32_t *>(p - a);
        gbase = (p - a) - input;
        cur = load(0);
        nxt = load(1);
        di = 0; buf = 0; nb = 0; skip = a;
        refill();
        buf >>= 8u * a;
        nb -= a;
        refill();
    }
    __device__ __forceinline__ void top_up() { if (__builtin_expect(nb < 4u, 0)) refill(); }      // leaves nb >= 4
    // payload bytes consumed so far: what entered the queue minus what is still queued (not counted per byte)
    __device__ __forceinline__ uint32_t taken() const { return 4u * di - skip - nb; }
    __device__ __forceinline__ uint32_t take()
    {
        const uint32_t b = (uint32_t)buf & 0xffu;
        buf >>= 8;
        nb--;
        return b;
    }
};

__device__ __forceinline__ uint32_t dpp_row_shr1_zero(uint32_t v)     // lane j <- lane j-1 inside a row of 16, lane 0 of a row <- 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
}

#define JPK_ICMP_ULT 36     // llvm::CmpInst::ICMP_ULT

// Both states of a symbol are renormalised in the slow block: up to four bytes leave the queue, in stream order (exponent
// state first, ans.cpp:50-86), and one refill restores the >= 4 queued bytes the next symbol relies on.
#define JPK_RENORM2(X, X2)                                        \
    {                                                             \
        if ((X) < RANS_L) {                                       \
            (X) = ((X) << 8) | bq.take();                         \
            if (__builtin_expect((X) < RANS_L, 0)) (X) = ((X) << 8) | bq.take();       \
        }                                                         \
        if ((X2) < RANS_L) {                                      \
            (X2) = ((X2) << 8) | bq.take();                       \
            if (__builtin_expect((X2) < RANS_L, 0)) (X2) = ((X2) << 8) | bq.take();    \
        }                                                         \
        bq.top_up();                                              \
    }

// one RLE0 symbol: exponent from state RA, mantissa from state RB (the states rotate, ans.cpp:50-86); LANEI = its lane in
// the 64-symbol output tile.
// Instruction order matters on a single wave: the exponent compare goes first so that the vector work that does not need
// the exponent (candidates, mantissa compare) covers the wait of the scalar unit for its mask; the model updates sit where
// the scalar unit would otherwise wait for a v_readlane result; sched_barrier keeps the compiler from undoing that.
// A symbol pays one branch plus a quarter of the loop's: the slow block is entered when a model rebuild is due, when the
// class is 6/7 (their lanes of `rem` hold 1 permanently) or when min(x, x2) says that a state needs bytes.
#define JPK_DEC_SYMBOL(RA, RB, LANEI)                                                                     \
    {                                                                                                     \
        /* range = state & 0xffff and state >> 16 are taken straight from the scalar state by SDWA / op_sel operands:  \
           v_cmp(range < HI), range - LO, FR * (state >> 16) + (range - LO) for the exponent and the mantissa model.   \
           One block fixes the order (the SDWA results are consumed two instructions later; FR < 65536 fits u16) */  \
        uint64_t eabove, qabove;                                                                          \
        uint32_t et_, qt_, ecand, qcand;                                                                  \
        asm("v_cmp_lt_u32_sdwa %0, %6, %7 src0_sel:WORD_0 src1_sel:DWORD\n\t"                            \
            "v_sub_u32_sdwa %2, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t" \
            "v_cmp_lt_u32_sdwa %1, %9, %10 src0_sel:WORD_0 src1_sel:DWORD\n\t"                           \
            "v_sub_u32_sdwa %3, %9, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t" \
            "v_mad_u32_u16 %4, %12, %6, %2 op_sel:[0,1,0,0]\n\t"                                          \
            "v_mad_u32_u16 %5, %13, %9, %3 op_sel:[0,1,0,0]"                                              \
            : "=&s"(eabove), "=&s"(qabove), "=&v"(et_), "=&v"(qt_), "=&v"(ecand), "=&v"(qcand)            \
            : "s"(RA), "v"(ehi), "v"(elo), "s"(RB), "v"(qhi), "v"(qlo), "v"(efr), "v"(qfr));              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        const uint32_t e = (uint32_t)__builtin_ctz((uint32_t)eabove);                                     \
        /* symbol s = lane s; class e owns the lanes whose class is e (none for e >= 6: the event block takes over).           \
           The three lane masks that depend on e are formed first: their users below then sit far enough behind them that no  \
           wait state is needed between a vector compare and the vector instruction that takes its mask */            \
        const bool below = (uint32_t)l < e;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)ecand, (int)e);                             \
        const bool counts = cls_of_lane == e;                                                             \
        const bool mine = lane_cls == e;                                                                  \
        const uint64_t minemask = __ballot(mine);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        /* AdaptiveModel update (model.cpp:60-77): entry i = j + 1 lives in lane j */                     \
        const int32_t mix = below ? emix_lo : emix_hi;                                                    \
        ehi = (uint32_t)((int32_t)ehi + ((mix - (int32_t)ehi) >> 5));                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        /* countdowns: lane e - 2 counts the symbols of class e until its rebuild (classes 6, 7: always due) */ \
        rem -= counts ? 1u : 0u;                                                                          \
        const uint64_t due = __builtin_amdgcn_uicmp(rem, 0u, 32 /* ICMP_EQ */);                           \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        elo = dpp_row_shr1_zero(ehi);                                                                     \
        efr = ehi - elo;                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        /* for e >= 6 the mask is empty: s_ff1 returns -1, v_readlane takes lane 63 and no lane counts; the event block redoes both */ \
        uint32_t sym;                                                                                     \
        asm("s_ff1_i32_b64 %0, %1" : "=s"(sym) : "s"(qabove & minemask));                                 \
        const bool hit = (uint32_t)l == sym;                                                              \
        uint32_t x2 = (uint32_t)__builtin_amdgcn_readlane((int)qcand, (int)sym);                          \
        const int32_t tgt = (sym & 1u) ? 1 : 65535;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        {                                                                                                 \
            qf += hit ? 1u : 0u;                                                 /* QuasiModel count, in units of 16 */ \
            /* AdaptiveModel update of the alphabet-2 pair (lanes 2e, 2e+1) when e < 2: both lanes carry a = cdf[1] */ \
            const int32_t d = mine ? (tgt - (int32_t)qa) >> 5 : 0;                                        \
            qa += (uint32_t)d;                                                                            \
            qhi = (uint32_t)(__mul24(d, even_lane) + (int32_t)qhi);                                       \
            qlo = (uint32_t)(__mul24(d, odd_lane) + (int32_t)qlo);                                        \
            qfr = (uint32_t)(__mul24(d, sign_lane) + (int32_t)qfr);                                       \
        }                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        /* ONE test sends a symbol to the slow block: an event is due, or one of the two states needs bytes */ \
        uint32_t gate;                                       /* 0 when an event is due, else min(x, x2) */ \
        asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 0, %2" : "=s"(gate) : "s"(due), "s"(x < x2 ? x : x2) : "scc"); \
        if (__builtin_expect(gate < RANS_L, 0)) {                                                         \
          if (__builtin_expect(due != 0ull, 0)) {        /* most visits are renormalisations */           \
            const uint32_t range_m = (RB) & 0xffffu, xs_m = (RB) >> 16;                                   \
            if (e == 6u) {                                                                                \
                /* classes 6 (64 symbols) and 7 (129): same lane-per-symbol search over one / three registers */ \
                const uint32_t cand = __umul24(fr6[0], xs_m) + (range_m - lo6[0]);                        \
                const uint32_t m = (uint32_t)__builtin_ctzll(__builtin_amdgcn_uicmp(range_m, hi6[0], JPK_ICMP_ULT)); \
                x2 = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)m);                              \
                f6[0] += ((uint32_t)l == m) ? 16u : 0u;                                                   \
                sym = 64u + m;                                                                            \
                if (++seen6 > expn6) { quasi_rebuild_regs<1>(64, l, hi6, lo6, fr6, f6, expn6); seen6 = 0; } \
                rem = (l == 4) ? 1u : rem;                                                                \
            } else if (e == 7u) {                                                                         \
                const uint64_t a0 = __builtin_amdgcn_uicmp(range_m, hi7[0], JPK_ICMP_ULT);                \
                const uint64_t a1 = __builtin_amdgcn_uicmp(range_m, hi7[1], JPK_ICMP_ULT);                \
                uint32_t m;                                                                               \
                if (a0) {                                                                                 \
                    m = (uint32_t)__builtin_ctzll(a0);                                                    \
                    x2 = (uint32_t)__builtin_amdgcn_readlane((int)(__umul24(fr7[0], xs_m) + (range_m - lo7[0])), (int)m); \
                    f7[0] += ((uint32_t)l == m) ? 16u : 0u;                                               \
                } else if (a1) {                                                                          \
                    m = (uint32_t)__builtin_ctzll(a1);                                                    \
                    x2 = (uint32_t)__builtin_amdgcn_readlane((int)(__umul24(fr7[1], xs_m) + (range_m - lo7[1])), (int)m); \
                    f7[1] += ((uint32_t)l == m) ? 16u : 0u;                                               \
                    m += 64u;                                                                             \
                } else {                                             /* the 129th symbol: lane 0 of the third register */ \
                    m = 128u;                                                                             \
                    x2 = (uint32_t)__builtin_amdgcn_readlane((int)(__umul24(fr7[2], xs_m) + (range_m - lo7[2])), 0); \
                    f7[2] += (l == 0) ? 16u : 0u;                                                         \
                }                                                                                         \
                sym = 128u + m;                                                                           \
                if (++seen7 > expn7) { quasi_rebuild_regs<3>(129, l, hi7, lo7, fr7, f7, expn7); seen7 = 0; } \
                rem = (l == 5) ? 1u : rem;                                                                \
            } else {                                                                                      \
                /* QuasiModel rebuild (model.cpp:160-204) of class e (2..5) in its lanes [2^e, 2^(e+1)) */ \
                const uint32_t kl = e - 2u;                                                               \
                const uint32_t A = 1u << e;                                                               \
                const bool mine = ((uint32_t)l >> e) == 1u;      /* shadows the symbol's flag: same lanes */ \
                uint32_t F = mine ? qf << 4 : 0u;                                                         \
                const uint32_t tot = wave_sum(F);                                                         \
                int lg = 0;                                                                               \
                while ((tot >> lg) + A > 65536u) lg++;                                                    \
                F = mine ? (F >> lg) + 1u : 0u;                                                           \
                const uint32_t t2 = wave_sum(F);                                                          \
                F = (65536u * F) / t2;                                                                    \
                const uint32_t t3 = wave_sum(F);                                                          \
                if ((uint32_t)l == A) F += 65536u - t3;                 /* first symbol of the class */   \
                const uint32_t inc = wave_incl_sum(F);                                                    \
                qhi = mine ? inc : qhi; qlo = mine ? inc - F : qlo; qfr = mine ? F : qfr; qf = mine ? 0u : qf; \
                const uint32_t ex0 = (uint32_t)__builtin_amdgcn_readlane((int)qexpn, (int)(kl & 3u));     \
                const uint32_t ex1 = (ex0 < 65536u) ? ex0 << 1 : 65536u;                                  \
                qexpn = ((uint32_t)l == kl) ? ex1 : qexpn;                                                \
                rem = ((uint32_t)l == kl) ? ex1 + 1u : rem;                                               \
            }                                                                                             \
          }                                                                                               \
          JPK_RENORM2(x, x2)                                                                              \
        }                                                                                                 \
        (RA) = x;                                                                                         \
        (RB) = x2;                                                                                        \
        /* value and lane select are both scalar: the lane select goes through M0 (one SGPR per VALU instruction on gfx9) */ \
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mysym) : "s"(sym), "s"((uint32_t)(LANEI)) : "m0"); \
    }

__global__ __launch_bounds__(64) void k_dec_rans(const DecBlock *__restrict__ blocks, const ChunkInfo *__restrict__ info, const uint32_t *__restrict__ order,
                                                uint32_t *__restrict__ status_all)
{
    __builtin_amdgcn_s_setprio(3);         // serial chain: win the issue arbitration on a shared SIMD
    const uint32_t c = order ? order[blockIdx.x] : blockIdx.x;
    const int l = lane_id();
    const ChunkInfo ci = info[c];
    const DecBlock B = blocks[ci.blk];
    const uint8_t *__restrict__ in = B.in;
    const int64_t in_len = B.in_len;
    uint16_t *__restrict__ rle = B.rle;
    uint32_t *status = status_all + ci.blk;
    const uint8_t *p = in + ci.in_off;
    const uint32_t clen = ci.clen, rlen = ci.rlen;
    uint16_t *out = rle + ci.rle_off;

    ByteQueue bq;
    bq.init(in, in_len, p, l);
    // exponent model (alphabet 8): lane j holds LO = cdf[j], HI = cdf[j+1]; lanes >= 7 keep HI = 65536
    uint32_t elo = (l < 8) ? uniform_cdf(8, l) : 65536u;
    uint32_t ehi = (l < 8) ? uniform_cdf(8, l + 1) : 65536u;
    uint32_t efr = ehi - elo;
    // mix targets of entry i = l + 1 (model.cpp:60-77): i when i <= symbol, i + 65536 - 8 otherwise; fixed lanes aim at 65536
    const int32_t emix_lo = (l < 7) ? l + 1 : 65536;
    const int32_t emix_hi = (l < 7) ? l + 1 + 65536 - 8 : 65536;
    // mantissa models of classes 0..5: symbol s = lane s.  Lanes 0..3 hold the two alphabet-2 AdaptiveModels (class 0 =
    // symbols 0,1; class 1 = symbols 2,3; cdf[1] = 32768), lanes 4..63 the QuasiModels of classes 2..5 (uniform start).
    uint32_t qlo, qhi, qfr, qf = 0;
    const uint32_t lane_cls = (l < 2) ? 0u : (uint32_t)(31 - __clz(l));
    int32_t even_lane = (l < 4 && !(l & 1)) ? 1 : 0, odd_lane = (l < 4 && (l & 1)) ? 1 : 0;
    int32_t sign_lane = even_lane - odd_lane;
    // opaque to the optimiser: three v_mad_i32_i24 per symbol instead of select + add pairs
    asm volatile("" : "+v"(even_lane), "+v"(odd_lane), "+v"(sign_lane));
    uint32_t qa = (l < 4) ? 32768u : 0u;                          // lanes 0..3: cdf[1] of the pair's AdaptiveModel
    {
        const int A = (l < 4) ? 2 : 1 << lane_cls, i = (l < 4) ? (l & 1) : l - A;
        qlo = uniform_cdf(A, i);
        qhi = uniform_cdf(A, i + 1);
        qfr = qhi - qlo;
    }
    // classes 6 and 7: symbol i of the class = lane l of register i / 64 (uniform start, model.cpp:85-95)
    uint32_t hi6[1], lo6[1], fr6[1], f6[1] = {0}, seen6 = 0, expn6 = 8;
    uint32_t hi7[3], lo7[3], fr7[3], f7[3] = {0, 0, 0}, seen7 = 0, expn7 = 8;
    lo6[0] = uniform_cdf(64, l); hi6[0] = uniform_cdf(64, l + 1); fr6[0] = hi6[0] - lo6[0];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int i = l + 64 * j;
        lo7[j] = (i < 129) ? uniform_cdf(129, i) : 65536u;
        hi7[j] = (i < 129) ? uniform_cdf(129, i + 1) : 65536u;
        fr7[j] = hi7[j] - lo7[j];
    }
    uint32_t qexpn = 8;                                          // lane k: EXP of class k + 2 (model.cpp:160-204)
    // lane k < 4: symbols until the rebuild of class k + 2; lanes 4, 5: classes 6, 7 (always due: they decode in the event block)
    uint32_t rem = (l < 4) ? 9u : ((l < 6) ? 1u : 0x40000000u);
    const uint32_t cls_of_lane = (l < 6) ? (uint32_t)l + 2u : 99u;   // lane k counts down for class k + 2
    uint32_t R0, R1, R2, R3;
    {
        uint32_t b[16];
        for (int k = 0; k < 16; k++) { b[k] = bq.take(); bq.top_up(); }
        R0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        R1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        R2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
        R3 = b[12] | (b[13] << 8) | (b[14] << 16) | (b[15] << 24);
    }
    uint32_t mysym = 0;
    uint32_t t = 0;
    // whole tiles of 64 symbols: the tile is collected with v_writelane and stored once
    for (; t + 64 <= rlen; t += 64) {
        uint32_t j = 0;
        do {
            JPK_DEC_SYMBOL(R0, R1, j)
            JPK_DEC_SYMBOL(R2, R3, j + 1)
            JPK_DEC_SYMBOL(R0, R1, j + 2)
            JPK_DEC_SYMBOL(R2, R3, j + 3)
            j += 4;
        } while (j < 64);
        out[t + l] = (uint16_t)mysym;
    }
    {
        // the last, partial tile one symbol at a time; the state pairs swap instead of a second copy of the symbol code (the
        // final check below looks at all four states, whichever names they ended up under)
        const uint32_t left = rlen - t;
        for (uint32_t j = 0; j < left; j++) {
            JPK_DEC_SYMBOL(R0, R1, j)
            uint32_t sw = R0; R0 = R2; R2 = sw;
            sw = R1; R1 = R3; R3 = sw;
        }
        if ((uint32_t)l < left) out[t + l] = (uint16_t)mysym;
    }
    // ans.cpp:91-92 (all four states back at the lower bound) and no byte taken from beyond the chunk's payload
    const bool bad = (R0 != RANS_L || R1 != RANS_L || R2 != RANS_L || R3 != RANS_L) || bq.taken() > clen;
    if (bad && l == 0) atomicOr(status, 1u);
}
#undef JPK_DEC_SYMBOL
#undef JPK_RENORM2

// ---------------------------------------------------------------------------------------------------------------
// RLE0 decode: one workgroup per chunk.  Output is pre-zeroed, so only symbols > 1 are written; a digit group
// contributes value-1 zeros at the position of its last digit (rle.cpp:52-74).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_dec_rle(const DecBlock *__restrict__ blocks, const ChunkInfo *__restrict__ info, uint32_t *__restrict__ status_all)
{
    const uint32_t c = blockIdx.x;
    const ChunkInfo ci = info[c];
    const DecBlock B = blocks[ci.blk];
    const uint16_t *__restrict__ rle = B.rle;
    uint8_t *__restrict__ out = B.ranks;
    uint32_t *status = status_all + ci.blk;
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
// list: positions 0..63 are one symbol per lane of register L0, so that the common update -- insert at a rank below 64 --
// is one DPP move, one select and one v_writelane.  Positions 64..255 are packed four to a lane in register P (lane l =
// positions 64 + 4l .. 67 + 4l, little endian; lanes 48.. unused): ranks >= 64 (incompressible data) pay a full shift of L0
// plus one branch-free byte shift-insert of P.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v)        // lane i <- lane i+1, lane 63 <- 0 (no user reads it): one DPP move
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);      // bound_ctrl: no tied old value, no copy
}

__device__ __forceinline__ uint32_t lane_write(uint32_t v, uint32_t value, uint32_t lane)      // v with lane `lane` = value (both wave-uniform)
{
    // value and lane select are both scalar: the lane select goes through M0 (one SGPR per VALU instruction on gfx9)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(lane) : "m0");
    return v;
}

// packed register: byte positions < r take the value of position + 1; INSERT: position r takes sym.  Branch-free:
// keep = bytes of this lane at or above position r (two half shifts, so that 4 bytes below r give 0).
template <bool INSERT>
__device__ __forceinline__ uint32_t packed_shift(uint32_t v, int l, uint32_t r, uint32_t sym)
{
    const uint32_t nextv = wave_shl1(v);
    const uint32_t shifted = __builtin_amdgcn_alignbyte(nextv, v, 1);      // (v >> 8) | (nextv << 24)
    const int nb = (int)r - 4 * l;                                          // bytes of this lane below position r
    const uint32_t h = 4u * (uint32_t)(nb < 0 ? 0 : (nb > 4 ? 4 : nb));
    const uint32_t keep = (0xFFFFFFFFu << h) << h;
    uint32_t res = (v & keep) | (shifted & ~keep);
    if (INSERT) {
        const uint32_t ins = ((uint32_t)nb < 4u) ? 0xFFu << (8u * ((uint32_t)nb & 3u)) : 0u;
        res = (res & ~ins) | ((sym * 0x01010101u) & ins);
    }
    return res;
}

// positions < r take the value of position + 1; INSERT: position r takes sym (rank.cpp:131-134), else it keeps its value
// (the front drop of an exhausted bucket, rank.cpp:140-147).  r <= 255.
template <bool INSERT>
__device__ __forceinline__ void list_shift(uint32_t &L0, uint32_t &P, int l, uint32_t r, uint32_t sym)
{
    const uint32_t nxt = wave_shl1(L0);
    if (r < 64u) {
        const uint32_t res = ((uint32_t)l < r) ? nxt : L0;
        L0 = INSERT ? lane_write(res, sym, r) : res;
    } else {
        L0 = lane_write(nxt, rfl(P) & 0xffu, 63u);          // position 63 takes position 64 = byte 0 of P's lane 0
        P = packed_shift<INSERT>(P, l, r - 64u, sym);
    }
}

// Every symbol keeps the next 64 ranks of its bucket in an LDS row, one rank per byte (row[used] = the rank its next
// occurrence reads), so the serial chain never waits for HBM: a run costs one LDS round trip, one ballot and one list
// update.  A row that runs low is topped up by a direct-to-LDS load (global_load_lds_ubyte: lane j's byte lands
// zero-extended in dword j behind the M0 base, no VGPR and therefore no compiler-inserted wait).  Entries past the end of
// a bucket are 0xFF (non-zero: they end the zero run).
//
// The inner loop is the common iteration -- a non-zero rank found inside the safe part of the row, at least 64 bytes of
// output left -- and touches one counter of the row's metadata; everything else (end of a bucket, empty or low rows,
// landing a top-up, the last bytes of the chunk) goes through the general path, which also re-normalises the metadata.
struct RankMeta {
    uint32_t used;      // entries consumed by fast iterations since the last normalisation (the row is read from here on)
    uint32_t fast;      // fast iterations may consume this many entries (as of the last normalisation)
    uint32_t nv;        // known entries in the row (as of the last normalisation)
    uint32_t left;      // ranks of the bucket not yet consumed (as of the last normalisation)
};
struct RankSpan {
    uint32_t gpos;      // offset in R of the first byte not yet requested
    uint32_t gend;      // end of the bucket in R
};
constexpr uint32_t RANK_LOW = 24;          // top a row up when fewer known entries remain

typedef const __attribute__((address_space(1))) void *jpk_gptr;
typedef __attribute__((address_space(3))) void *jpk_lptr;
typedef __attribute__((address_space(1))) uint8_t *jpk_gbytes;

__device__ __forceinline__ uint32_t rank_fast_limit(uint32_t nv, uint32_t left, bool all_loaded)
{
    const uint32_t slack = all_loaded ? 64u : (nv > RANK_LOW ? nv - RANK_LOW : 0u);
    const uint32_t a = nv < left ? nv : left;
    return a < slack ? a : slack;
}

__global__ __launch_bounds__(64) void k_dec_rank(const DecBlock *__restrict__ blocks, const ChunkInfo *__restrict__ info, const uint32_t *__restrict__ order,
                                                const int32_t *__restrict__ freq, uint32_t *__restrict__ status_all)
{
    __builtin_amdgcn_s_setprio(3);         // serial chain: win the issue arbitration on a shared SIMD
    const uint32_t c = order ? order[blockIdx.x] : blockIdx.x;
    const int l = lane_id();
    const ChunkInfo ci = info[c];
    const uint32_t len = rfl(ci.olen);
    if (len == 0) return;
    const DecBlock B = blocks[ci.blk];
    const uint8_t *R = B.ranks + ci.out_off;     // rank array
    uint8_t *T = B.out + ci.out_off;             // decoded symbols
    const jpk_gbytes Tg = (jpk_gbytes)T;
    __shared__ uint8_t rows[256][64];             // one byte per rank: 16 KiB
    __shared__ uint32_t stage[64];                // landing zone of the top-up in flight (LDS-DMA writes one dword per lane)
    __shared__ RankMeta meta[256];
    __shared__ RankSpan span[256];
    // 22 784 bytes in all: seven chains per CU (160 KiB).  The frequency table and the start list are only needed until the rows
    // are filled and live in the rows' own memory until then (with their own 2 KiB the kernel took 24.3 KiB: six per CU).
    uint32_t *const sf = reinterpret_cast<uint32_t *>(&rows[0][0]);
    uint32_t *const lst = sf + 256;
    const int32_t *fq = freq + (size_t)c * 256;
    for (int s = l; s < 256; s += 64) { sf[s] = (uint32_t)fq[s]; lst[s] = 0; }
    __syncthreads();
    uint32_t uniq = 0;
    // bucket layout in GenerateSortedMap order; list[R[bucket start]] = symbol (rank.cpp:114-123)
    uint32_t g4[4], e4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = l + 64 * k;
        const uint32_t f = sf[s];
        uint32_t b = 0;
        for (int j = 0; j < 256; j++) { uint32_t fj = sf[j]; if (fj > f || (fj == f && j < s)) b += fj; }
        RankMeta m;
        m.used = 0; m.nv = 64; m.left = f ? f - 1 : 0;
        m.fast = rank_fast_limit(m.nv, m.left, b + 1 + 64 >= b + f);
        meta[s] = m;
        RankSpan sp;
        sp.gpos = b + 1 + 64; sp.gend = b + f;
        span[s] = sp;
        g4[k] = b + 1; e4[k] = b + f;
        if (f > 0) lst[R[b]] = (uint32_t)s;           // a corrupt stream may name a position twice: the last writer wins, as in rank.cpp
        uniq += (f > 0);
    }
    uniq = rfl(wave_sum(uniq));
    __syncthreads();
    uint32_t L0 = lst[l];
    uint32_t P = (l < 48) ? (lst[64 + 4 * l] | (lst[65 + 4 * l] << 8) | (lst[66 + 4 * l] << 16) | (lst[67 + 4 * l] << 24)) : 0u;
    __syncthreads();                               // the start list has been read: its memory becomes rows
    // first window of every bucket (the offsets come from registers so the 256 loads pipeline)
#pragma unroll
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 64; j++) {
            const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)g4[k], j), ge = (uint32_t)__builtin_amdgcn_readlane((int)e4[k], j);
            rows[j + 64 * k][l] = (g + l < ge) ? R[g + l] : (uint8_t)0xFF;
        }
    __syncthreads();
    uint32_t sym = (uint32_t)__builtin_amdgcn_readlane((int)L0, 0);
    uint32_t psym = 256;                                         // row with a top-up in flight (256 = none)
    uint32_t poff = 0, pcnt = 0;                                 // ... its entries land at rows[psym][poff .. poff + pcnt)
    uint32_t i = 0;
    const uint32_t fast_end = rfl(len >= 64u ? len - 63u : 0u);  // the inner loop stores 64 bytes at i: it runs while i < fast_end
    uint32_t rl = rows[sym][l];                                  // entry `used` is the next unread rank (no wrap between normalisations)
    RankMeta m = meta[sym];
    for (;;) {
        // ---- inner loop: z zero ranks, then the real non-zero rank; z + 1 outputs.  The row itself is not touched: these
        // iterations only advance `used` (they never look past the entries known at the last normalisation).  The row and
        // metadata of the NEXT symbol are read before the list update: a run that ends in a non-zero rank always brings the
        // second list entry to the front, whatever the rank is.
        uint32_t big_r = 0, big_sym = 0;                        // a rank >= 64 leaves the loop for its insert (L0 and P), so that
        for (;;) {                                               // the loop itself carries L0 only
            const uint32_t used = rfl(m.used), room = rfl(m.fast) - used;
            const uint64_t nz = __ballot(rl != 0) >> (used & 63u);
            uint32_t z;
            asm("s_ff1_i32_b64 %0, %1" : "=s"(z) : "s"(nz));    // no non-zero entry: -1, which fails the test below
            uint32_t lim;                                        // room, or 0 behind fast_end (select on the scalar unit, no mask logic)
            asm("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, %3, 0" : "=s"(lim) : "s"(i), "s"(fast_end), "s"(room) : "scc");
            if (!(z < lim)) break;
            const uint32_t nsym = (uint32_t)__builtin_amdgcn_readlane((int)L0, 1);
            // the rank that ends the run is taken out of the row BEFORE the next row is loaded: the old row's register is dead by
            // then and the load can land in it (no copy per iteration)
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rl, (int)(used + z));
            const uint32_t nrl = rows[nsym][l];
            const RankMeta nm = meta[nsym];
            // all 64 lanes store: the bytes behind the run are rewritten by the runs that own them (same wave, program order)
            // (issued as asm: the store is never waited for -- nothing in this kernel reads T -- and the compiler's wait-count
            // bookkeeping for the LDS-DMA top-ups would otherwise put a vmcnt(0) in front of the next LDS read)
            asm volatile("global_store_byte %0, %1, %2" : : "v"(i + (uint32_t)l), "v"(sym), "s"(Tg));
            const uint32_t cnt = z + 1u;
            i += cnt;
            meta[sym].used = used + cnt;                         // every lane stores the same word: cheaper than masking to one lane
            const uint32_t cur = sym;
            sym = nsym; rl = nrl; m = nm;
            if (__builtin_expect(r >= 64u, 0)) { big_r = r; big_sym = cur; break; }
            const uint32_t nxt = wave_shl1(L0);
            L0 = lane_write(((uint32_t)l < r) ? nxt : L0, cur, r);
        }
        if (big_r) { list_shift<true>(L0, P, l, big_r, big_sym); continue; }
        if (i >= len) break;
        // ---- general path ----
        if (psym < 256u) {                                       // land the top-up in flight and unblock its row
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((uint32_t)l < pcnt) rows[psym][poff + l] = (uint8_t)stage[l];
            const RankMeta pm = meta[psym];
            const RankSpan ps = span[psym];
            if (l == 0) { meta[psym].nv = 64u; meta[psym].fast = rank_fast_limit(64u, pm.left, ps.gpos >= ps.gend); }
            psym = 256u;
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
            rl = rows[sym][l];                                   // start over with the rows as they are now
            m = meta[sym];
            continue;
        }
        const uint32_t used = rfl(m.used);
        const uint32_t rest = len - i;
        const RankSpan sp = span[sym];
        const uint32_t gpos = rfl(sp.gpos), gend = rfl(sp.gend);
        const uint32_t nv = rfl(m.nv) - used, left = rfl(m.left) - used;
        const uint32_t rlog = rows[sym][(l + used) & 63u];                   // logical order from here on
        const uint32_t rlk = ((uint32_t)l < nv) ? rlog : 0xFFu;             // unknown entries stop the scan
        const uint64_t nzk = __ballot(rlk != 0);
        const uint32_t zk = nzk ? (uint32_t)__builtin_ctzll(nzk) : 64u;
        const bool stop = zk < nv;                                           // a known non-zero entry ended the run
        uint32_t cnt = stop ? zk + 1u : zk;                                  // outputs = ranks consumed
        if (cnt > rest) cnt = rest;
        if ((uint32_t)l < cnt) T[i + l] = (uint8_t)sym;
        i += cnt;
        // normalise: the unread entries move to the front of the row, unknown entries and the tail become 0xFF
        const uint32_t shifted = ((uint32_t)l + cnt < nv) ? rows[sym][(l + used + cnt) & 63u] : 0xFFu;
        const bool all_loaded = gpos >= gend;
        const uint32_t nv2 = all_loaded ? 64u : nv - (cnt < nv ? cnt : nv);
        const uint32_t left2 = left - (cnt < left ? cnt : left);
        const uint32_t cur = sym;
        if (stop) {
            if (zk < left) {
                const uint32_t r = __builtin_amdgcn_readlane(rlk, zk);       // the non-zero rank that ends the run
                list_shift<true>(L0, P, l, r, sym);
                sym = (uint32_t)__builtin_amdgcn_readlane((int)L0, 0);
            } else if (uniq > 0) {                                           // bucket exhausted: drop the front
                uniq--;
                const uint32_t lim = uniq > 0 ? uniq : 1u;                  // rank.cpp:140-147; executes at least once
                list_shift<false>(L0, P, l, lim, 0u);
                sym = (uint32_t)__builtin_amdgcn_readlane((int)L0, 0);
            }
        }
        rows[cur][l] = (uint8_t)shifted;
        RankMeta w;
        w.used = 0; w.nv = nv2; w.left = left2;
        if (!all_loaded && nv2 < RANK_LOW + 8u) {                            // top the row up behind the shifted entries
            const uint32_t want = 64u - nv2;
            __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the shifted row is in LDS before the load may land
            const uint32_t avail = gend - gpos;                              // gpos < gend here
            pcnt = want < avail ? want : avail;
            poff = nv2;
            if ((uint32_t)l < pcnt)
                __builtin_amdgcn_global_load_lds((jpk_gptr)(R + gpos + l), (jpk_lptr)(&stage[0]), 1, 0, 0);
            psym = cur;
            // the known entries stay readable while the top-up is in flight (it lands behind them): the row only stops the
            // inner loop when they run out, and by then the load has had the time of ~24 of this symbol's runs
            w.fast = rank_fast_limit(nv2, left2, true);
            if (l == 0) span[cur].gpos = gpos + want;
        } else {
            w.fast = rank_fast_limit(nv2, left2, all_loaded);
        }
        if (l == 0) meta[cur] = w;
        __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): the normalised row and its metadata are in LDS
        rl = rows[sym][l];
        m = meta[sym];
    }
}

}  // namespace

// Ans::Decode of `nblk` independent blocks in one pass: header walk (count), buffers, header walk (fill), then ONE grid per
// serial kernel over the chunks of all blocks.  status[b] = JPK_OK / JPK_E_CORRUPT / JPK_E_CAPACITY per block; a corrupt block
// does not stop the others.  arena_skip: bytes at the start of the arena the caller keeps for itself.
int jpk_ans_decode_batch(jpk_ctx *ctx, int nblk, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out, const int32_t *out_cap,
                         int32_t *out_len, int32_t *status, size_t arena_skip)
{
    hipStream_t st = ctx->stream;
    std::vector<DecBlock> hb((size_t)nblk);
    for (int b = 0; b < nblk; b++) {
        out_len[b] = 0;
        status[b] = JPK_OK;
        memset(&hb[b], 0, sizeof(DecBlock));
        hb[b].in = d_in[b];
        hb[b].in_len = in_len[b];
        hb[b].out_cap = (uint64_t)out_cap[b];
        hb[b].max_chunks = (uint32_t)in_len[b] / 275u + 2u;      // a chunk is >= 259 header bytes + 16 state bytes
    }
    // ---- pass 1: count the chunks, total the sizes ----
    const size_t tab_bytes = jpk_align((size_t)nblk * sizeof(DecBlock) + 64), mail_bytes = jpk_align((size_t)nblk * 8 * 4 + 64);
    if (arena_skip && !jpk_arena_fits(ctx, arena_skip + tab_bytes + mail_bytes + 4096)) return JPK_E_ALLOC;
    JPK_TRY(jpk_arena_ensure(ctx, arena_skip + tab_bytes + mail_bytes + 4096));
    DecBlock *d_tab = reinterpret_cast<DecBlock *>(ctx->arena + arena_skip);
    uint32_t *d_mail = reinterpret_cast<uint32_t *>(ctx->arena + arena_skip + tab_bytes);
    JPK_HIP(hipMemcpyAsync(d_tab, hb.data(), (size_t)nblk * sizeof(DecBlock), hipMemcpyHostToDevice, st));
    JPK_LAUNCH(ctx, PROF_DEC_HEADERS, 0, (k_dec_headers<false>), dim3(nblk), dim3(64), d_tab, (ChunkInfo *)nullptr, (int32_t *)nullptr, d_mail);
    JPK_HIP(hipGetLastError());
    std::vector<uint32_t> mail((size_t)nblk * 8);
    JPK_HIP(hipMemcpyAsync(mail.data(), d_mail, mail.size() * 4, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipStreamSynchronize(st));
    if (ctx->prof_on) jpk_prof_resolve(ctx);
    uint64_t nch_total = 0, rle_total = 0, out_total = 0;
    std::vector<uint64_t> tot_out((size_t)nblk), tot_rle((size_t)nblk);
    std::vector<uint32_t> nchb((size_t)nblk);
    for (int b = 0; b < nblk; b++) {
        const uint32_t *m = &mail[(size_t)b * 8];
        if ((int32_t)m[0] != 0) { status[b] = (int32_t)m[0]; hb[b].in_len = 0; nchb[b] = 0; tot_out[b] = tot_rle[b] = 0; hb[b].cbase = (uint32_t)nch_total; continue; }
        nchb[b] = m[1];
        tot_out[b] = ((uint64_t)m[3] << 32) | m[2];
        tot_rle[b] = ((uint64_t)m[5] << 32) | m[4];
        hb[b].cbase = (uint32_t)nch_total;
        hb[b].max_chunks = nchb[b] + 1u;
        nch_total += nchb[b];
        rle_total += tot_rle[b];
        out_total += tot_out[b];
    }
    ctx->stats.ans_chunks = (int64_t)nch_total;
    ctx->stats.ans_rle_symbols = (int64_t)rle_total;
    if (nch_total == 0) return JPK_OK;

    // ---- buffers: block table, mail, status, chunk table, frequencies; per block the packed RLE0 symbols and the rank array ----
    size_t off = arena_skip + tab_bytes + mail_bytes;
    auto take = [&](size_t bytes) { size_t o = off; off += jpk_align(bytes + 64); return o; };
    const size_t o_status = take((size_t)nblk * 4), o_info = take((size_t)nch_total * sizeof(ChunkInfo)), o_freq = take((size_t)nch_total * 256 * 4);
    const size_t o_order = take((size_t)nch_total * 4);
    std::vector<size_t> o_rle((size_t)nblk), o_ranks((size_t)nblk);
    for (int b = 0; b < nblk; b++) { o_rle[b] = take(tot_rle[b] * 2); o_ranks[b] = take(tot_out[b]); }
    // A caller that keeps buffers of its own at the start of the arena (arena_skip != 0) has sized it for the worst case of the
    // streams' declared sizes; an arena that would have to grow -- and therefore move -- under those buffers is refused.
    if (arena_skip && !jpk_arena_fits(ctx, off + 4096)) return JPK_E_ALLOC;
    JPK_TRY(jpk_arena_ensure(ctx, off + 4096));            // may move the arena: every pointer is formed below
    d_tab = reinterpret_cast<DecBlock *>(ctx->arena + arena_skip);
    d_mail = reinterpret_cast<uint32_t *>(ctx->arena + arena_skip + tab_bytes);
    uint32_t *d_status = reinterpret_cast<uint32_t *>(ctx->arena + o_status);
    ChunkInfo *info = reinterpret_cast<ChunkInfo *>(ctx->arena + o_info);
    int32_t *freq = reinterpret_cast<int32_t *>(ctx->arena + o_freq);
    for (int b = 0; b < nblk; b++) {
        hb[b].rle = reinterpret_cast<uint16_t *>(ctx->arena + o_rle[b]);
        hb[b].ranks = ctx->arena + o_ranks[b];
        hb[b].out = d_out[b];
    }
    JPK_HIP(hipMemcpyAsync(d_tab, hb.data(), (size_t)nblk * sizeof(DecBlock), hipMemcpyHostToDevice, st));
    JPK_HIP(hipMemsetAsync(d_status, 0, (size_t)nblk * 4, st));
    // the rank arrays start as zeros: RLE0 decode writes only the non-zero ranks (k_dec_rle); they are contiguous in the arena
    JPK_HIP(hipMemsetAsync(ctx->arena + o_ranks[0], 0, off - o_ranks[0], st));
    // ---- pass 2: fill the chunk table, then one grid per stage over all chunks ----
    JPK_LAUNCH(ctx, PROF_DEC_HEADERS, 0, (k_dec_headers<true>), dim3(nblk), dim3(64), d_tab, info, freq, d_mail);
    // The two serial kernels are one wave per chunk.  While the chains fit one per SIMD (<= 1024) a 40 KB LDS reservation keeps
    // them to four workgroups per CU; larger batches run without it, because two chains on a SIMD fill each other's issue
    // bubbles and a second round of workgroups would wait for the first (64 blocks: 3.5 -> 4.8 GB/s, tools/dec_scaling.py).
    const unsigned g = (unsigned)nch_total;
    static const int lds_env = [] { const char *e = getenv("JPK_DEC_LDS"); if (!e) return -1; const int v = atoi(e); return v < 0 ? 0 : (v > 65536 ? 65536 : v); }();
    const size_t lds_cap = (size_t)(lds_env >= 0 ? lds_env : (g <= 1024u ? 40960 : 0));
    // more chains than run at once: longest first
    uint32_t *order = nullptr;
    if (g > 1024u) {
        order = reinterpret_cast<uint32_t *>(ctx->arena + o_order);
        JPK_LAUNCH(ctx, PROF_DEC_HEADERS, 0, k_dec_order, dim3(1), dim3(ORD_TB), info, g, order);
    }
    JPK_LAUNCH_LDS(ctx, PROF_DEC_RANS, 2 * rle_total, lds_cap, k_dec_rans, dim3(g), dim3(64), d_tab, info, order, d_status);
    JPK_LAUNCH(ctx, PROF_DEC_RLE, rle_total, k_dec_rle, dim3(g), dim3(1024), d_tab, info, d_status);
    JPK_LAUNCH_LDS(ctx, PROF_DEC_RANK, out_total, (lds_cap > 22784 ? lds_cap - 22784 : 0), k_dec_rank, dim3(g), dim3(64), d_tab, info, order, freq, d_status);
    JPK_HIP(hipGetLastError());
    std::vector<uint32_t> hs((size_t)nblk);
    JPK_HIP(hipMemcpyAsync(hs.data(), d_status, (size_t)nblk * 4, hipMemcpyDeviceToHost, st));
    JPK_HIP(hipStreamSynchronize(st));
    if (ctx->prof_on) jpk_prof_resolve(ctx);
    for (int b = 0; b < nblk; b++) {
        if (status[b] != JPK_OK) continue;
        if (hs[b]) { status[b] = JPK_E_CORRUPT; continue; }
        out_len[b] = (int32_t)tot_out[b];
    }
    return JPK_OK;
}

int jpk_ans_decode_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    *out_len = 0;
    if (len == 0) return JPK_OK;
    int32_t status = JPK_OK;
    JPK_TRY(jpk_ans_decode_batch(ctx, 1, &d_in, &len, &d_out, &out_cap, out_len, &status, 0));
    return status;
}

// Postcoder::Decode (rank.cpp:96-151) for one buffer (len <= 2^31), in place
int jpk_rank_decode_device(jpk_ctx *ctx, uint8_t *d_r, const int32_t *d_freq, int32_t len)
{
    if (len == 0) return JPK_OK;
    hipStream_t st = ctx->stream;
    Arena plan(ctx, true);
    plan.get<ChunkInfo>(1);
    plan.get<DecBlock>(1);
    plan.get<uint8_t>((size_t)len + 64);
    plan.get<int32_t>(256);
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    ChunkInfo *info = real.get<ChunkInfo>(1);
    DecBlock *tab = real.get<DecBlock>(1);
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
    DecBlock hb;
    memset(&hb, 0, sizeof hb);
    hb.ranks = d_r;
    hb.out = tmp;
    JPK_HIP(hipMemcpyAsync(info, &ci, sizeof ci, hipMemcpyHostToDevice, st));
    JPK_HIP(hipMemcpyAsync(tab, &hb, sizeof hb, hipMemcpyHostToDevice, st));
    JPK_HIP(hipStreamSynchronize(st));
    uint32_t *status = ctx->d_mail + 8;
    JPK_LAUNCH(ctx, PROF_DEC_RANK, 0, k_dec_rank, dim3(1), dim3(64), tab, info, (const uint32_t *)nullptr, hf, status);
    JPK_HIP(hipGetLastError());
    JPK_HIP(hipMemcpyAsync(d_r, tmp, (size_t)len, hipMemcpyDeviceToDevice, st));
    JPK_HIP(hipStreamSynchronize(st));
    return JPK_OK;
}
