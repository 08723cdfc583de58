// common.hpp -- context, HBM arena and launch helpers shared by all translation units of libjampack_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/jampack_abi.h"

#define JPK_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            if (getenv("JPK_VERBOSE"))                                                                  \
                fprintf(stderr, "[jampack_amd] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e),       \
                        __FILE__, __LINE__, #expr);                                                     \
            return (_e == hipErrorOutOfMemory) ? JPK_E_ALLOC : JPK_E_DEVICE;                            \
        }                                                                                               \
    } while (0)

#define JPK_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != JPK_OK) return _rc; \
    } while (0)

// HBM arena: one allocation per context, bump-allocated per call, grown (free + malloc) when a bigger
// block arrives.  Replaces the reference's five cudaMalloc/cudaFree per block (bwt.cpp:195-239).
// per-kernel HIP-event timing (bench.py roofline): ids index jpk_prof_name()
enum JpkProfId {
    PROF_RS_HIST = 0, PROF_RS_SCATTER, PROF_SCAN, PROF_SA_KEYS, PROF_SA_SEG, PROF_SA_RERANK, PROF_BWT_GATHER,
    PROF_INV_HIST, PROF_INV_BUILD, PROF_INV_WALK, PROF_INV_RANK, PROF_INV_COPY,
    PROF_ENC_HIST, PROF_ENC_MTF, PROF_ENC_RLE, PROF_ENC_CLASS, PROF_ENC_ADAPTIVE, PROF_ENC_PAIRS, PROF_ENC_RANS, PROF_ENC_EMIT,
    PROF_DEC_HEADERS, PROF_DEC_RANS, PROF_DEC_RLE, PROF_DEC_RANK, PROF_CHECKSUM, PROF_LG_HIST, PROF_LG_SCATTER, PROF_SA_PACK, PROF_COUNT
};
struct JpkProfPending { hipEvent_t a, b; int id; uint64_t units; };

struct jpk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // entropy stage: chunk groups, densest first; group g runs the whole stage on aux[g], the last group on `stream`
    static constexpr int ENC_GROUPS = 4;
    hipStream_t aux[ENC_GROUPS - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_pre[ENC_GROUPS] = {nullptr, nullptr, nullptr, nullptr}, ev_done[ENC_GROUPS] = {nullptr, nullptr, nullptr, nullptr};
    // jpk_dev_blocks_decompress of many SMALL blocks: more inverse BWTs side by side than the three of a batch of large ones (an inverse
    // BWT of a 1 MiB block is ~30 dependent launches, each a fraction of the chip).  Streams and events made on first use.
    static constexpr int INV_LANES_MAX = 16;
    hipStream_t inv_lane[INV_LANES_MAX] = {};
    hipEvent_t ev_inv[INV_LANES_MAX] = {};
    // heavy-phase gate (abi.hip): events that mark the end of this context's GPU-saturating work -- [0] the suffix sort, [1..] the
    // wide entropy kernels of each chunk group -- and whether this context currently holds the device's gate
    static constexpr int GATE_EVENTS = 5;
    hipEvent_t ev_gate[GATE_EVENTS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int gate_nev = 0;
    bool gate_held = false;
    hipEvent_t ev_batch = nullptr;                 // jpk_dev_blocks_*: "the caller's stream has reached the batch call"
    hipEvent_t ev_sa[2] = {nullptr, nullptr};      // suffix sort: "the count round r left behind has reached the host"
    bool sa_stats_pending = false;   // per-round statistics of the last suffix sort are still in the pinned mailbox
    uint32_t *h_map = nullptr;       // pinned, 4096 words
    uint8_t *arena = nullptr;
    size_t arena_cap = 0;
    size_t arena_off = 0;
    size_t arena_base = 0;           // where the bump allocator starts (non-zero while a batch runs several stage instances side by side)
    // small pinned host mailbox for device->host scalars
    uint32_t *h_mail = nullptr;   // 256 words pinned
    uint32_t *d_mail = nullptr;   // 256 words device
    // persistent staging buffers for the host-buffer entry points
    uint8_t *stage_in = nullptr, *stage_out = nullptr, *stage_res = nullptr;
    size_t stage_in_cap = 0, stage_out_cap = 0, stage_res_cap = 0;
    jpk_stats stats;
    // profiler
    bool prof_on = false;
    std::vector<JpkProfPending> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[PROF_COUNT] = {0};
    uint64_t prof_launches[PROF_COUNT] = {0};
    uint64_t prof_units[PROF_COUNT] = {0};
};

void jpk_prof_begin(jpk_ctx *ctx, int id, uint64_t units);
void jpk_prof_end(jpk_ctx *ctx);
void jpk_prof_resolve(jpk_ctx *ctx);   // call after a stream synchronisation

// JPK_LAUNCH(ctx, prof id, units processed, kernel, grid, block, args...)
#define JPK_LAUNCH(ctx, id, units, kernel, grid, block, ...)                                   \
    do {                                                                                       \
        if ((ctx)->prof_on) jpk_prof_begin((ctx), (id), (uint64_t)(units));                    \
        hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);               \
        if ((ctx)->prof_on) jpk_prof_end((ctx));                                               \
    } while (0)

// the same with a dynamic LDS reservation (bytes): used to cap the number of resident workgroups of a kernel per CU
#define JPK_LAUNCH_LDS(ctx, id, units, lds, kernel, grid, block, ...)                          \
    do {                                                                                       \
        if ((ctx)->prof_on) jpk_prof_begin((ctx), (id), (uint64_t)(units));                    \
        hipLaunchKernelGGL(kernel, grid, block, (lds), (ctx)->stream, __VA_ARGS__);            \
        if ((ctx)->prof_on) jpk_prof_end((ctx));                                               \
    } while (0)

static inline size_t jpk_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Arena {
    jpk_ctx *c;
    size_t need = 0;      // planning pass accumulates here
    bool planning;
    Arena(jpk_ctx *ctx, bool plan) : c(ctx), planning(plan) { if (!plan) c->arena_off = c->arena_base; }
    template <typename T> T *get(size_t count)
    {
        size_t bytes = jpk_align(count * sizeof(T) + 64);
        if (planning) { need += bytes; return nullptr; }
        T *p = (T *)(c->arena + c->arena_off);
        c->arena_off += bytes;
        return p;
    }
};

int jpk_arena_ensure(jpk_ctx *ctx, size_t bytes);
bool jpk_arena_fits(const jpk_ctx *ctx, size_t bytes);      // jpk_arena_ensure(bytes) would keep the arena where it is
int jpk_stage_ensure(jpk_ctx *ctx, size_t in_bytes, size_t out_bytes);
// copy `words` u32 from device mailbox offset to host (synchronises the stream)
int jpk_read_mail(jpk_ctx *ctx, uint32_t *dst, int words);

static inline unsigned jpk_grid(size_t work, unsigned per_block) { return (unsigned)((work + per_block - 1) / per_block); }

static inline int jpk_bits_for(uint32_t maxval)
{
    int b = 0;
    while (b < 32 && (maxval >> b)) b++;
    return b;
}

// ---- primitives implemented in scan.hip / radix.hip -------------------------------------------------
// scratch requirements are sized by the *_scratch_words helpers; all buffers come from the arena.
size_t jpk_scan_scratch_words(size_t n);
int jpk_exclusive_sum_u32(jpk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch, uint32_t *d_total /*nullable*/);
int jpk_inclusive_max_u32(jpk_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n, uint32_t *scratch);

size_t jpk_radix_scratch_words(size_t n);
// LSD radix sort on bit ranges; result is left in keys/vals (alt buffers used for ping-pong)
int jpk_radix_sort_pairs_u64(jpk_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt, size_t n,
                             const int *shifts, int nshifts, uint32_t *scratch);
int jpk_radix_sort_pairs_u64_nocopy(jpk_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt, size_t n,
                                    const int *shifts, int nshifts, uint32_t *scratch, uint64_t **keys_out, uint32_t **vals_out);
bool jpk_radix_onesweep();   // round 0's radix passes in the one-pass form (default; JPK_ONESWEEP=0: histogram + scan + scatter per pass)
int jpk_radix_sort_slot_keys(jpk_ctx *ctx, uint32_t n, uint64_t *keysA, uint32_t *valsA, uint64_t *keysB, uint32_t *valsB,
                             uint32_t *scratch, uint64_t **keys_out, uint32_t **vals_out, bool group, const uint8_t *slot_tag = nullptr, int tag_shift = 26,
                             uint32_t slot_n = 0);
// Heavy-phase gate (experiment, off by default -- see gate_on() in abi.hip for the numbers): the GPU-saturating phases of the
// blocks in flight on one device -- the suffix sort and the wide kernels of the entropy stage in front of the rANS chains --
// run one block after the other on the GPU, in the order the blocks arrive here, while the chains (a few waves that run for
// milliseconds) of earlier blocks run beside them.  A stream-level dependency, no host blocking on GPU
// work: enter() makes ctx->stream wait for the previous holder's events; jpk_gate_mark() records one of this context's
// events on a stream; leave() publishes them and releases the gate.  The gate is held while the heavy phases are being
// ENQUEUED.
// compress-side calls (forward BWT, rANS encode) of the contexts of ONE DEVICE that are running right now: the encoder cuts its
// chains into fewer launch groups when other blocks are in flight on the same GPU (abi.hip)
int jpk_compress_inflight_enter(int device);        // returns the count on that device including the caller
void jpk_compress_inflight_leave(int device);
int jpk_enc_groups_for(int inflight, uint32_t nch);
struct JpkCompressInflight {
    int n, device;
    explicit JpkCompressInflight(int dev) : n(jpk_compress_inflight_enter(dev)), device(dev) {}
    ~JpkCompressInflight() { jpk_compress_inflight_leave(device); }
    JpkCompressInflight(const JpkCompressInflight &) = delete;
    JpkCompressInflight &operator=(const JpkCompressInflight &) = delete;
};
int jpk_gate_enter(jpk_ctx *ctx);
int jpk_gate_mark(jpk_ctx *ctx, hipStream_t stream);
void jpk_gate_leave(jpk_ctx *ctx);
void jpk_gate_forget(jpk_ctx *ctx);
// moves the per-round statistics of the last suffix sort from the pinned mailbox into ctx->stats (call after a stream sync)
void jpk_sa_stats_sync(jpk_ctx *ctx);

// ---- stage drivers (device buffers) -----------------------------------------------------------------
int jpk_fwd_bwt_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out);
int jpk_suffix_array_device(jpk_ctx *ctx, const uint8_t *d_t, int32_t n, int32_t *d_sa);
// several small blocks as ONE suffix sort (bwt_fwd.hip): images to d_img[b] (len[b] + 480 bytes each); enqueued, no synchronisation
int jpk_fwd_bwt_group_device(jpk_ctx *ctx, int nblk, const uint8_t *const *d_in, const int32_t *len, uint8_t *const *d_img);
size_t jpk_fwd_bwt_group_arena_bytes(uint32_t total_nlen, int nblk);
int jpk_inv_bwt_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len_with_trailer, uint8_t *d_out);
// no host round trip: d_verdict[0..4) (device) receives {status, trailer index, overflow slots, bytes the head chain covers}
int jpk_inv_bwt_enqueue(jpk_ctx *ctx, const uint8_t *d_in, int32_t len_with_trailer, uint8_t *d_out, uint32_t *d_verdict);
int jpk_inv_bwt_chains120_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len_with_trailer, uint8_t *d_out, float *chase_ms);
int jpk_ans_encode_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
int jpk_ans_decode_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
int jpk_ans_decode_batch(jpk_ctx *ctx, int nblk, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out, const int32_t *out_cap,
                         int32_t *out_len, int32_t *status, size_t arena_skip);
int jpk_ans_encode_group_device(jpk_ctx *ctx, int nblk, const uint8_t *d_stage, const uint32_t *first_chunk, const int32_t *mid_len, uint8_t *const *d_out,
                                const int32_t *out_cap, int32_t *out_len, int32_t *status);
size_t jpk_ans_encode_group_arena_bytes(uint32_t nchunks, int nblk);
size_t jpk_inv_bwt_arena_bytes(uint32_t n);
size_t jpk_inv_bwt_batch_arena_bytes(int njobs, const int32_t *len_with_trailer);
constexpr int JPK_INV_BATCH_MAX_JOBS = 65535;               // one set of launches, blockIdx.y = the job
void jpk_inv_bwt_batch_plan_add(int32_t len_with_trailer, size_t *job_bytes, size_t *tiles);
size_t jpk_inv_bwt_batch_plan_total(int njobs, size_t job_bytes, size_t tiles);
int jpk_inv_bwt_batch_enqueue(jpk_ctx *ctx, int njobs, const uint8_t *const *d_in, const int32_t *len_with_trailer, uint8_t *const *d_out, uint32_t *d_verdict,
                              const int *verdict_slot, std::vector<uint8_t> &host_jobs);
size_t jpk_fwd_bwt_arena_bytes(uint32_t n);
size_t jpk_ans_encode_arena_bytes(uint32_t len);          // text-like data (0.55 RLE0 symbols per byte); the arena grows for denser blocks
size_t jpk_ans_encode_arena_bytes_worst(uint32_t len);    // every byte a symbol
int jpk_rank_encode_device(jpk_ctx *ctx, uint8_t *d_t, int32_t *d_freq, int32_t len);
int jpk_rank_decode_device(jpk_ctx *ctx, uint8_t *d_r, const int32_t *d_freq, int32_t len);
int jpk_rle_encode_device(jpk_ctx *ctx, const uint8_t *d_ranks, int32_t len, uint16_t *d_rle, int32_t *rlen);
int jpk_model_pairs_device(jpk_ctx *ctx, const uint16_t *d_rle, int32_t rlen, uint32_t *d_pairs);
int jpk_checksum_device(jpk_ctx *ctx, const uint8_t *d_in, int32_t len, uint32_t *d_result);
