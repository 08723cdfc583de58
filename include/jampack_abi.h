/*
 * jampack_abi.h -- C ABI of libjampack_amd.so: the MI355X (gfx950) implementation of Jampack's block hot path.
 *
 * Every entry point replaces one interface of the reference (loxxous/Jampack, cited file:line).  Plain
 * pointers and sizes only.  Return value: 0 (JPK_OK) or a negative jpk_status; the library never calls
 * exit() (the reference's Error(), format.cpp:6-10, does -- the C++ shim in jampack_amd/csrc/shim maps a
 * non-zero status back to Error() to stay drop-in).
 *
 * Two families:
 *   host-buffer entry points  jpk_*      -- what the reference's call sites would bind (buffers owned by the
 *                                           caller exactly like `Buffer{block,size}`, format.hpp:36-40); data is
 *                                           staged over PCIe on a per-thread context (re-entrant: jampack.cpp:215,
 *                                           313 call these from OpenMP threads, one Jampack instance each).
 *   device-buffer entry points jpk_dev_*  -- same operations on HBM-resident buffers with an explicit context and
 *                                           stream (used by the fused block pipeline, bench.py and the tests).
 *
 * There is no CPU fallback: without a usable gfx950 device every call returns JPK_E_NODEVICE.
 */
#ifndef JAMPACK_ABI_H
#define JAMPACK_ABI_H

#include <stdint.h>

#if defined(__GNUC__)
#define JPK_API __attribute__((visibility("default")))
#else
#define JPK_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define JPK_BWT_UNITS 120                 /* format.hpp:26  BWT_UNITS */
#define JPK_TRAILER_BYTES (JPK_BWT_UNITS * 4)
#define JPK_ANS_CHUNK (1 << 20)           /* ans.hpp:21     StackSize */
#define JPK_MIN_BLOCKSIZE (1 << 20)       /* format.hpp:21  MIN_BLOCKSIZE */
#define JPK_MAX_BLOCKSIZE (1000 << 20)    /* format.hpp:22  MAX_BLOCKSIZE */
/* the forward BWT (jpk_bwt_forward, jpk_block_compress and their jpk_dev_ / batch forms) takes in_len < JPK_FWD_BWT_LIMIT and returns
 * JPK_E_ARG above (the sort keeps the two upper bits of a 32-bit rank for flags): the format's largest block is below it */
#define JPK_FWD_BWT_LIMIT (1u << 30)
#define JPK_SA_MAX_ROUNDS 40               /* rounds reported in jpk_stats (h doubles: 7 * 2^31 > any block) */
#define JPK_JAM_HEADER_BYTES 15           /* jampack.cpp:128-131: "JAM" + crc + payload size + BlockSize */

typedef enum jpk_status {
    JPK_OK = 0,
    JPK_E_ARG = -1,        /* null pointer / negative size */
    JPK_E_CAPACITY = -2,   /* output buffer too small (the reference would overflow) */
    JPK_E_CORRUPT = -3,    /* malformed stream (reference: Error("...") at ans.cpp:92, 298; rle.cpp:72; rank.cpp:107) */
    JPK_E_DEVICE = -4,     /* HIP runtime error */
    JPK_E_ALLOC = -5,      /* device/host allocation failed */
    JPK_E_NODEVICE = -6    /* no gfx950 device visible */
} jpk_status;

typedef struct jpk_ctx jpk_ctx;

/* per-call statistics of the last operation on a context (for bench.py / DESIGN.md accounting) */
typedef struct jpk_stats {
    int32_t sa_rounds;            /* prefix-doubling rounds of the last forward BWT */
    int32_t sa_key_depth;         /* symbols of a suffix that round 0's key holds: the average over the block with the variable-length codes
                                   * (order 0: about 56 / H0 -- 12 for English-like text, 10 over enwik8's byte alphabet; with the order-1 /
                                   * order-2 context codes 13-15), exactly floor(56 / ceil(log2 sigma)) with the fixed-width code (flat
                                   * histograms, blocks above 2^28 bytes, JPK_VARKEYS=0): 7 above 128 byte values */
    int64_t sa_sorted_elems;      /* sum over rounds of active suffixes that went through a sort */
    int64_t inv_splitters;        /* walkers used by the last inverse BWT */
    int64_t inv_overflow_slots;   /* sub-lists that exceeded one scratch slot */
    int64_t workspace_bytes;      /* HBM arena currently held by the context */
    int64_t ans_chunks;           /* 1 MiB chunks in the last entropy call */
    int64_t ans_rle_symbols;      /* RLE0 symbols in the last entropy call */
    /* per round r (r = 0: the radix round on the key; r >= 1: every group of tied suffixes compared at its own depth -- sa_key_depth * 2^(r-1)
     * with the fixed-width code -- or a pair round, see sa_pair_rounds) of the last forward BWT:
     * suffixes still unresolved when the round starts / of those, members of groups too large for the LDS path */
    int32_t sa_round_active[JPK_SA_MAX_ROUNDS];
    int32_t sa_round_large[JPK_SA_MAX_ROUNDS];
    /* the rANS chunk whose four state chains ran longest in the last entropy encode (the serial floor of the stage):
     * shader cycles, nanoseconds (from the 100 MHz real-time counter), encoder steps per chain */
    int64_t enc_chain_cycles;
    int64_t enc_chain_ns;
    int64_t enc_chain_steps;
    /* bit r set: round r of the last forward BWT was a pair round (the induction step over long repeats, bwt_fwd.hip k_pair_*: it
     * resolves whole groups from the order of their successors and leaves the doubling distance alone) instead of a doubling round;
     * r >= 1 then stands for the r-th round after round 0, h = sa_key_depth * 2^(doubling rounds before it) */
    int64_t sa_pair_rounds;
    /* the code of round 0's keys in the last forward BWT: -1 = fixed width, 0 = the variable-length code of the bytes alone, 1 / 2 = every
     * symbol behind the one / the two bytes in front of it (chosen per block from a sample) */
    int32_t sa_key_order;
    int32_t reserved0;
} jpk_stats;

/* ---- contexts ------------------------------------------------------------------------------------------ */
/* stream: a hipStream_t to launch on (NULL = the context creates its own non-blocking stream).
 * Replaces the reference's per-call cudaMalloc/cudaMemcpy/cudaFree staging (bwt.cpp:189-239). */
JPK_API int jpk_ctx_create(jpk_ctx **out, int device, void *hip_stream);
JPK_API void jpk_ctx_destroy(jpk_ctx *ctx);
JPK_API int jpk_ctx_stats(jpk_ctx *ctx, jpk_stats *out);
/* pre-size the HBM arena for blocks up to max_block_bytes (otherwise it grows on demand): the maximum of the four stages' own
 * layouts (jpk_debug_arena_bytes).  JPK_E_ARG for max_block_bytes < 0 or > JPK_MAX_BLOCKSIZE, JPK_E_ALLOC when HBM is short. */
JPK_API int jpk_ctx_reserve(jpk_ctx *ctx, int64_t max_block_bytes);
/* per-kernel timing with HIP events recorded on the context's stream: enable = 1 on, 2 on + reset, 0 off + reset.
 * id < jpk_ctx_profile_count(); units = elements the timed launches processed (see DESIGN.md for bytes per unit). */
JPK_API int jpk_ctx_profile(jpk_ctx *ctx, int enable);
JPK_API int jpk_ctx_profile_count(void);
JPK_API const char *jpk_ctx_profile_name(int id);
JPK_API int jpk_ctx_profile_get(jpk_ctx *ctx, int id, double *ms, int64_t *launches, int64_t *units);
JPK_API int jpk_device_count(void);
JPK_API const char *jpk_strerror(int status);
JPK_API const char *jpk_version(void);

/* ---- process-wide set-up of the host-buffer entry points -------------------------------------- */
/* Selects the devices the host-buffer entry points may use: bit d of device_mask = HIP device d, 0 = every visible
 * gfx950 device.  Calling threads are dealt round robin over the selected devices and keep a persistent context
 * (stream + HBM arena + staging) from a process-wide pool, so the OpenMP block loop of jampack.cpp:215/313 spreads
 * its blocks over the node.  Returns the number of devices selected (> 0) or a negative jpk_status.  Optional: without
 * it the first host-buffer call selects all devices (or the one named by the JPK_DEVICE environment variable).
 * Replaces the reference's per-block cudaMalloc/cudaFree and its single-device choice (bwt.cpp:98-114, 189-239). */
JPK_API int jpk_init(uint64_t device_mask);
/* the devices selected by jpk_init (or by the implicit selection), in round-robin order; returns their count */
JPK_API int jpk_init_devices(int32_t *devices, int32_t cap);
/* device the calling thread has been dealt (creates its context if needed), or a negative jpk_status */
JPK_API int jpk_thread_device(void);
/* destroys every pooled context (arenas, streams, staging).  No host-buffer call may be in flight.  Threads that call
 * again afterwards get fresh contexts. */
JPK_API void jpk_shutdown(void);
/* gives back what the batch entries keep between calls and nobody is using right now: the idle worker contexts of
 * jpk_dev_blocks_compress / jpk_dev_blocks_decompress (one arena each: ~54 bytes per block byte of the largest block they have seen) and
 * the multi-device entries' slabs.  May run beside other calls; they create what they need again.  Returns the contexts destroyed.
 * (The reference frees its device buffers after every block: bwt.cpp:98-114.) */
JPK_API int jpk_release_idle(void);

/* ---- host-buffer entry points (drop-in boundary) ------------------------------------------------------- */
/* BlockSort::Bwt::ForwardBwt(Buffer,Buffer)            bwt.hpp:15, bwt.cpp:22-65.   *out_len = in_len + 480.  in_len < JPK_FWD_BWT_LIMIT. */
JPK_API int jpk_bwt_forward(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len);
/* BlockSort::Bwt::InverseBwt(Buffer,Buffer,Options)    bwt.hpp:16, bwt.cpp:72-282.  threads/use_gpu mirror
 * Options.Threads / Options.Gpu (format.hpp:46-54); they do not change the bytes and are accepted for ABI fidelity. */
JPK_API int jpk_bwt_inverse(const uint8_t *in, int32_t in_len_with_trailer, uint8_t *out, int32_t out_cap, int32_t *out_len,
                    int32_t threads, int32_t use_gpu);
/* Ans::Encode(Buffer,Buffer,Options)                   ans.hpp:32, ans.cpp:113-234. The reference clobbers its
 * input (rank.cpp:88); this implementation leaves it intact but the contract still allows clobbering. */
JPK_API int jpk_ans_encode(uint8_t *in_clobbered, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len);
/* Ans::Decode(Buffer,Buffer,Options)                   ans.hpp:33, ans.cpp:236-270 */
JPK_API int jpk_ans_decode(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, int32_t threads);
/* Header-only walk of an Ans stream (ReadHeader per chunk, ans.cpp:254-261, 287-302) on the host: the decoded length
 * (sum of the chunks' original lengths) and the chunk count, with the reference's header sanity checks.  The reference's
 * Ans::Decode has no capacity argument -- it trusts that the caller's buffer holds the frame (jampack.cpp:156-159) -- so the
 * C++ shim uses this to bound jpk_ans_decode by what the stream itself declares instead of by Options.BlockSize, which on
 * the decompress path is the CLI default, not the frame's block size (main.cpp:60, jampack.cpp:146-159). */
JPK_API int jpk_ans_decoded_size(const uint8_t *in, int32_t in_len, int64_t *decoded_len, int32_t *chunks);
/* Postcoder::Encode / Decode                           rank.hpp:12-13, rank.cpp:45-151 (in place) */
JPK_API int jpk_rank_encode(uint8_t *t, int32_t *freq256, int32_t len);
JPK_API int jpk_rank_decode(uint8_t *ranks, const int32_t *freq256, int32_t len);
/* fused Jampack::Comp()/Decomp() tail: ForwardBwt -> Ans::Encode / Ans::Decode -> InverseBwt with the BWT
 * image kept in HBM between the two stages (jampack.cpp:40-41, 49-50). */
JPK_API int jpk_block_compress(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_block_decompress(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len);
/* Checksum::IntegrityCheck(Buffer)                     checksum.hpp:15, checksum.cpp:12-36 */
JPK_API int jpk_checksum(const uint8_t *in, int32_t in_len, uint32_t *crc);
/* One framed block of a .jam stream: Jampack::Comp() (crc of the input, jampack.cpp:31) + CompWriteBlock
 * (jampack.cpp:122-135):  "JAM" | u32 crc | i32 payload size | i32 BlockSize | payload   (15-byte header, LE).
 * The payload is jpk_block_compress(in): the reference CLI additionally runs its LZ77 / filter / LPX pre-stages in
 * front of the BWT (jampack.cpp:33-38), so frames interchange with a reference build whose Comp()/Decomp() call
 * this path, not with the stock CLI.  block_size is Options.BlockSize and must lie in
 * [JPK_MIN_BLOCKSIZE, JPK_MAX_BLOCKSIZE] with in_len <= block_size. */
JPK_API int jpk_jam_block_write(const uint8_t *in, int32_t in_len, int32_t block_size, uint8_t *out, int32_t out_cap, int32_t *out_len);
/* DecompReadBlock + Decomp() (jampack.cpp:140-164, 47-60): validates the header exactly as the reference does
 * (magic, BlockSize range, 0 <= payload size <= MAX_BLOCKSIZE), decodes the payload and checks the crc.
 * *consumed = 15 + payload size (where the next frame starts).  JPK_E_CORRUPT on a bad header, a payload that
 * runs past in_len, or a crc mismatch ("Detected corrupt block!", jampack.cpp:59). */
JPK_API int jpk_jam_block_read(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, int32_t *consumed);

/* ---- pre-stage decoders + frames of the stock CLI (host code, no GPU involved) ------------ */
/* Lz77::Decompress(Buffer,Buffer)                      lz77.hpp:22, lz77.cpp:678-714 */
JPK_API int jpk_lz77_decompress(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len);
/* Lpx::Decode(Buffer,Buffer,Options)                   lpx.hpp:32, lpx.cpp:101-169 (output length = input length) */
JPK_API int jpk_lpx_decode(const uint8_t *in, int32_t len, uint8_t *out);
/* Filters::Decode(Buffer,Buffer)                       filters.hpp:44, filters.cpp:442-490 */
JPK_API int jpk_filters_decode(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len);
/* Checksum::IntegrityCheck on the host                 checksum.cpp:12-36 */
JPK_API uint32_t jpk_checksum_host(const uint8_t *p, int32_t size);
/* One frame written by an unmodified `jampack c` (any -m / -f setting): DecompReadBlock + the whole Jampack::Decomp()
 * (jampack.cpp:47-60, 140-164) -- Ans::Decode and InverseBwt on the GPU, Lz77::Decompress, Lpx::Decode,
 * Filters::Decode, Lz77::Decompress on the host, then the crc check.  *consumed = 15 + payload size. */
JPK_API int jpk_jam_cli_block_read(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, int32_t *consumed);

/* ---- device-buffer entry points (all pointers except ctx/out_len are HBM addresses on ctx's device) ---- */
JPK_API int jpk_dev_bwt_forward(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_bwt_inverse(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len_with_trailer, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_ans_encode(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_ans_decode(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_rank_encode(jpk_ctx *ctx, uint8_t *d_t, int32_t *d_freq256, int32_t len);
JPK_API int jpk_dev_rank_decode(jpk_ctx *ctx, uint8_t *d_ranks, const int32_t *d_freq256, int32_t len);
JPK_API int jpk_dev_block_compress(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_block_decompress(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_checksum(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint32_t *crc);
JPK_API int jpk_dev_jam_block_write(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, int32_t block_size, uint8_t *d_out, int32_t out_cap, int32_t *out_len);
JPK_API int jpk_dev_jam_block_read(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len, int32_t *consumed);

/* ---- batches of independent blocks (device buffers) ------------------------------------------------------------ */
/* Jampack::Decompress's multi-block mode (jampack.cpp:286-317: Threads blocks read, Decomp() in an OpenMP loop, written in
 * order) for blocks that already sit in HBM: Ans::Decode (+ InverseBwt) of nblocks independent blocks in ONE pass -- every
 * serial entropy kernel runs a single grid over the 1 MiB chunks of all blocks, which is what fills the GPU (one block is 65
 * chains on 1024 SIMDs).  Arrays of nblocks device pointers / sizes (the arrays themselves are host memory).  status may be
 * NULL; otherwise status[b] receives block b's jpk_status and a corrupt block does not stop the others.
 * jpk_dev_blocks_decompress: the inverse BWTs follow on up to three streams of the context; a batch of many small blocks (>= 32 blocks
 * of <= 4 MiB, a stream of the reference's smallest block size, format.hpp:22) runs them through one set of launches over all blocks
 * and needs scratch for all of them at once (about 10 bytes per block byte of the context's arena). */
JPK_API int jpk_dev_blocks_ans_decode(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                              const int32_t *out_cap, int32_t *out_len, int32_t *status);
JPK_API int jpk_dev_blocks_decompress(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                              const int32_t *out_cap, int32_t *out_len, int32_t *status);

/* The compress direction of the same loop (jampack.cpp:205-224: Threads blocks read, Comp() in an OpenMP loop, written in order)
 * for blocks that sit in HBM: ForwardBwt + Ans::Encode of nblocks independent blocks in ONE call.  Compression wants several
 * blocks IN FLIGHT rather than one wide grid (the suffix sort fills the GPU by itself; the entropy stage of the other blocks
 * hides in its latency), so the library runs `in_flight` worker threads (<= 0: 10; at most 16), each with a context of its own on
 * ctx's device (kept by the library between calls, released by jpk_shutdown), that take the work in array order.  Blocks of up to
 * 16 MiB -- the reference's default block is 8 MiB, its smallest 1 MiB (format.hpp:20-22) -- are compressed in GROUPS of
 * consecutive blocks (a quarter of their total bytes, 8 .. 64 MiB, at most 256 blocks; JPK_GROUP_MIB fixes the size, JPK_GROUP=0
 * turns grouping off): one suffix sort over the blocks of a group, one set of entropy grids over all their chunks, and a handful of
 * host synchronisations per group instead of one per block -- the suffix sort's per-round counts from its third round on, the symbol
 * layout, the chunk sizes and the end of the emit kernels (one more when a block does not fit its buffer); every block's bytes are those
 * of jpk_dev_block_compress for that block.  A group that fails as a whole (its arena, a device error) is retried block by block through
 * the single-block path on the same context; a block whose out_cap is too small reports JPK_E_CAPACITY alone.
 * The calling thread works too (with ctx) and returns when every block is done.  status may be NULL; otherwise
 * status[b] receives block b's jpk_status (JPK_E_CAPACITY when out_cap[b] is too small, ...) and the other blocks still complete.
 * Stream order: as for every jpk_dev_* call, work already queued on ctx's stream (the producers of d_in[], readers of an earlier
 * d_out[]) is ordered in front of the batch -- the workers' streams wait for an event recorded on ctx's stream at entry -- and
 * every block is complete when the call returns. */
JPK_API int jpk_dev_blocks_compress(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                            const int32_t *out_cap, int32_t *out_len, int32_t *status, int32_t in_flight);

/* ---- kernel-level probes used by tests/ and bench.py (device buffers) ------------------------------------ */
/* Comparator for BASELINE config 3 ("120-way parallel LF-map"): the reference's own GPU kernel shape -- 120 threads, one per
 * stored index, p = Map[p-1] (CUDAInverse<<<40,3>>>, bwt.cpp:8-19, 176-183, 226-229) -- on the same Map.  Same bytes as
 * jpk_dev_bwt_inverse; *chase_ms = device time of the chase kernel alone.  A measured baseline, never a product path. */
JPK_API int jpk_dev_bwt_inverse_chains120(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len_with_trailer, uint8_t *d_out, int32_t out_cap,
                                  int32_t *out_len, float *chase_ms);
/* suffix array of d_t[0..n) into d_sa (int32[n]) -- the divsufsort() replacement, divsufsort.cpp:1721 */
JPK_API int jpk_dev_suffix_array(jpk_ctx *ctx, const uint8_t *d_t, int32_t n, int32_t *d_sa);
/* stable LSD radix sort of (u64 key, u32 value) pairs on bits [bit_lo, bit_hi) */
JPK_API int jpk_dev_sort_pairs_u64(jpk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, int32_t n, int32_t bit_lo, int32_t bit_hi);
/* exclusive prefix sum of uint32[n] in place; returns the total in *total */
JPK_API int jpk_dev_exclusive_scan_u32(jpk_ctx *ctx, uint32_t *d_data, int32_t n, uint32_t *total);
/* entropy sub-stages of one chunk: rank array -> RLE0 symbols; symbols -> packed (low | freq<<16) pairs */
JPK_API int jpk_dev_rle_encode(jpk_ctx *ctx, const uint8_t *d_ranks, int32_t len, uint16_t *d_rle, int32_t *rlen);
JPK_API int jpk_dev_model_pairs(jpk_ctx *ctx, const uint16_t *d_rle, int32_t rlen, uint32_t *d_pairs);

/* ---- host-logic probes (no device call; tests/test_abi_and_host.py) ----------------------------------------------- */
/* The encoder cuts one block's chunks into 1..4 graded launch groups by the number of blocks that are in their forward BWT or
 * entropy encode ON THE SAME DEVICE at that moment (jampack.cpp:215 runs one block per OpenMP thread; with jpk_init over eight
 * GPUs every block is alone on its device).  jpk_debug_compress_inflight: delta > 0 registers a block on `device` and returns the
 * count including it, delta < 0 removes one and returns what is left, delta == 0 reads.  jpk_debug_enc_groups: the launch groups
 * a block of `nch` chunks arriving on `device` now would get. */
JPK_API int jpk_debug_compress_inflight(int device, int delta);
/* HBM arena bytes stage 0 (forward BWT) / 1 (rANS encode, text-like data: 0.55 RLE0 symbols per byte) / 2 (inverse BWT) /
 * 3 (rANS decode, bound) plans for one block of block_bytes; jpk_ctx_reserve takes their maximum.  Stage 4: rANS encode of the
 * densest data (every byte a symbol) -- what a context's arena grows to the first time such a block arrives. */
JPK_API int64_t jpk_debug_arena_bytes(int64_t block_bytes, int stage);
/* jpk_ans_decode calls that arrive from different threads at about the same time are merged into one batched pass on their
 * device (ans.cpp:254-264 decodes Threads chunks at a time; jampack.cpp:313 calls Decomp() from Threads OpenMP threads): the
 * number of requests the most recent pass on `device` carried (1: a lone caller, single-block path).  JPK_COMBINE_US=<grace
 * in microseconds, default 300; negative: never merge> in the environment. */
JPK_API int jpk_debug_combiner_last_batch(int device);
JPK_API int jpk_debug_enc_groups(int device, int32_t nch);
/* The block loops of Jampack::Compress / Jampack::Decompress (jampack.cpp:205-224, 286-317) over the GPUs of one node, natively:
 * block b runs on the (b mod G)-th device of `device_mask` (bit d = device d, 0 = every visible gfx950 device; one worker thread per
 * device; `in[b]` are HOST buffers, copied into a per-device slab the library keeps between calls), THROUGH THE LIBRARY'S BATCH ENTRY
 * on that device -- jpk_dev_blocks_compress with `in_flight` blocks in flight (<= 0: its default, 10) and small blocks in groups, or
 * jpk_dev_blocks_decompress (one pass over the chunks of all the device's blocks) -- and the results are gathered in block order into
 * `d_out`, a buffer of out_cap bytes on the FIRST device of the mask: block b occupies [out_off[b], out_off[b + 1]) (out_off has
 * nblocks + 1 entries).  Decompress: in_len[b] = compressed bytes, raw_len[b] = the block's decompressed size (the frame header's
 * BlockSize or what jpk_ans_decoded_size reports minus the trailer); the gather moves raw_len[b] bytes per block.
 * The gather is one ncclSend / ncclRecv pair of exactly the block's bytes per block of a non-root device, grouped, over a
 * single-process RCCL communicator (ncclCommInitAll) that the library loads at first use and keeps until jpk_shutdown; the root's
 * own blocks are device-to-device copies (JPK_MULTI_FORCE_RCCL=1: through RCCL as well).  Compress: a device's inputs travel on a
 * stream of their own, block by block, while the blocks before them are being compressed (jampack.cpp:205-224 overlaps its reads the
 * same way).
 * STATUS AND RESULTS: status[b] (nullable) receives every block's own status; a block that was never reached (its device failed
 * early) reports an error, never JPK_OK.  Every block whose status is JPK_OK IS GATHERED, also when other blocks failed (a corrupt
 * frame among healthy ones): its bytes are at [out_off[b], out_off[b + 1]), a failed block's range is empty, and the call returns the
 * first failed block's status.  When the gather itself cannot run (JPK_E_CAPACITY: d_out too small for the blocks that are done;
 * RCCL missing) nothing is in d_out, EVERY status is an error and every range is empty.
 * ONE CALL AT A TIME PER DEVICE: a call holds the mutexes of the devices of its mask; calls on disjoint device sets run side by side,
 * calls that share a device (and jpk_shutdown) queue.  The caller's current HIP device is restored on return.
 * jpk_debug_multi_plan: the ownership rule alone, for `ndev_visible` devices (no device call).  jpk_debug_multi_lock_probe: takes the
 * device mutexes of `device_mask` for hold_ms milliseconds (no device call; returns the number of mutexes). */
JPK_API int jpk_blocks_compress_multi(uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, uint8_t *d_out, int64_t out_cap,
                                      int64_t *out_off, int32_t *status);
JPK_API int jpk_blocks_compress_multi_ex(uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, uint8_t *d_out, int64_t out_cap,
                                         int64_t *out_off, int32_t *status, int32_t in_flight);
JPK_API int jpk_blocks_decompress_multi(uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, const int32_t *raw_len, uint8_t *d_out,
                                        int64_t out_cap, int64_t *out_off, int32_t *status);
JPK_API int jpk_debug_multi_plan(uint64_t device_mask, int32_t ndev_visible, int32_t nblocks, int32_t *owner);
JPK_API int jpk_debug_multi_lock_probe(uint64_t device_mask, int32_t hold_ms);
/* host-logic probe: the work list jpk_dev_blocks_compress forms for these block lengths (groups of small blocks, large blocks alone):
 * task t covers blocks [first[t], first[t] + count[t]); returns the number of tasks.  No device call. */
JPK_API int jpk_debug_group_plan(int32_t nblocks, const int32_t *in_len, int32_t *first, int32_t *count);
/* host-logic probe: which rounds of a suffix sort the host makes PAIR rounds (bwt_fwd.hip PairSchedule: behind a doubling round that left >= 90 % of
 * its list, two doubling rounds apart, twice as far behind a pair round that left most of its list; JPK_PAIR_* override), given what the host
 * knows -- list[r] = the unresolved suffixes round r started with (list[0] = n; 0 = the sort had ended).  is_pair[r] = 1 for pair rounds; returns
 * their number.  No device call. */
JPK_API int jpk_debug_pair_schedule(int64_t n, int32_t nrounds, const uint32_t *list, int32_t runs_heavy, int32_t *is_pair);
/* Hooks that CHANGE live state work only in a process with JPK_DEBUG_HOOKS=1 in its environment (JPK_E_ARG otherwise):
 * jpk_debug_compress_inflight with delta != 0, and jpk_debug_combiner_fail_next(n): the next n merged decode passes fail as a
 * whole before they run -- every merged request must then come back through its own thread's single-block path. */
JPK_API int jpk_debug_combiner_fail_next(int n);
/* (JPK_DEBUG_HOOKS=1 only) the next n groups of jpk_dev_blocks_compress fail as a whole before they run: their blocks must come back
 * through the single-block path with the same bytes */
JPK_API int jpk_debug_group_fail_next(int n);

#ifdef __cplusplus
}
#endif
#endif /* JAMPACK_ABI_H */
