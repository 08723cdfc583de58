// abi.hip -- C ABI of libjampack_amd.so (include/jampack_abi.h): contexts, HBM arena, PCIe staging for the
// host-buffer (drop-in) entry points, and the fused block pipeline.  No CPU fallback: without a gfx950 device
// every entry point returns JPK_E_NODEVICE.
#include <chrono>
#include <atomic>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

// Independent blocks run on separate contexts (= HIP streams).  ROCm multiplexes streams onto GPU_MAX_HW_QUEUES
// hardware queues, round robin in creation order (default 4: tools/conctest.hip shows exactly four kernels at a time whatever
// the number of streams; with 32 queues sixteen single-wave kernels ran fully concurrently).  Two streams that share a queue
// serialise -- with four streams per context (main + three encoder group streams) and 16 queues the main streams of contexts
// 0 and 4 collided, which capped round 1's blocks in flight at eight.  Ask for 32 queues unless the user has chosen a
// value; this runs when the library is loaded, i.e. before the first HIP call of a program that links it.
__attribute__((constructor)) static void jpk_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "32", 0); }

// ---- arena / staging ---------------------------------------------------------------------------------------
bool jpk_arena_fits(const jpk_ctx *ctx, size_t bytes) { return jpk_align(bytes + 4096, 1 << 20) <= ctx->arena_cap; }

int jpk_arena_ensure(jpk_ctx *ctx, size_t bytes)
{
    bytes = jpk_align(bytes + 4096, 1 << 20);
    if (bytes <= ctx->arena_cap) return JPK_OK;
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    for (int g = 0; g + 1 < jpk_ctx::ENC_GROUPS; g++)
        if (ctx->aux[g]) JPK_HIP(hipStreamSynchronize(ctx->aux[g]));
    for (int k = 0; k < jpk_ctx::INV_LANES_MAX; k++)
        if (ctx->inv_lane[k]) JPK_HIP(hipStreamSynchronize(ctx->inv_lane[k]));
    const bool growing = ctx->arena != nullptr;
    if (ctx->arena) { JPK_HIP(hipFree(ctx->arena)); ctx->arena = nullptr; ctx->arena_cap = 0; }
    // an arena that has to GROW grows geometrically, so that a sequence of slightly larger blocks does not re-allocate every time;
    // the first allocation (jpk_ctx_reserve, or a context's first block) takes what was asked for
    size_t want = growing ? bytes + bytes / 8 : bytes;
    hipError_t e = hipMalloc((void **)&ctx->arena, want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc((void **)&ctx->arena, want);
        if (e != hipSuccess) { ctx->arena = nullptr; return JPK_E_ALLOC; }
    }
    ctx->arena_cap = want;
    ctx->stats.workspace_bytes = (int64_t)want;
    return JPK_OK;
}

static int buf_ensure(jpk_ctx *ctx, uint8_t **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return JPK_OK;
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    if (*p) { JPK_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    bytes = jpk_align(bytes + bytes / 16 + 4096, 1 << 16);
    if (hipMalloc((void **)p, bytes) != hipSuccess) { *p = nullptr; return JPK_E_ALLOC; }
    *cap = bytes;
    return JPK_OK;
}

int jpk_stage_ensure(jpk_ctx *ctx, size_t in_bytes, size_t out_bytes)
{
    JPK_TRY(buf_ensure(ctx, &ctx->stage_in, &ctx->stage_in_cap, in_bytes));
    JPK_TRY(buf_ensure(ctx, &ctx->stage_out, &ctx->stage_out_cap, out_bytes));
    return JPK_OK;
}

int jpk_read_mail(jpk_ctx *ctx, uint32_t *dst, int words)
{
    JPK_HIP(hipMemcpyAsync(ctx->h_mail, ctx->d_mail, (size_t)words * 4, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->h_mail, (size_t)words * 4);
    if (ctx->prof_on) jpk_prof_resolve(ctx);
    return JPK_OK;
}

// ---- heavy-phase gate (common.hpp) -----------------------------------------------------------------------------------------
namespace {
struct Gate {
    std::mutex mu[64];
    jpk_ctx *owner[64] = {nullptr};        // per device: the context whose heavy phases were admitted last
};
Gate &gate() { static Gate g; return g; }
// OFF unless JPK_GATE=1: measured on MI355X the gate LOSES -- default 2-block step 1.89 GB/s with it against 1.99 without,
// enwik9-like with 8 blocks in flight 2.28 against 2.68 GB/s -- the kernels of interleaved blocks fill each other's latency
// gaps better than a strict block-after-block order does.  Kept as a switch for the record and for other mixes of blocks.
bool gate_on() { static const bool on = getenv("JPK_GATE") && atoi(getenv("JPK_GATE")) == 1; return on; }
}  // namespace

int jpk_gate_enter(jpk_ctx *ctx)
{
    if (!gate_on() || ctx->gate_held || ctx->device < 0 || ctx->device >= 64) return JPK_OK;
    Gate &g = gate();
    g.mu[ctx->device].lock();
    jpk_ctx *prev = g.owner[ctx->device];
    if (prev && prev != ctx)
        for (int k = 0; k < prev->gate_nev; k++)
            if (hipStreamWaitEvent(ctx->stream, prev->ev_gate[k], 0) != hipSuccess) { g.mu[ctx->device].unlock(); return JPK_E_DEVICE; }
    ctx->gate_held = true;
    ctx->gate_nev = 0;
    return JPK_OK;
}

int jpk_gate_mark(jpk_ctx *ctx, hipStream_t stream)
{
    if (!ctx->gate_held || ctx->gate_nev >= jpk_ctx::GATE_EVENTS) return JPK_OK;
    JPK_HIP(hipEventRecord(ctx->ev_gate[ctx->gate_nev], stream));
    ctx->gate_nev++;
    return JPK_OK;
}

void jpk_gate_leave(jpk_ctx *ctx)
{
    if (!ctx->gate_held) return;
    Gate &g = gate();
    g.owner[ctx->device] = ctx;
    ctx->gate_held = false;
    g.mu[ctx->device].unlock();
}

void jpk_gate_forget(jpk_ctx *ctx)
{
    if (ctx->device < 0 || ctx->device >= 64) return;
    Gate &g = gate();
    if (ctx->gate_held) { ctx->gate_held = false; g.owner[ctx->device] = nullptr; g.mu[ctx->device].unlock(); return; }
    std::lock_guard<std::mutex> lk(g.mu[ctx->device]);
    if (g.owner[ctx->device] == ctx) g.owner[ctx->device] = nullptr;
}

// ---- per-kernel HIP-event profiler (events are recorded on the launch stream) ------------------------------
static hipEvent_t prof_event(jpk_ctx *ctx)
{
    hipEvent_t e = nullptr;
    if (!ctx->prof_pool.empty()) { e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
}
void jpk_prof_begin(jpk_ctx *ctx, int id, uint64_t units)
{
    JpkProfPending p;
    p.a = prof_event(ctx); p.b = prof_event(ctx); p.id = id; p.units = units;
    (void)hipEventRecord(p.a, ctx->stream);
    ctx->prof_pending.push_back(p);
}
void jpk_prof_end(jpk_ctx *ctx) { (void)hipEventRecord(ctx->prof_pending.back().b, ctx->stream); }
void jpk_prof_resolve(jpk_ctx *ctx)
{
    for (auto &p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof_ms[p.id] += ms;
            ctx->prof_launches[p.id] += 1;
            ctx->prof_units[p.id] += p.units;
        }
        ctx->prof_pool.push_back(p.a);
        ctx->prof_pool.push_back(p.b);
    }
    ctx->prof_pending.clear();
}
static const char *const PROF_NAMES[PROF_COUNT] = {
    "k_rs_hist/k_os_digits", "k_rs_scatter/k_os_scatter", "k_scan_*/k_tab_*/k_win_*", "k_gather_win", "k_seg_round", "k_r0_*/k_lg_finish/k_cmp_*", "k_bwt_image",
    "k_hist", "k_build_nxt", "k_walk", "k_rank_jump", "k_copy_out",
    "k_enc_hist/k_enc_prep", "k_enc_mtf", "k_rle_*", "k_cls_*/k_quasi_build", "k_adaptive", "k_pairs", "k_rans_lanes", "k_emit_*/k_put_*",
    "k_dec_headers", "k_dec_rans", "k_dec_rle", "k_dec_rank", "k_chk_*", "k_lg_hist", "k_lg_scatter", "k_sym_present/k_pack_keys"};

extern "C" int jpk_ctx_profile(jpk_ctx *ctx, int enable)
{
    if (!ctx) return JPK_E_ARG;
    JPK_HIP(hipSetDevice(ctx->device));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    jpk_prof_resolve(ctx);
    ctx->prof_on = enable != 0;
    if (enable == 2 || enable == 0) {   // 2 = enable and reset counters
        for (int i = 0; i < PROF_COUNT; i++) { ctx->prof_ms[i] = 0; ctx->prof_launches[i] = 0; ctx->prof_units[i] = 0; }
    }
    return JPK_OK;
}
extern "C" int jpk_ctx_profile_count(void) { return PROF_COUNT; }
extern "C" const char *jpk_ctx_profile_name(int id) { return (id >= 0 && id < PROF_COUNT) ? PROF_NAMES[id] : ""; }
extern "C" int jpk_ctx_profile_get(jpk_ctx *ctx, int id, double *ms, int64_t *launches, int64_t *units)
{
    if (!ctx || id < 0 || id >= PROF_COUNT) return JPK_E_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    jpk_prof_resolve(ctx);
    if (ms) *ms = ctx->prof_ms[id];
    if (launches) *launches = (int64_t)ctx->prof_launches[id];
    if (units) *units = (int64_t)ctx->prof_units[id];
    return JPK_OK;
}

// ---- contexts ----------------------------------------------------------------------------------------------
extern "C" int jpk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int jpk_ctx_create(jpk_ctx **out, int device, void *hip_stream)
{
    if (!out) return JPK_E_ARG;
    *out = nullptr;
    int n = jpk_device_count();
    if (n <= 0 || device < 0 || device >= n) return JPK_E_NODEVICE;
    JPK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    JPK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("JPK_ALLOW_ANY_ARCH")) return JPK_E_NODEVICE;
    jpk_ctx *c = new (std::nothrow) jpk_ctx();
    if (!c) return JPK_E_ALLOC;
    memset(&c->stats, 0, sizeof c->stats);
    c->device = device;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return JPK_E_DEVICE; }
        c->own_stream = true;
    }
    if (hipHostMalloc((void **)&c->h_mail, 256 * 4, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c->h_map, 4096 * 4, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&c->d_mail, 256 * 4) != hipSuccess) {
        jpk_ctx_destroy(c);
        return JPK_E_ALLOC;
    }
    for (int k = 0; k < 2; k++)
        if (hipEventCreateWithFlags(&c->ev_sa[k], hipEventDisableTiming) != hipSuccess) { jpk_ctx_destroy(c); return JPK_E_ALLOC; }
    if (hipEventCreateWithFlags(&c->ev_batch, hipEventDisableTiming) != hipSuccess) { jpk_ctx_destroy(c); return JPK_E_ALLOC; }
    for (int k = 0; k < jpk_ctx::GATE_EVENTS; k++)
        if (hipEventCreateWithFlags(&c->ev_gate[k], hipEventDisableTiming) != hipSuccess) { jpk_ctx_destroy(c); return JPK_E_ALLOC; }
    for (int g = 0; g < jpk_ctx::ENC_GROUPS; g++) {
        // (the encoder's group streams c->aux[] are created on first use: a context that only decodes owns one stream, and
        // streams are dealt round robin onto the hardware queues when they are created)
        if (hipEventCreateWithFlags(&c->ev_pre[g], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_done[g], hipEventDisableTiming) != hipSuccess) {
            jpk_ctx_destroy(c);
            return JPK_E_ALLOC;
        }
    }
    *out = c;
    return JPK_OK;
}

extern "C" void jpk_ctx_destroy(jpk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    jpk_gate_forget(c);
    for (int k = 0; k < jpk_ctx::GATE_EVENTS; k++)
        if (c->ev_gate[k]) (void)hipEventDestroy(c->ev_gate[k]);
    if (c->arena) (void)hipFree(c->arena);
    if (c->stage_in) (void)hipFree(c->stage_in);
    if (c->stage_out) (void)hipFree(c->stage_out);
    if (c->stage_res) (void)hipFree(c->stage_res);
    if (c->d_mail) (void)hipFree(c->d_mail);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    if (c->h_map) (void)hipHostFree(c->h_map);
    for (int g = 0; g < jpk_ctx::ENC_GROUPS; g++) {
        if (g + 1 < jpk_ctx::ENC_GROUPS && c->aux[g]) { (void)hipStreamSynchronize(c->aux[g]); (void)hipStreamDestroy(c->aux[g]); }
        if (c->ev_pre[g]) (void)hipEventDestroy(c->ev_pre[g]);
        if (c->ev_done[g]) (void)hipEventDestroy(c->ev_done[g]);
    }
    for (int k = 0; k < 2; k++)
        if (c->ev_sa[k]) (void)hipEventDestroy(c->ev_sa[k]);
    for (int k = 0; k < jpk_ctx::INV_LANES_MAX; k++) {
        if (c->inv_lane[k]) { (void)hipStreamSynchronize(c->inv_lane[k]); (void)hipStreamDestroy(c->inv_lane[k]); }
        if (c->ev_inv[k]) (void)hipEventDestroy(c->ev_inv[k]);
    }
    if (c->ev_batch) (void)hipEventDestroy(c->ev_batch);
    for (auto &p : c->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int jpk_ctx_stats(jpk_ctx *ctx, jpk_stats *out)
{
    if (!ctx || !out) return JPK_E_ARG;
    if (ctx->sa_stats_pending) {
        JPK_HIP(hipSetDevice(ctx->device));
        JPK_HIP(hipStreamSynchronize(ctx->stream));
        jpk_sa_stats_sync(ctx);
    }
    *out = ctx->stats;
    return JPK_OK;
}

extern "C" int jpk_ctx_reserve(jpk_ctx *ctx, int64_t max_block_bytes)
{
    if (!ctx || max_block_bytes < 0) return JPK_E_ARG;
    JPK_HIP(hipSetDevice(ctx->device));
    if (max_block_bytes > (int64_t)JPK_MAX_BLOCKSIZE) return JPK_E_ARG;
    // the largest of the four stages' own layouts (their planning passes), not a guess (jpk_debug_arena_bytes; DESIGN.md section 3,
    // tests/test_abi_and_host.py pins the factors): forward BWT ~46 n, rANS encode -- see jpk_ans_encode_arena_bytes --, inverse BWT
    // ~10 n, rANS decode 3 n + chunk tables.  Blocks above JPK_MAX_BLOCKSIZE are refused with JPK_E_ARG.
    const uint32_t n = (uint32_t)max_block_bytes, mid = n + JPK_TRAILER_BYTES;
    size_t need = jpk_fwd_bwt_arena_bytes(n);
    const size_t enc = jpk_ans_encode_arena_bytes(mid), inv = jpk_inv_bwt_arena_bytes(n);
    const size_t dec = (size_t)mid * 3 + ((size_t)mid / JPK_ANS_CHUNK + 2) * 1100 + (1u << 20);
    if (enc > need) need = enc;
    if (inv > need) need = inv;
    if (dec > need) need = dec;
    return jpk_arena_ensure(ctx, need);
}

// host-logic probe: arena bytes stage `stage` (0 forward BWT, 1 rANS encode of text-like data, 2 inverse BWT, 3 rANS decode bound,
// 4 rANS encode of the densest data: what the arena grows to when such a block arrives) plans for one
// block of block_bytes -- what jpk_ctx_reserve takes the maximum of (DESIGN.md section 3 quotes these)
extern "C" int64_t jpk_debug_arena_bytes(int64_t block_bytes, int stage)
{
    if (block_bytes < 0 || block_bytes > (int64_t)JPK_MAX_BLOCKSIZE) return JPK_E_ARG;
    const uint32_t n = (uint32_t)block_bytes, mid = n + JPK_TRAILER_BYTES;
    switch (stage) {
    case 0: return (int64_t)jpk_fwd_bwt_arena_bytes(n);
    case 1: return (int64_t)jpk_ans_encode_arena_bytes(mid);
    case 2: return (int64_t)jpk_inv_bwt_arena_bytes(n);
    case 3: return (int64_t)((size_t)mid * 3 + ((size_t)mid / JPK_ANS_CHUNK + 2) * 1100 + (1u << 20));
    case 4: return (int64_t)jpk_ans_encode_arena_bytes_worst(mid);
    default: return JPK_E_ARG;
    }
}

extern "C" const char *jpk_strerror(int s)
{
    switch (s) {
    case JPK_OK: return "ok";
    case JPK_E_ARG: return "invalid argument";
    case JPK_E_CAPACITY: return "output buffer too small";
    case JPK_E_CORRUPT: return "corrupt or misaligned stream";
    case JPK_E_DEVICE: return "HIP runtime error";
    case JPK_E_ALLOC: return "allocation failed";
    case JPK_E_NODEVICE: return "no gfx950 device";
    default: return "unknown status";
    }
}

extern "C" const char *jpk_version(void) { return "jampack_amd 0.1 (gfx950)"; }

// ---- device-buffer entry points ----------------------------------------------------------------------------
// Blocks in their forward BWT or entropy encode right now, PER DEVICE: the encoder's launch grouping follows the load of the GPU
// the block runs on -- with one thread per GPU (jpk_init over eight devices) every block is alone on its device, whatever the
// other seven are doing.
namespace { std::atomic<int> g_compress_inflight[64]; }
int jpk_compress_inflight_enter(int device)
{
    if (device < 0 || device >= 64) return 1;
    return g_compress_inflight[device].fetch_add(1, std::memory_order_relaxed) + 1;
}
void jpk_compress_inflight_leave(int device)
{
    if (device >= 0 && device < 64) g_compress_inflight[device].fetch_sub(1, std::memory_order_relaxed);
}
// launch groups of the entropy encoder for a block of `nch` chunks that is one of `inflight` blocks on its device (ans_enc.hip):
// graded groups shorten ONE block and cost throughput when other blocks fill the machine anyway
int jpk_enc_groups_for(int inflight, uint32_t nch)
{
    int ngroups = (int)(nch / 8u);
    const int gmax = inflight <= 1 ? jpk_ctx::ENC_GROUPS : (inflight == 2 ? 2 : 1);
    if (ngroups > gmax) ngroups = gmax;
    if (ngroups < 1) ngroups = 1;
    return ngroups;
}
// test hooks (host logic only, no device call): the per-device accounting and the grouping rule.  The hooks that CHANGE live state
// (this one with delta != 0, jpk_debug_combiner_fail_next) work only in a process that sets JPK_DEBUG_HOOKS=1: no caller of the
// product library can move the in-flight count that drives the encoder's launch grouping by accident.
static bool debug_hooks_on() { const char *e = getenv("JPK_DEBUG_HOOKS"); return e && atoi(e) != 0; }
extern "C" int jpk_debug_compress_inflight(int device, int delta)
{
    if (device < 0 || device >= 64) return JPK_E_ARG;
    if (delta != 0 && !debug_hooks_on()) return JPK_E_ARG;
    if (delta > 0) return jpk_compress_inflight_enter(device);
    if (delta < 0) { jpk_compress_inflight_leave(device); return g_compress_inflight[device].load(std::memory_order_relaxed); }
    return g_compress_inflight[device].load(std::memory_order_relaxed);
}
extern "C" int jpk_debug_enc_groups(int device, int32_t nch)
{
    if (device < 0 || device >= 64 || nch < 0) return JPK_E_ARG;
    // what a block that arrives on `device` now would choose (it counts itself)
    return jpk_enc_groups_for(g_compress_inflight[device].load(std::memory_order_relaxed) + 1, (uint32_t)nch);
}

#define JPK_ENTER(ctx)                         \
    if (!(ctx)) return JPK_E_ARG;              \
    JPK_HIP(hipSetDevice((ctx)->device))

extern "C" int jpk_dev_bwt_forward(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_out || !out_len || in_len < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    if ((int64_t)in_len + JPK_TRAILER_BYTES > (int64_t)out_cap) return JPK_E_CAPACITY;
    JPK_TRY(jpk_fwd_bwt_device(ctx, d_in, in_len, d_out));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    jpk_sa_stats_sync(ctx);
    *out_len = in_len + JPK_TRAILER_BYTES;
    return JPK_OK;
}

extern "C" int jpk_dev_bwt_inverse(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_in || !d_out || !out_len || in_len < 0) return JPK_E_ARG;
    if (in_len < JPK_TRAILER_BYTES) return JPK_E_CORRUPT;
    if (in_len - JPK_TRAILER_BYTES > out_cap) return JPK_E_CAPACITY;
    JPK_TRY(jpk_inv_bwt_device(ctx, d_in, in_len, d_out));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = in_len - JPK_TRAILER_BYTES;
    return JPK_OK;
}

extern "C" int jpk_dev_bwt_inverse_chains120(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len,
                                            float *chase_ms)
{
    JPK_ENTER(ctx);
    if (!d_in || !d_out || !out_len || in_len < 0) return JPK_E_ARG;
    if (in_len < JPK_TRAILER_BYTES) return JPK_E_CORRUPT;
    if (in_len - JPK_TRAILER_BYTES > out_cap) return JPK_E_CAPACITY;
    JPK_TRY(jpk_inv_bwt_chains120_device(ctx, d_in, in_len, d_out, chase_ms));
    *out_len = in_len - JPK_TRAILER_BYTES;
    return JPK_OK;
}

extern "C" int jpk_dev_ans_encode(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_out || !out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    return jpk_ans_encode_device(ctx, d_in, in_len, d_out, out_cap, out_len);
}

extern "C" int jpk_dev_ans_decode(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_out || !out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    return jpk_ans_decode_device(ctx, d_in, in_len, d_out, out_cap, out_len);
}

extern "C" int jpk_dev_rank_encode(jpk_ctx *ctx, uint8_t *d_t, int32_t *d_freq256, int32_t len)
{
    JPK_ENTER(ctx);
    if (!d_freq256 || len < 0 || (len > 0 && !d_t)) return JPK_E_ARG;
    JPK_TRY(jpk_rank_encode_device(ctx, d_t, d_freq256, len));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    return JPK_OK;
}

extern "C" int jpk_dev_rank_decode(jpk_ctx *ctx, uint8_t *d_ranks, const int32_t *d_freq256, int32_t len)
{
    JPK_ENTER(ctx);
    if (!d_freq256 || len < 0 || (len > 0 && !d_ranks)) return JPK_E_ARG;
    return jpk_rank_decode_device(ctx, d_ranks, d_freq256, len);
}

// fused tail of Jampack::Comp(): the BWT image stays in HBM (ctx->stage_out) between the stages
extern "C" int jpk_dev_block_compress(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_out || !out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    if ((uint32_t)in_len >= JPK_FWD_BWT_LIMIT) return JPK_E_ARG;      // before anything is allocated for it (jpk_fwd_bwt_device)
    const size_t mid = (size_t)in_len + JPK_TRAILER_BYTES;
    JPK_TRY(buf_ensure(ctx, &ctx->stage_out, &ctx->stage_out_cap, mid));
    if (in_len < JPK_BWT_UNITS) JPK_HIP(hipMemsetAsync(ctx->stage_out, 0, mid, ctx->stream));   // untouched trailer: defined bytes
    // the gate stays closed from the first kernel of the suffix sort to the last wide kernel of the entropy stage; the encoder
    // opens it (jpk_gate_leave) once those are enqueued, in front of its first host synchronisation
    JPK_TRY(jpk_gate_enter(ctx));
    int rc = jpk_fwd_bwt_device(ctx, d_in, in_len, ctx->stage_out);
    if (rc == JPK_OK) rc = jpk_ans_encode_device(ctx, ctx->stage_out, (int32_t)mid, d_out, out_cap, out_len);    // synchronises the stream
    jpk_gate_leave(ctx);                     // no-op unless an error kept the encoder from doing it
    if (rc == JPK_OK) jpk_sa_stats_sync(ctx);
    return rc;
}

extern "C" int jpk_dev_block_decompress(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_out || !out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    const size_t mid_cap = (size_t)out_cap + JPK_TRAILER_BYTES;
    JPK_TRY(buf_ensure(ctx, &ctx->stage_out, &ctx->stage_out_cap, mid_cap));
    int32_t mid = 0;
    JPK_TRY(jpk_ans_decode_device(ctx, d_in, in_len, ctx->stage_out, (int32_t)(mid_cap > 0x7fffffff ? 0x7fffffff : mid_cap), &mid));
    if (mid < JPK_TRAILER_BYTES) return JPK_E_CORRUPT;
    JPK_TRY(jpk_inv_bwt_device(ctx, ctx->stage_out, mid, d_out));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = mid - JPK_TRAILER_BYTES;
    return JPK_OK;
}

// ---- batches of independent blocks -------------------------------------------------------------------------------------------
// The serial entropy-decode kernels run one wave per 1 MiB chunk; a 64 MiB block keeps 65 of the chip's 1024 SIMDs busy.  A
// batch runs ONE grid per stage over the chunks of all its blocks, so sixteen blocks fill the chip whatever the hardware
// queues do with concurrent kernels (tools/conctest.hip).
extern "C" int jpk_dev_blocks_ans_decode(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                                         const int32_t *out_cap, int32_t *out_len, int32_t *status)
{
    JPK_ENTER(ctx);
    if (nblocks < 0 || (nblocks > 0 && (!d_in || !in_len || !d_out || !out_cap || !out_len))) return JPK_E_ARG;
    if (nblocks == 0) return JPK_OK;
    std::vector<int32_t> st_local((size_t)nblocks);
    int32_t *stp = status ? status : st_local.data();
    for (int b = 0; b < nblocks; b++)
        if (in_len[b] < 0 || out_cap[b] < 0 || !d_out[b] || (in_len[b] > 0 && !d_in[b])) return JPK_E_ARG;
    JPK_TRY(jpk_ans_decode_batch(ctx, nblocks, d_in, in_len, d_out, out_cap, out_len, stp, 0));
    if (!status)
        for (int b = 0; b < nblocks; b++) if (stp[b] != JPK_OK) return stp[b];
    return JPK_OK;
}

extern "C" int jpk_dev_blocks_decompress(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                                         const int32_t *out_cap, int32_t *out_len, int32_t *status)
{
    JPK_ENTER(ctx);
    if (nblocks < 0 || (nblocks > 0 && (!d_in || !in_len || !d_out || !out_cap || !out_len))) return JPK_E_ARG;
    if (nblocks == 0) return JPK_OK;
    std::vector<int32_t> st_local((size_t)nblocks), mid_cap((size_t)nblocks), mid_len((size_t)nblocks);
    int32_t *stp = status ? status : st_local.data();
    size_t mid_total = 0, bound = 0;
    uint32_t nmax = 1;
    for (int b = 0; b < nblocks; b++) {
        if (in_len[b] < 0 || out_cap[b] < 0 || !d_out[b] || (in_len[b] > 0 && !d_in[b])) return JPK_E_ARG;
        const int64_t mc = (int64_t)out_cap[b] + JPK_TRAILER_BYTES;
        if (mc > 0x7fffffffLL) return JPK_E_ARG;
        mid_cap[b] = (int32_t)mc;
        mid_total += jpk_align((size_t)mc + 64);
        if ((uint32_t)out_cap[b] > nmax) nmax = (uint32_t)out_cap[b];
        // what the batch decoder can need for this block at most: rank array + 2-byte RLE0 symbols + chunk tables
        bound += (size_t)mc * 3 + ((size_t)in_len[b] / 275 + 2) * 1100 + 4096;
    }
    // arena: [inverse-BWT scratch of the largest block x lanes][one verdict per block][BWT images of all blocks][the batch
    // decoder's buffers]; sized once so that it cannot move while the images sit in it.
    // Lanes: an inverse BWT is ~30 dependent launches of which only the walk fills the chip (1.6 of 2.5 ms for 64 MiB, the rest
    // is 20-microsecond rank-jump rounds and launch gaps), so up to three run side by side, each on its own stream with its own
    // scratch, behind one event on the decode stream.
    static const int max_lanes = [] { const char *e = getenv("JPK_INV_LANES"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > jpk_ctx::ENC_GROUPS ? jpk_ctx::ENC_GROUPS : v); }();
    // ... and up to JPK_INV_LANES_SMALL (default 12) when the batch is many SMALL blocks (<= 4 MiB): an inverse BWT of a 1 MiB block is
    // the same ~30 dependent launches, each a sliver of the chip, and 256 of them on three lanes took 100 ms of a 245 ms call whose
    // chains had finished after 145 (round 4: 1 070 -> 1 400 MB/s with 8 to 16 lanes; what is left is the host's launch rate, 7 700
    // launches per call).  Blocks of 8 MiB gain nothing (nine chains per block: the chains are the call) and would pay 12 scratch areas.
    static const int small_lanes = [] { const char *e = getenv("JPK_INV_LANES_SMALL"); const int v = e ? atoi(e) : 12; return v < 1 ? 1 : (v > jpk_ctx::INV_LANES_MAX ? jpk_ctx::INV_LANES_MAX : v); }();
    // Only a batch with many blocks gets lanes: they save the latency part of each inverse BWT (~1 ms of 2.5), nothing next to
    // the 270 ms of a small batch's chains, and every extra stream of a context lands on a hardware queue that another
    // context's chain kernels may be using (eight contexts decoding 2-block passes lost 18 % with a second stream each).
    const bool many_small = nblocks >= 32 && nmax <= (4u << 20);
    // ... or, by default, through ONE set of launches over all of them (bwt_inv.hip jpk_inv_bwt_batch_enqueue; JPK_INV_BATCH=0 keeps the lanes)
    static const bool batch_on = [] { const char *e = getenv("JPK_INV_BATCH"); return e ? atoi(e) != 0 : true; }();
    // (in groups of consecutive blocks whose scratch fits JPK_INV_BATCH_GIB, default 8.  Batches of LARGE blocks gain nothing from it --
    // 16 / 32 / 64 blocks of 64 MiB: 310 / 350 / 438 ms against 309 / 348 / 435 on the three lanes: their walks fill the chip -- and stay
    // on the lanes.)
    static const size_t batch_budget = [] { const char *e = getenv("JPK_INV_BATCH_GIB"); const long v = e ? atol(e) : 8; return (size_t)(v < 1 ? 1 : v) << 30; }();
    const bool batched = batch_on && many_small;
    int lanes = (nblocks < 8 || batched) ? 1 : (many_small ? (small_lanes > max_lanes ? small_lanes : max_lanes) : max_lanes);
    hipStream_t lane_stream[jpk_ctx::INV_LANES_MAX] = {};
    hipEvent_t lane_event[jpk_ctx::INV_LANES_MAX] = {};
    lane_stream[0] = ctx->stream;
    for (int k = 1; k < lanes; k++) {
        hipStream_t *slot = k < jpk_ctx::ENC_GROUPS ? &ctx->aux[k - 1] : &ctx->inv_lane[k];
        hipEvent_t *ev = k < jpk_ctx::ENC_GROUPS ? &ctx->ev_done[k] : &ctx->ev_inv[k];
        if ((!*slot && hipStreamCreateWithFlags(slot, hipStreamNonBlocking) != hipSuccess) ||
            (!*ev && hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess)) { lanes = k; break; }
        lane_stream[k] = *slot;
        lane_event[k] = *ev;
    }
    const size_t inv_one = jpk_align(jpk_inv_bwt_arena_bytes(nmax), 4096), verdict_bytes = jpk_align((size_t)nblocks * 16 + 64, 4096);
    // groups of consecutive blocks whose batch scratch fits the budget (many small blocks: one group)
    std::vector<int> group_end;
    size_t batch_bytes_max = 0;
    if (batched) {
        int b0 = 0;
        while (b0 < nblocks) {
            // (extended block by block: a job adds its own bytes and tiles, the tables follow from the totals; at most 65 535 jobs --
            // the batch launches use blockIdx.y = the job)
            int e = b0 + 1;
            size_t job_bytes = 0, tiles = 0;
            jpk_inv_bwt_batch_plan_add(mid_cap[(size_t)b0], &job_bytes, &tiles);
            size_t need = jpk_inv_bwt_batch_plan_total(1, job_bytes, tiles);
            while (e < nblocks && e - b0 < JPK_INV_BATCH_MAX_JOBS) {
                size_t jb2 = job_bytes, t2 = tiles;
                jpk_inv_bwt_batch_plan_add(mid_cap[(size_t)e], &jb2, &t2);
                const size_t more = jpk_inv_bwt_batch_plan_total(e + 1 - b0, jb2, t2);
                if (more > batch_budget) break;
                need = more; job_bytes = jb2; tiles = t2;
                e++;
            }
            if (need > batch_bytes_max) batch_bytes_max = need;
            group_end.push_back(e);
            b0 = e;
        }
    }
    const size_t inv_bytes = (batched ? jpk_align(batch_bytes_max, 4096) : inv_one * (size_t)lanes) + verdict_bytes;
    JPK_TRY(jpk_arena_ensure(ctx, inv_bytes + mid_total + bound + (1u << 20)));
    uint32_t *d_verdict = reinterpret_cast<uint32_t *>(ctx->arena + inv_bytes - verdict_bytes);
    std::vector<uint8_t *> mid((size_t)nblocks);
    {
        size_t off = inv_bytes;
        for (int b = 0; b < nblocks; b++) { mid[b] = ctx->arena + off; off += jpk_align((size_t)mid_cap[b] + 64); }
    }
    JPK_TRY(jpk_ans_decode_batch(ctx, nblocks, d_in, in_len, mid.data(), mid_cap.data(), mid_len.data(), stp, inv_bytes + mid_total));
    // the inverse BWTs are enqueued without a host round trip between them (trailer index, slot count and the head check stay
    // on the device); their verdicts come back in one copy
    std::vector<int> ran;
    std::vector<std::vector<uint8_t>> host_jobs;               // the batched inverse BWTs' job tables: copied from here asynchronously
    {
        struct Restore {                                       // whatever happens below, the context gets its stream and base back
            jpk_ctx *c; hipStream_t s; hipStream_t *lane; int lanes; bool failed = true;
            ~Restore()
            {
                c->stream = s; c->arena_base = 0;
                // an early return leaves inverse BWTs queued on the lane streams that read the images and write the arena and the
                // callers' buffers: nothing may reuse the arena before they have drained (the main stream never waited for them)
                if (failed) for (int k = 1; k < lanes; k++) if (lane[k]) (void)hipStreamSynchronize(lane[k]);
            }
        } restore{ctx, ctx->stream, lane_stream, lanes};
        hipStream_t main_stream = ctx->stream;
        if (lanes > 1) {
            JPK_HIP(hipEventRecord(ctx->ev_batch, main_stream));       // the decoded images are complete
            for (int k = 1; k < lanes; k++) JPK_HIP(hipStreamWaitEvent(lane_stream[k], ctx->ev_batch, 0));
        }
        int next = 0;
        std::vector<int> jb;                                   // blocks that go through the batched inverse BWT
        for (int b = 0; b < nblocks; b++) {
            out_len[b] = 0;
            if (stp[b] != JPK_OK) continue;
            if (mid_len[b] < JPK_TRAILER_BYTES) { stp[b] = JPK_E_CORRUPT; continue; }
            if (mid_len[b] - JPK_TRAILER_BYTES > out_cap[b]) { stp[b] = JPK_E_CAPACITY; continue; }
            if (batched && mid_len[b] - JPK_TRAILER_BYTES >= JPK_BWT_UNITS) { jb.push_back(b); continue; }   // (a shorter image has no sorted part: two copies, below)
            const int k = next % lanes;
            ctx->stream = lane_stream[k];
            ctx->arena_base = inv_one * (size_t)k;
            const int rc = jpk_inv_bwt_enqueue(ctx, mid[b], mid_len[b], d_out[b], d_verdict + 4 * (size_t)b);
            if (rc != JPK_OK) { stp[b] = rc; continue; }
            ran.push_back(b);
            next++;
        }
        ctx->stream = main_stream;
        // one batch per group, one after the other on the main stream (they share the scratch at the start of the arena)
        size_t jq = 0;
        for (size_t g = 0; g < group_end.size() && jq < jb.size(); g++) {
            std::vector<int> part;
            while (jq < jb.size() && jb[jq] < group_end[g]) part.push_back(jb[jq++]);
            if (part.empty()) continue;
            std::vector<const uint8_t *> jin(part.size());
            std::vector<uint8_t *> jout(part.size());
            std::vector<int32_t> jlen(part.size());
            for (size_t q = 0; q < part.size(); q++) { jin[q] = mid[part[q]]; jout[q] = d_out[part[q]]; jlen[q] = mid_len[part[q]]; }
            // (job q reports into the verdict slot of its block, part[q])
            ctx->arena_base = 0;
            host_jobs.emplace_back();
            const int rc = jpk_inv_bwt_batch_enqueue(ctx, (int)part.size(), jin.data(), jlen.data(), jout.data(), d_verdict, part.data(), host_jobs.back());
            if (rc != JPK_OK) { for (int b : part) stp[b] = rc; }
            else for (int b : part) ran.push_back(b);
        }
        for (int k = 1; k < lanes; k++) {                      // the main stream continues behind every lane
            JPK_HIP(hipEventRecord(lane_event[k], lane_stream[k]));
            JPK_HIP(hipStreamWaitEvent(main_stream, lane_event[k], 0));
        }
        restore.failed = false;
    }
    std::vector<uint32_t> verdict((size_t)nblocks * 4);
    if (!ran.empty()) JPK_HIP(hipMemcpyAsync(verdict.data(), d_verdict, (size_t)nblocks * 16, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->prof_on) jpk_prof_resolve(ctx);
    for (int b : ran) {
        if (verdict[4 * (size_t)b]) { stp[b] = JPK_E_CORRUPT; continue; }
        out_len[b] = mid_len[b] - JPK_TRAILER_BYTES;
    }
    if (!status)
        for (int b = 0; b < nblocks; b++) if (stp[b] != JPK_OK) return stp[b];
    return JPK_OK;
}

// ---- probes ------------------------------------------------------------------------------------------------
extern "C" int jpk_dev_suffix_array(jpk_ctx *ctx, const uint8_t *d_t, int32_t n, int32_t *d_sa)
{
    JPK_ENTER(ctx);
    if (n < 0 || (n > 0 && (!d_t || !d_sa))) return JPK_E_ARG;
    JPK_TRY(jpk_suffix_array_device(ctx, d_t, n, d_sa));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    jpk_sa_stats_sync(ctx);
    return JPK_OK;
}

extern "C" int jpk_dev_sort_pairs_u64(jpk_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, int32_t n, int32_t bit_lo, int32_t bit_hi)
{
    JPK_ENTER(ctx);
    if (n < 0 || bit_lo < 0 || bit_hi > 64 || bit_lo > bit_hi || (n > 0 && (!d_keys || !d_vals))) return JPK_E_ARG;
    if (n == 0) return JPK_OK;
    Arena plan(ctx, true);
    plan.get<uint64_t>(n); plan.get<uint32_t>(n); plan.get<uint32_t>(jpk_radix_scratch_words(n));
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    uint64_t *ka = real.get<uint64_t>(n);
    uint32_t *va = real.get<uint32_t>(n);
    uint32_t *sc = real.get<uint32_t>(jpk_radix_scratch_words(n));
    int shifts[8], ns = 0;
    for (int s = bit_lo; s < bit_hi; s += 8) shifts[ns++] = s;
    JPK_TRY(jpk_radix_sort_pairs_u64(ctx, d_keys, d_vals, ka, va, (size_t)n, shifts, ns, sc));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    return JPK_OK;
}

extern "C" int jpk_dev_exclusive_scan_u32(jpk_ctx *ctx, uint32_t *d_data, int32_t n, uint32_t *total)
{
    JPK_ENTER(ctx);
    if (n < 0 || (n > 0 && !d_data)) return JPK_E_ARG;
    Arena plan(ctx, true);
    plan.get<uint32_t>(jpk_scan_scratch_words(n));
    JPK_TRY(jpk_arena_ensure(ctx, plan.need));
    Arena real(ctx, false);
    uint32_t *sc = real.get<uint32_t>(jpk_scan_scratch_words(n));
    JPK_TRY(jpk_exclusive_sum_u32(ctx, d_data, d_data, (size_t)n, sc, ctx->d_mail));
    uint32_t t = 0;
    JPK_TRY(jpk_read_mail(ctx, &t, 1));
    if (total) *total = t;
    return JPK_OK;
}

extern "C" int jpk_dev_rle_encode(jpk_ctx *ctx, const uint8_t *d_ranks, int32_t len, uint16_t *d_rle, int32_t *rlen)
{
    JPK_ENTER(ctx);
    if (len < 0 || !rlen || (len > 0 && (!d_ranks || !d_rle))) return JPK_E_ARG;
    return jpk_rle_encode_device(ctx, d_ranks, len, d_rle, rlen);
}

extern "C" int jpk_dev_model_pairs(jpk_ctx *ctx, const uint16_t *d_rle, int32_t rlen, uint32_t *d_pairs)
{
    JPK_ENTER(ctx);
    if (rlen < 0 || (rlen > 0 && (!d_rle || !d_pairs))) return JPK_E_ARG;
    return jpk_model_pairs_device(ctx, d_rle, rlen, d_pairs);
}

// ---- block container: checksum + 15-byte frame (jampack.cpp:29-60, 122-164) ---------------------------------
extern "C" int jpk_dev_checksum(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint32_t *crc)
{
    JPK_ENTER(ctx);
    if (!crc || in_len < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    JPK_TRY(jpk_checksum_device(ctx, d_in, in_len, ctx->d_mail));
    return jpk_read_mail(ctx, crc, 1);
}

namespace {
bool jam_block_size_ok(int32_t bs) { return bs >= JPK_MIN_BLOCKSIZE && bs <= JPK_MAX_BLOCKSIZE; }
}

extern "C" int jpk_dev_jam_block_write(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, int32_t block_size, uint8_t *d_out, int32_t out_cap,
                                       int32_t *out_len)
{
    JPK_ENTER(ctx);
    if (!d_out || !out_len || in_len < 0 || (in_len > 0 && !d_in)) return JPK_E_ARG;
    if (!jam_block_size_ok(block_size) || in_len > block_size) return JPK_E_ARG;     // InitComp, jampack.cpp:70
    if (out_cap < JPK_JAM_HEADER_BYTES) return JPK_E_CAPACITY;
    uint32_t crc = 0;
    JPK_TRY(jpk_dev_checksum(ctx, d_in, in_len, &crc));
    int32_t n = 0;
    JPK_TRY(jpk_dev_block_compress(ctx, d_in, in_len, d_out + JPK_JAM_HEADER_BYTES, out_cap - JPK_JAM_HEADER_BYTES, &n));
    uint8_t *h = reinterpret_cast<uint8_t *>(ctx->h_mail + 128);    // pinned; words 128.. are not used by jpk_read_mail callers
    memcpy(h, "JAM", 3);
    memcpy(h + 3, &crc, 4);
    memcpy(h + 7, &n, 4);
    memcpy(h + 11, &block_size, 4);
    JPK_HIP(hipMemcpyAsync(d_out, h, JPK_JAM_HEADER_BYTES, hipMemcpyHostToDevice, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = n + JPK_JAM_HEADER_BYTES;
    return JPK_OK;
}

extern "C" int jpk_dev_jam_block_read(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len,
                                      int32_t *consumed)
{
    JPK_ENTER(ctx);
    if (!d_in || !d_out || !out_len || in_len < 0 || out_cap < 0) return JPK_E_ARG;
    if (in_len < JPK_JAM_HEADER_BYTES) return JPK_E_CORRUPT;
    uint8_t *h = reinterpret_cast<uint8_t *>(ctx->h_mail + 128);
    JPK_HIP(hipMemcpyAsync(h, d_in, JPK_JAM_HEADER_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t crc;
    int32_t csize, block_size;
    memcpy(&crc, h + 3, 4);
    memcpy(&csize, h + 7, 4);
    memcpy(&block_size, h + 11, 4);
    // DecompReadBlock, jampack.cpp:150: "Refusing to read from corrupt header!"
    if (memcmp(h, "JAM", 3) != 0 || !jam_block_size_ok(block_size) || csize < 0 || csize > JPK_MAX_BLOCKSIZE) return JPK_E_CORRUPT;
    if ((int64_t)csize + JPK_JAM_HEADER_BYTES > in_len) return JPK_E_CORRUPT;
    int32_t n = 0;
    JPK_TRY(jpk_dev_block_decompress(ctx, d_in + JPK_JAM_HEADER_BYTES, csize, d_out, out_cap, &n));
    uint32_t got = 0;
    JPK_TRY(jpk_dev_checksum(ctx, d_out, n, &got));
    if (got != crc) return JPK_E_CORRUPT;                            // "Detected corrupt block!", jampack.cpp:59
    *out_len = n;
    if (consumed) *consumed = csize + JPK_JAM_HEADER_BYTES;
    return JPK_OK;
}

// ---- host-buffer (drop-in) entry points --------------------------------------------------------------------
// Re-entrant from the OpenMP block loop of jampack.cpp:215/313: every calling thread works on a context of its own
// (stream + HBM arena + staging buffers).  Contexts live in a process-wide pool: a thread borrows one on its first
// call and hands it back when it exits, so the arenas persist across thread teams (the reference pays five
// cudaMalloc/cudaFree per block instead, bwt.cpp:195-239).  Threads are dealt round robin over the devices selected
// by jpk_init(device_mask) -- `jampack -t 8` on an 8-GPU node puts one block on each GPU.
namespace {
struct CtxPool {
    std::mutex mu;
    std::vector<int> devices;                       // selected devices (empty = not initialised yet)
    std::vector<std::vector<jpk_ctx *>> idle;       // per selected device: contexts nobody holds
    std::vector<jpk_ctx *> all;                     // every context the pool has created
    uint64_t generation = 1;                        // bumped by jpk_shutdown: borrowed handles of older generations are dead
    uint32_t next_thread = 0;
    int init_rc = JPK_OK;
};
CtxPool &pool() { static CtxPool p; return p; }

// caller holds pool().mu
int pool_select(CtxPool &p, uint64_t mask)
{
    p.devices.clear();
    p.idle.clear();
    const int n = jpk_device_count();
    if (n <= 0) return JPK_E_NODEVICE;
    for (int d = 0; d < n && d < 64; d++) {
        if (mask && !((mask >> d) & 1u)) continue;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("JPK_ALLOW_ANY_ARCH")) continue;
        p.devices.push_back(d);
    }
    if (p.devices.empty()) return JPK_E_NODEVICE;
    p.idle.resize(p.devices.size());
    return JPK_OK;
}

struct TlsCtx {
    jpk_ctx *ctx = nullptr;
    uint64_t generation = 0;
    size_t slot = 0;
    ~TlsCtx()
    {
        if (!ctx) return;
        CtxPool &p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        if (generation == p.generation && slot < p.idle.size()) p.idle[slot].push_back(ctx);   // else: jpk_shutdown already destroyed it
    }
};
thread_local TlsCtx tls;

int tls_ctx(jpk_ctx **out)
{
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    if (tls.ctx && tls.generation == p.generation) { *out = tls.ctx; return JPK_OK; }
    tls.ctx = nullptr;
    if (p.devices.empty()) {
        // no jpk_init(): JPK_DEVICE pins one device, otherwise every visible gfx950 device takes part
        uint64_t mask = 0;
        if (const char *e = getenv("JPK_DEVICE")) { const int d = atoi(e); if (d < 0 || d >= 64) return JPK_E_NODEVICE; mask = 1ull << d; }
        JPK_TRY(pool_select(p, mask));
    }
    const size_t slot = (size_t)(p.next_thread++ % p.devices.size());
    jpk_ctx *c = nullptr;
    if (!p.idle[slot].empty()) { c = p.idle[slot].back(); p.idle[slot].pop_back(); }
    else {
        JPK_TRY(jpk_ctx_create(&c, p.devices[slot], nullptr));
        p.all.push_back(c);
    }
    tls.ctx = c; tls.generation = p.generation; tls.slot = slot;
    *out = c;
    return JPK_OK;
}
}  // namespace

// ---- jpk_dev_blocks_compress: blocks in flight inside the library ---------------------------------------------------------
namespace {
// worker contexts of the batch compress call, per device.  They are the library's own (a list apart from the host-buffer pool:
// a device-API call must not make jpk_init() believe that host-buffer contexts exist); jpk_shutdown destroys them too.
std::vector<std::vector<jpk_ctx *>> &batch_idle() { static std::vector<std::vector<jpk_ctx *>> v(64); return v; }
std::vector<jpk_ctx *> &batch_all() { static std::vector<jpk_ctx *> v; return v; }

int batch_ctx_acquire(int device, jpk_ctx **out)
{
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    auto &idle = batch_idle()[(size_t)device];
    if (!idle.empty()) { *out = idle.back(); idle.pop_back(); return JPK_OK; }
    jpk_ctx *c = nullptr;
    JPK_TRY(jpk_ctx_create(&c, device, nullptr));
    batch_all().push_back(c);
    *out = c;
    return JPK_OK;
}
void batch_ctx_release(int device, jpk_ctx *c, uint64_t generation)
{
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    if (generation == p.generation) batch_idle()[(size_t)device].push_back(c);      // else jpk_shutdown already destroyed it
}
}  // namespace

namespace {
// Small blocks -- the reference's default block is 8 MiB, its smallest 1 MiB (format.hpp:20-22), and Jampack::Compress feeds
// `Threads` of them at a time (jampack.cpp:205-224) -- are compressed in GROUPS: one suffix sort over the blocks of a group
// (jpk_fwd_bwt_group_device), one set of entropy grids over all their chunks (jpk_ans_encode_group_device), one host
// synchronisation per stage boundary of a group (four in all, below) instead of ~200 launches and a synchronisation per block.  Bytes per block are those of
// jpk_dev_block_compress.
constexpr int32_t GROUP_BLOCK_MAX = 16 << 20;       // blocks up to this size are grouped
// Bytes per group: large groups amortise best (a 256 MiB stream of 1 MiB blocks: 4.6 GB/s in groups of 64 MiB, 4.2 in groups of
// 16, 2.9 in groups of 8; of 8 MiB blocks: 3.6 / 3.5 / 3.1 -- profiles/r04_group_size.txt), but a short stream still wants several
// groups in flight: a quarter of the small blocks' bytes, between 8 and 64 MiB.  JPK_GROUP_MIB fixes it.
size_t group_target_bytes(size_t small_bytes_total)
{
    static const long fixed = [] { const char *e = getenv("JPK_GROUP_MIB"); return e ? atol(e) : 0L; }();
    if (fixed > 0) return (size_t)(fixed > 512 ? 512 : fixed) << 20;
    size_t t = small_bytes_total / 4;
    if (t < ((size_t)8 << 20)) t = (size_t)8 << 20;
    if (t > ((size_t)64 << 20)) t = (size_t)64 << 20;
    return t;
}

struct Task { int first, count; };
// the work list of jpk_dev_blocks_compress: every block exactly once, in order; consecutive blocks of <= GROUP_BLOCK_MAX bytes form
// groups of at most 256 blocks and (beyond the first block) at most the target size; a larger block is a task of its own
void plan_tasks(int nblocks, const int32_t *in_len, std::vector<Task> &tasks)
{
    static const bool grouping = [] { const char *e = getenv("JPK_GROUP"); return e ? atoi(e) != 0 : true; }();
    size_t small_total = 0;
    for (int k = 0; k < nblocks; k++) if (in_len[k] <= GROUP_BLOCK_MAX) small_total += (size_t)in_len[k];
    const size_t target = group_target_bytes(small_total);
    int b = 0;
    while (b < nblocks) {
        if (!grouping || in_len[b] > GROUP_BLOCK_MAX) { tasks.push_back(Task{b, 1}); b++; continue; }
        int e = b;
        size_t bytes = 0;
        while (e < nblocks && in_len[e] <= GROUP_BLOCK_MAX && e - b < 256 && (e == b || bytes + (size_t)in_len[e] <= target)) { bytes += (size_t)in_len[e]; e++; }
        tasks.push_back(Task{b, e - b});
        b = e;
    }
}

// what a group of these blocks needs: the staging buffer for the images and the arena (the larger of the two stages' layouts + the
// encoder's capacity sink)
void group_needs(int nb, const int32_t *in_len, size_t *stage_bytes, size_t *arena_bytes)
{
    uint64_t nlen_total = 0;
    uint32_t nch = 0;
    for (int b = 0; b < nb; b++) {
        nch += ((uint32_t)in_len[b] + JPK_TRAILER_BYTES + JPK_ANS_CHUNK - 1) / JPK_ANS_CHUNK;
        nlen_total += (uint32_t)in_len[b] - (uint32_t)in_len[b] % JPK_BWT_UNITS;
    }
    *stage_bytes = (size_t)nch * JPK_ANS_CHUNK;
    const size_t a_sort = jpk_fwd_bwt_group_arena_bytes((uint32_t)nlen_total, nb), a_enc = jpk_ans_encode_group_arena_bytes(nch, nb);
    *arena_bytes = (a_sort > a_enc ? a_sort : a_enc) + (size_t)GROUP_BLOCK_MAX + (1u << 20);
}

// test hook (JPK_DEBUG_HOOKS=1 only): the next `n` groups fail as a whole before they run, as if their arena could not be had --
// every block of such a group must then come back through the single-block path with the same bytes
std::atomic<int> g_group_fail_next{0};

int group_compress(jpk_ctx *c, int nb, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out, const int32_t *out_cap, int32_t *out_len,
                   int32_t *status)
{
    if (g_group_fail_next.load() > 0 && g_group_fail_next.fetch_sub(1) > 0) return JPK_E_ALLOC;
    std::vector<uint32_t> first((size_t)nb);
    std::vector<int32_t> mid((size_t)nb);
    uint64_t nlen_total = 0;
    uint32_t nch = 0;
    for (int b = 0; b < nb; b++) {
        first[b] = nch;
        mid[b] = in_len[b] + JPK_TRAILER_BYTES;
        nch += ((uint32_t)mid[b] + JPK_ANS_CHUNK - 1) / JPK_ANS_CHUNK;
        nlen_total += (uint32_t)in_len[b] - (uint32_t)in_len[b] % JPK_BWT_UNITS;
    }
    (void)nlen_total;
    size_t stage_bytes, arena_bytes;
    group_needs(nb, in_len, &stage_bytes, &arena_bytes);
    JPK_TRY(buf_ensure(c, &c->stage_out, &c->stage_out_cap, stage_bytes));
    JPK_TRY(jpk_arena_ensure(c, arena_bytes));
    std::vector<uint8_t *> img((size_t)nb);
    for (int b = 0; b < nb; b++) {
        img[b] = c->stage_out + (size_t)first[b] * JPK_ANS_CHUNK;
        // bwt.cpp:35: a block shorter than 120 bytes leaves its 480 trailer bytes alone -- defined bytes here, as in jpk_dev_block_compress
        if (in_len[b] < JPK_BWT_UNITS) JPK_HIP(hipMemsetAsync(img[b], 0, (size_t)mid[b], c->stream));
    }
    JPK_TRY(jpk_gate_enter(c));
    int rc = jpk_fwd_bwt_group_device(c, nb, d_in, in_len, img.data());
    if (rc == JPK_OK) rc = jpk_ans_encode_group_device(c, nb, c->stage_out, first.data(), mid.data(), d_out, out_cap, out_len, status);   // synchronises
    jpk_gate_leave(c);
    if (rc == JPK_OK) jpk_sa_stats_sync(c);
    else (void)hipStreamSynchronize(c->stream);            // the pinned tables and the staged images stay valid until the queue is empty
    return rc;
}
}  // namespace

namespace {
// "block k's input has arrived on the device": set by the copier thread of the multi-device entry (multi_run), waited for -- on the host,
// task by task -- by the workers below, so that block k + 1 is still on its way in while block k is being compressed
struct BlocksReady {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> ok;
    bool failed = false;
    explicit BlocksReady(size_t n) : ok(n, 0) {}
    void set(size_t k) { { std::lock_guard<std::mutex> g(mu); ok[k] = 1; } cv.notify_all(); }
    void fail() { { std::lock_guard<std::mutex> g(mu); failed = true; } cv.notify_all(); }
    bool wait(int first, int count)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] {
            if (failed) return true;
            for (int k = first; k < first + count; k++) if (!ok[(size_t)k]) return false;
            return true;
        });
        return !failed;
    }
};
int blocks_compress_body(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                         const int32_t *out_cap, int32_t *out_len, int32_t *status, int32_t in_flight, BlocksReady *ready);
}  // namespace

extern "C" int jpk_dev_blocks_compress(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                                       const int32_t *out_cap, int32_t *out_len, int32_t *status, int32_t in_flight)
{
    return blocks_compress_body(ctx, nblocks, d_in, in_len, d_out, out_cap, out_len, status, in_flight, nullptr);
}

namespace {
int blocks_compress_body(jpk_ctx *ctx, int32_t nblocks, const uint8_t *const *d_in, const int32_t *in_len, uint8_t *const *d_out,
                         const int32_t *out_cap, int32_t *out_len, int32_t *status, int32_t in_flight, BlocksReady *ready)
{
    JPK_ENTER(ctx);
    if (nblocks < 0 || (nblocks > 0 && (!d_in || !in_len || !d_out || !out_cap || !out_len))) return JPK_E_ARG;
    if (nblocks == 0) return JPK_OK;
    if (ctx->device < 0 || ctx->device >= 64) return JPK_E_ARG;
    for (int b = 0; b < nblocks; b++)
        if (in_len[b] < 0 || out_cap[b] < 0 || !d_out[b] || (in_len[b] > 0 && !d_in[b])) return JPK_E_ARG;
    std::vector<int32_t> st_local((size_t)nblocks);
    int32_t *stp = status ? status : st_local.data();
    // tasks: a large block, or a group of consecutive small ones (up to the target size / 256 blocks)
    std::vector<Task> tasks;
    plan_tasks(nblocks, in_len, tasks);
    const int ntasks = (int)tasks.size();
    // every worker sizes its staging buffer and arena for the largest group once, before it takes its first task: which worker gets
    // which group changes from call to call, and an arena that grows in the middle of a call costs a free + malloc + synchronise
    size_t max_stage = 0, max_arena = 0;
    for (const Task &t : tasks)
        if (t.count > 1) {
            size_t sb, ab;
            group_needs(t.count, in_len + t.first, &sb, &ab);
            if (sb > max_stage) max_stage = sb;
            if (ab > max_arena) max_arena = ab;
        } else if (in_len[t.first] <= (128 << 20)) {        // (larger blocks grow a worker's arena when it meets one: ten arenas of a 1000 MiB block would not fit)
            // ... and for the largest single block (end of round 6): a worker whose first call brought it the stream's short last blocks only grew
            // its arena in the middle of the NEXT call -- one call in six of a fresh process took 4.1-4.3 GB/s instead of 6.1-6.5
            const uint32_t n1 = (uint32_t)in_len[t.first];
            const size_t mid = (size_t)n1 + JPK_TRAILER_BYTES;
            const size_t fb = jpk_fwd_bwt_arena_bytes(n1), eb = jpk_ans_encode_arena_bytes((uint32_t)mid);
            if (mid > max_stage) max_stage = mid;
            if (fb > max_arena) max_arena = fb;
            if (eb > max_arena) max_arena = eb;
        }
    int nw = in_flight > 0 ? in_flight : 10;       // (10 since round 6, like the bench line: profiles/r06_blocks_in_flight.txt; 8 in rounds 4-5; an arena is 3.7 GB)
    if (nw > ntasks) nw = ntasks;
    if (nw > 16) nw = 16;
    uint64_t generation;
    { CtxPool &p = pool(); std::lock_guard<std::mutex> g(p.mu); generation = p.generation; }
    // Stream order: the workers' streams are the library's own.  Whatever the caller has queued on ctx->stream before this call
    // (the kernels that produce d_in[], the consumers of a previous d_out[]) must be ordered in front of them, as it is for
    // jpk_dev_block_compress on the caller's context: an event on ctx->stream that every worker stream waits for.  The call
    // returns when every block is complete (each worker synchronises its stream), so nothing needs ordering at the exit.
    JPK_HIP(hipEventRecord(ctx->ev_batch, ctx->stream));
    std::atomic<int> next{0};
    auto work = [&](jpk_ctx *c) {
        if (hipSetDevice(c->device) != hipSuccess) return;                  // a fresh thread starts on device 0
        if (c != ctx && hipStreamWaitEvent(c->stream, ctx->ev_batch, 0) != hipSuccess) return;   // leaves its share to the others
        if (max_arena && (buf_ensure(c, &c->stage_out, &c->stage_out_cap, max_stage) != JPK_OK || jpk_arena_ensure(c, max_arena) != JPK_OK)) {
            // no room for the largest group on this context: the groups say so themselves when they get here (single blocks may still fit)
        }
        for (;;) {
            const int k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= ntasks) return;
            const int b = tasks[(size_t)k].first, nb = tasks[(size_t)k].count;
            if (ready && !ready->wait(b, nb)) {                 // the copy of an input failed: the blocks nobody has taken yet say so
                for (int j = b; j < b + nb; j++) { out_len[j] = 0; stp[j] = JPK_E_DEVICE; }
                continue;
            }
            if (nb == 1) {
                out_len[b] = 0;
                stp[b] = jpk_dev_block_compress(c, d_in[b], in_len[b], d_out[b], out_cap[b], &out_len[b]);
                continue;
            }
            const int rc = group_compress(c, nb, d_in + b, in_len + b, d_out + b, out_cap + b, out_len + b, stp + b);
            if (rc != JPK_OK)                                   // the group as a whole failed (its ~46 B/byte arena or staging buffer, a device error):
                for (int j = b; j < b + nb; j++) {              // every block goes through the single-block path on this context (~50 B/byte of ITS size), as before grouping
                    out_len[j] = 0;
                    stp[j] = jpk_dev_block_compress(c, d_in[j], in_len[j], d_out[j], out_cap[j], &out_len[j]);
                }
        }
    };
    // workers 1 .. nw-1 on contexts of their own; a worker that cannot get one (or a thread that cannot be started) leaves its
    // share to the others
    std::vector<std::thread> threads;
    std::vector<jpk_ctx *> held;
    for (int k = 1; k < nw; k++) {
        jpk_ctx *c = nullptr;
        if (batch_ctx_acquire(ctx->device, &c) != JPK_OK) break;
        held.push_back(c);
        try { threads.emplace_back(work, c); } catch (...) { break; }      // std::system_error: the caller's thread does that share
    }
    work(ctx);
    for (auto &t : threads) t.join();
    for (jpk_ctx *c : held) batch_ctx_release(ctx->device, c, generation);
    if (!status)
        for (int b = 0; b < nblocks; b++) if (stp[b] != JPK_OK) return stp[b];
    return JPK_OK;
}
}  // namespace

extern "C" int jpk_debug_group_fail_next(int n)
{
    if (!debug_hooks_on() || n < 0) return JPK_E_ARG;
    g_group_fail_next.store(n);
    return JPK_OK;
}

// host-logic probe (no device call): the tasks jpk_dev_blocks_compress would form for these block lengths: task t covers blocks
// [first[t], first[t] + count[t]); returns the number of tasks (at most nblocks)
extern "C" int jpk_debug_group_plan(int32_t nblocks, const int32_t *in_len, int32_t *first, int32_t *count)
{
    if (nblocks < 0 || (nblocks > 0 && (!in_len || !first || !count))) return JPK_E_ARG;
    for (int b = 0; b < nblocks; b++) if (in_len[b] < 0) return JPK_E_ARG;
    std::vector<Task> tasks;
    plan_tasks(nblocks, in_len, tasks);
    for (size_t t = 0; t < tasks.size(); t++) { first[t] = tasks[t].first; count[t] = tasks[t].count; }
    return (int)tasks.size();
}

// ---- jpk_blocks_compress_multi: the block loop of Jampack::Compress over the GPUs of one node, natively ---------------------------
// Blocks are independent (jampack.cpp:215-219), so block b goes to device devices[b mod G] -- one worker thread and context per device,
// the thread moves its blocks in from host memory and compresses them there -- and the only exchange is the gather of the compressed
// blocks on the root (the first device of the mask), in block order = the order of the in-order CompWriteBlock loop
// (jampack.cpp:220-224): sizes are known on the host, so every non-root device sends each block with ONE ncclSend of exactly its
// bytes and the root posts the matching ncclRecv at the block's final offset (grouped: ncclGroupStart/End over a single-process
// communicator from ncclCommInitAll; every pair of GPUs has a direct xGMI link).  RCCL is loaded at first use (dlopen): the library
// has no link-time dependency on it, and a one-device mask never touches it unless JPK_MULTI_FORCE_RCCL=1 asks for the root's own
// blocks to travel through a send/receive to itself (exercises the path on a 1-GPU box).
#include <dlfcn.h>
namespace {
typedef struct jpkNcclComm *jpk_nccl_comm_t;
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(jpk_nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(jpk_nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, jpk_nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, jpk_nccl_comm_t, hipStream_t) = nullptr;
    bool ok = false;
};
constexpr int NCCL_UINT8 = 1;                        // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)
Rccl &rccl()
{
    static Rccl r = [] {
        Rccl q;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { q.h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (q.h) break; }
        if (!q.h) return q;
        q.CommInitAll = (decltype(q.CommInitAll))dlsym(q.h, "ncclCommInitAll");
        q.CommDestroy = (decltype(q.CommDestroy))dlsym(q.h, "ncclCommDestroy");
        q.GroupStart = (decltype(q.GroupStart))dlsym(q.h, "ncclGroupStart");
        q.GroupEnd = (decltype(q.GroupEnd))dlsym(q.h, "ncclGroupEnd");
        q.Send = (decltype(q.Send))dlsym(q.h, "ncclSend");
        q.Recv = (decltype(q.Recv))dlsym(q.h, "ncclRecv");
        q.ok = q.CommInitAll && q.CommDestroy && q.GroupStart && q.GroupEnd && q.Send && q.Recv;
        return q;
    }();
    return r;
}
// one communicator set per device list, kept until jpk_shutdown
struct MultiComm { std::vector<int> devices; std::vector<jpk_nccl_comm_t> comms; };
std::deque<MultiComm> &multi_comms() { static std::deque<MultiComm> v; return v; }      // (a deque: a call holds a pointer into it while another call, on other devices, adds a set)
std::mutex &multi_mu() { static std::mutex m; return m; }

// devices of the mask that exist among `ndev` visible devices, ascending (mask 0 = all)
std::vector<int> multi_devices(uint64_t mask, int ndev)
{
    std::vector<int> d;
    for (int k = 0; k < ndev && k < 64; k++)
        if (!mask || ((mask >> k) & 1u)) d.push_back(k);
    return d;
}
}  // namespace

// host-logic probe (no device call): owner[b] = the device block b of `nblocks` would run on for `device_mask` when `ndev_visible`
// devices are visible; returns the number of devices taking part (the root is the first of them)
extern "C" int jpk_debug_multi_plan(uint64_t device_mask, int32_t ndev_visible, int32_t nblocks, int32_t *owner)
{
    if (ndev_visible < 0 || nblocks < 0 || (nblocks > 0 && !owner)) return JPK_E_ARG;
    const std::vector<int> dv = multi_devices(device_mask, ndev_visible);
    if (dv.empty()) return JPK_E_NODEVICE;
    for (int b = 0; b < nblocks; b++) owner[b] = dv[(size_t)b % dv.size()];
    return (int)dv.size();
}

// Shared body of the two multi-device entries.  One call at a time PER DEVICE: a call holds the mutexes of the devices of its mask
// (taken in ascending order) from its first device call to its last, so two calls on disjoint device sets -- two files on two halves
// of a node -- run side by side, and calls that share a device queue (round 5: one process-wide mutex).  What the locks protect: the
// per-device slabs and copy streams below, and the RCCL communicators, which are cached per device LIST and are not safe for
// concurrent use -- two calls with the same list share all its devices and therefore exclude each other.  jpk_shutdown takes every
// device's mutex before it destroys slabs and communicators.
namespace {
std::mutex *multi_dev_mu() { static std::mutex m[64]; return m; }
struct MultiLocks {                                          // the devices of a call, ascending; released in reverse
    std::vector<int> held;
    explicit MultiLocks(const std::vector<int> &devs) { for (int d : devs) if (d >= 0 && d < 64) { multi_dev_mu()[d].lock(); held.push_back(d); } }
    ~MultiLocks() { for (size_t k = held.size(); k-- > 0;) multi_dev_mu()[held[k]].unlock(); }
    MultiLocks(const MultiLocks &) = delete;
    MultiLocks &operator=(const MultiLocks &) = delete;
};
// per device: one input slab and one output slab, kept between calls (grown when a larger call arrives), and the stream the inputs
// travel on while the context's stream computes; freed by jpk_shutdown / jpk_release_idle
struct MultiSlab { uint8_t *in = nullptr, *out = nullptr; size_t in_cap = 0, out_cap = 0; hipStream_t copy = nullptr; };
MultiSlab *multi_slabs() { static MultiSlab v[64]; return v; }
int slab_ensure(uint8_t **p, size_t *cap, size_t bytes)
{
    if (*cap >= bytes && *p) return JPK_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = jpk_align(bytes + bytes / 8 + 4096, 1 << 20);
    if (hipMalloc((void **)p, want) != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return JPK_E_ALLOC; }
    *cap = want;
    return JPK_OK;
}
// (the caller holds device d's mutex and restores the current device)
void multi_slab_free(int d)
{
    MultiSlab &m = multi_slabs()[d];
    if (!m.in && !m.out && !m.copy) return;
    if (hipSetDevice(d) == hipSuccess) {
        if (m.copy) { (void)hipStreamSynchronize(m.copy); (void)hipStreamDestroy(m.copy); }
        if (m.in) (void)hipFree(m.in);
        if (m.out) (void)hipFree(m.out);
    }
    m = MultiSlab();
}
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};
size_t multi_comp_cap(int32_t len) { return (size_t)((int64_t)(len + JPK_TRAILER_BYTES) * 5 / 4) + 4096 + 1400 * ((size_t)(len + JPK_TRAILER_BYTES) / JPK_ANS_CHUNK + 1); }

// compress: in[b] host blocks of in_len[b] bytes -> compressed blocks; decompress: in[b] host compressed blocks, raw_len[b] = the
// block's decompressed size.  Either way block b runs on devices[b mod G] through the library's batch entry on that device
// (jpk_dev_blocks_compress with `in_flight` blocks in flight and grouped small blocks / jpk_dev_blocks_decompress: one pass over the
// chunks of all its blocks), and the results are gathered in block order on the first device.
int multi_run(bool compress, uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, const int32_t *raw_len, uint8_t *d_out, int64_t out_cap,
              int64_t *out_off, int32_t *status, int32_t in_flight)
{
    if (nblocks < 0 || out_cap < 0 || !out_off || (nblocks > 0 && (!in || !in_len || !d_out || (!compress && !raw_len)))) return JPK_E_ARG;
    out_off[0] = 0;
    if (nblocks == 0) return JPK_OK;
    for (int b = 0; b < nblocks; b++)
        if (in_len[b] < 0 || (in_len[b] > 0 && !in[b]) || (!compress && raw_len[b] < 0)) return JPK_E_ARG;
    std::vector<int32_t> st_local((size_t)nblocks), olen((size_t)nblocks, 0);
    int32_t *stp = status ? status : st_local.data();
    for (int b = 0; b < nblocks; b++) { stp[b] = JPK_E_DEVICE; out_off[b + 1] = 0; }   // a block nobody got to says so (a device whose worker bails out early)
    const int ndev = jpk_device_count();
    if (ndev <= 0) return JPK_E_NODEVICE;
    const std::vector<int> cand = multi_devices(device_mask, ndev);
    const MultiLocks locks(cand);                            // (ascending; before the first device call of this call)
    DeviceRestore restore;                                   // the caller's current device comes back on every exit path
    std::vector<int> devs;
    for (int d : cand) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("JPK_ALLOW_ANY_ARCH")) continue;
        devs.push_back(d);
    }
    if (devs.empty()) return JPK_E_NODEVICE;
    const int G = (int)devs.size(), root = devs[0];
    uint64_t generation;
    { CtxPool &p = pool(); std::lock_guard<std::mutex> g(p.mu); generation = p.generation; }

    struct Dev { jpk_ctx *c = nullptr; MultiSlab *slab = nullptr; std::vector<int> blocks; std::vector<size_t> ioff, ooff; int rc = JPK_OK; };
    std::vector<Dev> dv((size_t)G);
    for (int b = 0; b < nblocks; b++) dv[(size_t)(b % G)].blocks.push_back(b);
    auto cap_of = [&](int b) { return compress ? multi_comp_cap(in_len[b]) : (size_t)raw_len[b]; };
    auto work = [&](int g) {
        Dev &D = dv[(size_t)g];
        if (D.blocks.empty()) return;
        const int dev = devs[(size_t)g];
        if (hipSetDevice(dev) != hipSuccess) { D.rc = JPK_E_DEVICE; return; }
        if ((D.rc = batch_ctx_acquire(dev, &D.c)) != JPK_OK) return;
        D.slab = &multi_slabs()[dev];
        size_t itotal = 0, ototal = 0;
        for (int b : D.blocks) {
            D.ioff.push_back(itotal); itotal += jpk_align((size_t)in_len[b] + 64, 256);
            D.ooff.push_back(ototal); ototal += jpk_align(cap_of(b), 256);
        }
        if ((D.rc = slab_ensure(&D.slab->in, &D.slab->in_cap, itotal + 256)) != JPK_OK) return;
        if ((D.rc = slab_ensure(&D.slab->out, &D.slab->out_cap, ototal + 256)) != JPK_OK) return;
        if (!D.slab->copy && hipStreamCreateWithFlags(&D.slab->copy, hipStreamNonBlocking) != hipSuccess) { D.slab->copy = nullptr; D.rc = JPK_E_DEVICE; return; }
        const size_t nb = D.blocks.size();
        std::vector<const uint8_t *> din(nb);
        std::vector<uint8_t *> dout(nb);
        std::vector<int32_t> ilen(nb), ocap(nb), ol(nb, 0), stl(nb, JPK_E_DEVICE);
        for (size_t k = 0; k < nb; k++) {
            const int b = D.blocks[k];
            din[k] = D.slab->in + D.ioff[k];
            dout[k] = D.slab->out + D.ooff[k];
            ilen[k] = in_len[b];
            const size_t cap = cap_of(b);
            ocap[k] = (int32_t)(cap > 0x7fffffff ? 0x7fffffff : cap);
        }
        int rc;
        if (compress) {
            // Round 6: the inputs travel on the slab's own stream, block by block in the order the batch entry takes them, from a
            // copier thread; a worker of the batch entry waits (on the host) only for the blocks of ITS next task, so block k + 1 is on
            // its way in while block k is being compressed -- the reference's loop overlaps its reads the same way, thread k compresses
            // while thread k + 1 still reads (jampack.cpp:205-224).  Rounds 4-5 copied ALL of a device's blocks in before its first kernel.
            // (Pageable host memory: the runtime stages it through pinned buffers of its own; the copies run at PCIe speed on this box,
            // tools/pcietest.hip.)
            BlocksReady ready(nb);
            std::thread copier([&] {
                if (hipSetDevice(dev) != hipSuccess) { ready.fail(); return; }
                for (size_t k = 0; k < nb; k++) {
                    const int b = D.blocks[k];
                    if (in_len[b] && (hipMemcpyAsync(D.slab->in + D.ioff[k], in[b], (size_t)in_len[b], hipMemcpyHostToDevice, D.slab->copy) != hipSuccess ||
                                      hipStreamSynchronize(D.slab->copy) != hipSuccess)) { ready.fail(); return; }
                    ready.set(k);
                }
            });
            rc = blocks_compress_body(D.c, (int32_t)nb, din.data(), ilen.data(), dout.data(), ocap.data(), ol.data(), stl.data(), in_flight, &ready);
            copier.join();
        } else {
            // the decode is ONE pass over the chunks of all the device's blocks and needs every stream in place; compressed inputs are a
            // fifth of the data and the pass is long: the copies stay in front of it, on the context's stream
            for (size_t k = 0; k < nb; k++) {
                const int b = D.blocks[k];
                if (in_len[b] && hipMemcpyAsync(D.slab->in + D.ioff[k], in[b], (size_t)in_len[b], hipMemcpyHostToDevice, D.c->stream) != hipSuccess) { D.rc = JPK_E_DEVICE; return; }
            }
            rc = jpk_dev_blocks_decompress(D.c, (int32_t)nb, din.data(), ilen.data(), dout.data(), ocap.data(), ol.data(), stl.data());
        }
        for (size_t k = 0; k < nb; k++) {
            const int b = D.blocks[k];
            stp[b] = rc != JPK_OK && stl[k] == JPK_OK ? rc : stl[k];
            olen[(size_t)b] = stp[b] == JPK_OK ? ol[k] : 0;
        }
        if (rc != JPK_OK) D.rc = rc;
    };
    {
        std::vector<std::thread> th;
        for (int g = 1; g < G; g++) { try { th.emplace_back(work, g); } catch (...) { dv[(size_t)g].rc = JPK_E_DEVICE; } }
        work(0);
        for (auto &t : th) t.join();
    }
    // What is reported and what is gathered: a block's status is its own; a device that failed as a whole fails all its
    // blocks; EVERY block whose status is JPK_OK is gathered -- a corrupt frame among healthy ones costs its own bytes only, the others
    // are in d_out at [out_off[b], out_off[b + 1]) as the header promises -- and when the gather itself cannot run (the output buffer
    // is too small, RCCL is not there) NO block is reported as done: every status is an error and every range is empty.
    int rc_blocks = JPK_OK;
    for (int g = 0; g < G; g++)
        if (dv[(size_t)g].rc != JPK_OK) {
            if (rc_blocks == JPK_OK) rc_blocks = dv[(size_t)g].rc;
            for (int b : dv[(size_t)g].blocks) if (stp[b] == JPK_OK) { stp[b] = dv[(size_t)g].rc; olen[(size_t)b] = 0; }     // a failed device: none of its blocks is reported as done
        }
    for (int b = 0; b < nblocks && rc_blocks == JPK_OK; b++) if (stp[b] != JPK_OK) rc_blocks = stp[b];
    int64_t total = 0;
    for (int b = 0; b < nblocks; b++) { out_off[b] = total; total += olen[(size_t)b]; }
    out_off[nblocks] = total;
    int rc = JPK_OK;                                         // of the gather
    if (total > out_cap) rc = JPK_E_CAPACITY;

    // the gather: root's own blocks are device-to-device copies; the others travel over RCCL, one send / receive pair per block.
    // A communicator set belongs to a device LIST and is used under the mutexes of all its devices (this call holds them).
    static const bool force_rccl = [] { const char *e = getenv("JPK_MULTI_FORCE_RCCL"); return e && atoi(e) != 0; }();
    const bool use_rccl = rc == JPK_OK && total > 0 && (G > 1 || force_rccl);
    const std::vector<jpk_nccl_comm_t> *comms = nullptr;
    if (use_rccl) {
        std::lock_guard<std::mutex> lk(multi_mu());
        if (!rccl().ok) rc = JPK_E_DEVICE;
        else {
            for (auto &m : multi_comms()) if (m.devices == devs) comms = &m.comms;
            if (!comms) {
                MultiComm m;
                m.devices = devs;
                m.comms.resize((size_t)G);
                if (rccl().CommInitAll(m.comms.data(), G, devs.data()) != 0) rc = JPK_E_DEVICE;
                else { multi_comms().push_back(m); comms = &multi_comms().back().comms; }
            }
        }
    }
    if (rc == JPK_OK && total > 0) {
        if (hipSetDevice(root) != hipSuccess) rc = JPK_E_DEVICE;
        Dev &R = dv[0];
        hipStream_t rs = R.c ? R.c->stream : nullptr;
        if (rc == JPK_OK && use_rccl && comms) {
            bool grouped = rccl().GroupStart() == 0;
            for (int g = 0; g < G && grouped; g++) {
                Dev &D = dv[(size_t)g];
                if ((g == 0 && !force_rccl) || !D.c || !D.slab) continue;
                for (size_t k = 0; k < D.blocks.size(); k++) {
                    const int b = D.blocks[k];
                    if (olen[(size_t)b] == 0) continue;
                    if (rccl().Send(D.slab->out + D.ooff[k], (size_t)olen[(size_t)b], NCCL_UINT8, 0, (*comms)[(size_t)g], D.c->stream) != 0) rc = JPK_E_DEVICE;
                    if (rccl().Recv(d_out + out_off[b], (size_t)olen[(size_t)b], NCCL_UINT8, g, (*comms)[0], rs) != 0) rc = JPK_E_DEVICE;
                }
            }
            if (!grouped || rccl().GroupEnd() != 0) rc = JPK_E_DEVICE;
        }
        if (rc == JPK_OK && !(use_rccl && force_rccl) && R.c && R.slab)
            for (size_t k = 0; k < R.blocks.size(); k++) {
                const int b = R.blocks[k];
                if (olen[(size_t)b] && hipMemcpyAsync(d_out + out_off[b], R.slab->out + R.ooff[k], (size_t)olen[(size_t)b], hipMemcpyDeviceToDevice, rs) != hipSuccess) rc = JPK_E_DEVICE;
            }
    }
    // every stream that took part has finished before the slabs can be used by the next call (also on the error paths)
    for (int g = 0; g < G; g++) {
        Dev &D = dv[(size_t)g];
        if (!D.c) continue;
        if (hipSetDevice(devs[(size_t)g]) != hipSuccess || hipStreamSynchronize(D.c->stream) != hipSuccess) rc = rc == JPK_OK ? JPK_E_DEVICE : rc;
        batch_ctx_release(devs[(size_t)g], D.c, generation);
    }
    if (rc != JPK_OK) {                                      // no gather: nothing is in d_out, and nobody is told otherwise
        for (int b = 0; b < nblocks; b++) { if (stp[b] == JPK_OK) stp[b] = rc; out_off[b] = 0; }
        out_off[nblocks] = 0;
        return rc;
    }
    return rc_blocks;
}
}  // namespace

// host-logic probe (no device call): takes the device mutexes a multi-device call with this mask would take (bit d = device d; every
// bit counts, whether such a device exists or not), holds them for hold_ms milliseconds and lets go.  Two probes with disjoint masks
// overlap, two that share a bit queue: tests/test_abi_and_host.py.
extern "C" int jpk_debug_multi_lock_probe(uint64_t device_mask, int32_t hold_ms)
{
    if (hold_ms < 0 || hold_ms > 10000) return JPK_E_ARG;
    std::vector<int> devs;
    for (int d = 0; d < 64; d++) if ((device_mask >> d) & 1u) devs.push_back(d);
    const MultiLocks locks(devs);
    std::this_thread::sleep_for(std::chrono::milliseconds(hold_ms));
    return (int)devs.size();
}

extern "C" int jpk_blocks_compress_multi(uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, uint8_t *d_out, int64_t out_cap,
                                         int64_t *out_off, int32_t *status)
{
    return multi_run(true, device_mask, nblocks, in, in_len, nullptr, d_out, out_cap, out_off, status, 0);
}
extern "C" int jpk_blocks_compress_multi_ex(uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, uint8_t *d_out, int64_t out_cap,
                                            int64_t *out_off, int32_t *status, int32_t in_flight)
{
    return multi_run(true, device_mask, nblocks, in, in_len, nullptr, d_out, out_cap, out_off, status, in_flight);
}
extern "C" int jpk_blocks_decompress_multi(uint64_t device_mask, int32_t nblocks, const uint8_t *const *in, const int32_t *in_len, const int32_t *raw_len, uint8_t *d_out,
                                           int64_t out_cap, int64_t *out_off, int32_t *status)
{
    return multi_run(false, device_mask, nblocks, in, in_len, raw_len, d_out, out_cap, out_off, status, 0);
}

extern "C" int jpk_init(uint64_t device_mask)
{
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    if (!p.all.empty()) return JPK_E_ARG;           // contexts exist: jpk_shutdown() first
    JPK_TRY(pool_select(p, device_mask));
    p.next_thread = 0;
    return (int)p.devices.size();
}

extern "C" int jpk_init_devices(int32_t *devices, int32_t cap)
{
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    const int n = (int)p.devices.size();
    for (int i = 0; i < n && i < cap && devices; i++) devices[i] = p.devices[(size_t)i];
    return n;
}

extern "C" int jpk_thread_device(void)
{
    jpk_ctx *ctx;
    int rc = tls_ctx(&ctx);
    return rc == JPK_OK ? ctx->device : rc;
}

extern "C" void jpk_shutdown(void)
{
    std::vector<int> every;
    for (int d = 0; d < 64; d++) every.push_back(d);
    const MultiLocks multi_lock(every);                         // no multi-device call is using the communicators / slabs destroyed below
    DeviceRestore restore;                                      // (destroying contexts and slabs moves the current device around: ADVICE r5)
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    for (jpk_ctx *c : p.all) jpk_ctx_destroy(c);    // synchronises each context's streams first
    p.all.clear();
    p.idle.clear();
    for (jpk_ctx *c : batch_all()) jpk_ctx_destroy(c);      // the batch-compress workers' contexts
    batch_all().clear();
    for (auto &v : batch_idle()) v.clear();
    {
        std::lock_guard<std::mutex> lk(multi_mu());
        for (auto &m : multi_comms())
            for (jpk_nccl_comm_t cm : m.comms) if (cm && rccl().ok) (void)rccl().CommDestroy(cm);
        multi_comms().clear();
    }
    for (int d = 0; d < 64; d++) multi_slab_free(d);
    p.devices.clear();
    p.generation++;
    p.next_thread = 0;
}

// Gives back what the batch entries keep between calls and nobody is using: the idle worker contexts of jpk_dev_blocks_compress /
// jpk_dev_blocks_decompress (an arena of ~54 bytes per block byte each) and the multi-device entries' slabs.  The host-buffer pool's
// contexts stay (threads hold them).  Safe beside running calls: a context that is in use is not in the idle lists.  Returns the
// number of contexts destroyed.
extern "C" int jpk_release_idle(void)
{
    int freed = 0;
    DeviceRestore restore;                                      // jpk_ctx_destroy and the slabs' hipFree set the device: the caller's comes back
    for (int d = 0; d < 64; d++) {
        std::unique_lock<std::mutex> dev_lock(multi_dev_mu()[d], std::try_to_lock);    // a multi-device call in flight on the device keeps its slabs
        if (dev_lock.owns_lock()) multi_slab_free(d);
    }
    CtxPool &p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    for (auto &idle : batch_idle()) {
        for (jpk_ctx *c : idle) {
            auto &all = batch_all();
            all.erase(std::remove(all.begin(), all.end(), c), all.end());
            jpk_ctx_destroy(c);
            freed++;
        }
        idle.clear();
    }
    return freed;
}

namespace {
typedef int (*dev_fn)(jpk_ctx *, const uint8_t *, int32_t, uint8_t *, int32_t, int32_t *);

// H2D -> device entry -> D2H.  prefill_out: copy the caller's out bytes to the device first (used where the
// reference leaves part of the output untouched).
int staged(dev_fn fn, const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, bool prefill_out)
{
    if (!out || !out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !in)) return JPK_E_ARG;
    jpk_ctx *ctx;
    JPK_TRY(tls_ctx(&ctx));
    JPK_HIP(hipSetDevice(ctx->device));
    JPK_TRY(buf_ensure(ctx, &ctx->stage_in, &ctx->stage_in_cap, (size_t)in_len + 64));
    // the fused entry points use stage_out as their intermediate, so the host-visible result gets its own buffer
    JPK_TRY(buf_ensure(ctx, &ctx->stage_res, &ctx->stage_res_cap, (size_t)out_cap + 64));
    uint8_t *d_res = ctx->stage_res;
    static const bool timing = getenv("JPK_TIME_HOST") != nullptr;      // JPK_TIME_HOST=1: where a host-buffer call spends its time
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = timing ? now() : 0;
    if (in_len) JPK_HIP(hipMemcpyAsync(ctx->stage_in, in, (size_t)in_len, hipMemcpyHostToDevice, ctx->stream));
    if (prefill_out && out_cap) JPK_HIP(hipMemcpyAsync(d_res, out, (size_t)out_cap, hipMemcpyHostToDevice, ctx->stream));
    if (timing) (void)hipStreamSynchronize(ctx->stream);
    const double t1 = timing ? now() : 0;
    int32_t n = 0;
    JPK_TRY(fn(ctx, ctx->stage_in, in_len, d_res, out_cap, &n));
    if (timing) (void)hipStreamSynchronize(ctx->stream);
    const double t2 = timing ? now() : 0;
    if (n > 0) JPK_HIP(hipMemcpyAsync(out, d_res, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    if (timing) fprintf(stderr, "[jampack_amd] host call: in %d B, out %d B: H2D %.2f ms, device %.2f ms, D2H %.2f ms\n", in_len, n, t1 - t0, t2 - t1, now() - t2);
    *out_len = n;
    return JPK_OK;
}
}  // namespace

extern "C" int jpk_bwt_forward(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    if (in_len >= 0 && (int64_t)in_len + JPK_TRAILER_BYTES > (int64_t)out_cap) return JPK_E_CAPACITY;
    // a block shorter than 120 bytes leaves the trailer untouched (bwt.cpp:35): round-trip the caller's bytes
    const bool keep = in_len >= 0 && in_len < JPK_BWT_UNITS;
    return staged(jpk_dev_bwt_forward, in, in_len, out, keep ? in_len + JPK_TRAILER_BYTES : out_cap, out_len, keep);
}

extern "C" int jpk_bwt_inverse(const uint8_t *in, int32_t in_len_with_trailer, uint8_t *out, int32_t out_cap, int32_t *out_len,
                               int32_t threads, int32_t use_gpu)
{
    (void)threads; (void)use_gpu;      // Options.Threads / Options.Gpu do not change the bytes (bwt.cpp:92-132)
    return staged(jpk_dev_bwt_inverse, in, in_len_with_trailer, out, out_cap, out_len, false);
}

extern "C" int jpk_ans_encode(uint8_t *in_clobbered, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    return staged(jpk_dev_ans_encode, in_clobbered, in_len, out, out_cap, out_len, false);
}

// ---- decode combiner: concurrent host-buffer Ans::Decode calls share one batched grid ---------------------------------------
// The reference decodes `Threads` chunks at a time inside one Ans::Decode call (ans.cpp:254-264) and its multi-block loop calls
// Decomp() from `Threads` OpenMP threads at once (jampack.cpp:313).  On the GPU one block is 65 single-wave chains on 1024
// SIMDs, so the rate lives in the batch (jpk_ans_decode_batch: ONE grid per serial kernel over the chunks of all blocks).  A
// plain drop-in caller never sees the batch ABI; instead, calls that arrive from different threads within a short window are
// merged here: every thread stages its own input (its context, its stream), the first one to arrive leads -- it waits until
// nobody is on the way in any more (threads between entry and submission are counted) plus a grace period that applies only
// when the process has shown concurrency before, decodes all requests in one pass on a combiner context of the device, and
// wakes the others, which copy their own results back.  A lone caller runs exactly the single-block path at once; a thread
// that arrives while a batch is already on the GPU leads a new batch immediately (nobody ever waits for a running batch).
#include <condition_variable>
namespace {
struct DecReq {
    const uint8_t *d_in; int32_t in_len; uint8_t *d_out; int32_t out_cap;
    hipEvent_t staged;              // the request's input has reached the device
    int32_t out_len = 0, status = JPK_OK;
    bool done = false;
    bool alone = false;             // the merged pass did not take place for this request: its own thread decodes it on its own context
};
struct Combiner {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<DecReq *> pending;
    bool collecting = false;        // a leader is gathering `pending`
    std::atomic<int> arriving{0};   // threads between entry and submission
    int last_batch = 1;             // requests of the most recent batch (> 1: the process decodes concurrently)
};
Combiner &combiner(int device) { static Combiner c[64]; return c[device & 63]; }
// grace in microseconds the leader grants late arrivals once the process has shown concurrent decode calls; < 0 = combiner off
int combine_grace_us()
{
    static const int v = [] { const char *e = getenv("JPK_COMBINE_US"); return e ? atoi(e) : 300; }();
    return v;
}
}  // namespace

// test hook (JPK_DEBUG_HOOKS=1 only): the next `n` merged decode passes are treated as failed as a whole before they run, as if the
// combiner context could not be had -- every merged request must then come back through its own thread's single-block path
namespace { std::atomic<int> g_combiner_fail_next{0}; }
extern "C" int jpk_debug_combiner_fail_next(int n)
{
    if (!debug_hooks_on() || n < 0) return JPK_E_ARG;
    g_combiner_fail_next.store(n);
    return JPK_OK;
}

// test hook: requests the most recent combined decode on `device` carried (1 = a lone caller took the single-block path)
extern "C" int jpk_debug_combiner_last_batch(int device)
{
    if (device < 0 || device >= 64) return JPK_E_ARG;
    Combiner &cb = combiner(device);
    std::lock_guard<std::mutex> lk(cb.mu);
    return cb.last_batch;
}

extern "C" int jpk_ans_decode(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, int32_t threads)
{
    (void)threads;
    if (combine_grace_us() < 0) return staged(jpk_dev_ans_decode, in, in_len, out, out_cap, out_len, false);
    if (!out || !out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !in)) return JPK_E_ARG;
    jpk_ctx *ctx;
    JPK_TRY(tls_ctx(&ctx));
    JPK_HIP(hipSetDevice(ctx->device));
    if (ctx->device < 0 || ctx->device >= 64 || in_len == 0) return staged(jpk_dev_ans_decode, in, in_len, out, out_cap, out_len, false);
    Combiner &cb = combiner(ctx->device);
    cb.arriving.fetch_add(1);
    struct Arrived { Combiner &c; bool in = true; ~Arrived() { if (in) c.arriving.fetch_sub(1); } void submitted() { if (in) { c.arriving.fetch_sub(1); in = false; } } } arrived{cb};
    JPK_TRY(buf_ensure(ctx, &ctx->stage_in, &ctx->stage_in_cap, (size_t)in_len + 64));
    JPK_TRY(buf_ensure(ctx, &ctx->stage_res, &ctx->stage_res_cap, (size_t)out_cap + 64));
    JPK_HIP(hipMemcpyAsync(ctx->stage_in, in, (size_t)in_len, hipMemcpyHostToDevice, ctx->stream));
    JPK_HIP(hipEventRecord(ctx->ev_batch, ctx->stream));
    DecReq req;
    req.d_in = ctx->stage_in; req.in_len = in_len; req.d_out = ctx->stage_res; req.out_cap = out_cap; req.staged = ctx->ev_batch;
    bool leader;
    {
        std::unique_lock<std::mutex> lk(cb.mu);
        cb.pending.push_back(&req);
        arrived.submitted();
        leader = !cb.collecting;
        if (leader) cb.collecting = true;
        else cb.cv.wait(lk, [&] { return req.done; });
    }
    if (leader) {
        // gather: until nobody is on the way in; once the process has decoded concurrently before, a grace period on top
        const auto t0 = std::chrono::steady_clock::now();
        int grace;
        { std::lock_guard<std::mutex> lk(cb.mu); grace = cb.last_batch > 1 ? combine_grace_us() : 0; }
        for (;;) {
            const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (cb.arriving.load() == 0 && waited >= grace) break;
            if (waited > 20000) break;                              // a thread stuck in its staging must not hold the others
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        std::vector<DecReq *> batch;
        {
            std::lock_guard<std::mutex> lk(cb.mu);
            batch.swap(cb.pending);
            cb.collecting = false;                                   // the next arrival leads a batch of its own, beside this one
        }
        // the merged pass is bounded: requests beyond JPK_COMBINE_MAX_MIB (default 2048) of output capacity stay with their own
        // threads, so that the hidden combiner context's arena (~3 bytes per output byte) cannot grow without limit
        static const size_t max_bytes = [] { const char *e = getenv("JPK_COMBINE_MAX_MIB"); const long m = e ? atol(e) : 2048; return (size_t)(m < 1 ? 1 : m) << 20; }();
        std::vector<DecReq *> turned_away;
        {
            size_t bytes = 0;
            std::vector<DecReq *> kept;
            for (DecReq *r : batch) {
                if (!kept.empty() && bytes + (size_t)r->out_cap > max_bytes) { r->alone = true; turned_away.push_back(r); continue; }
                bytes += (size_t)r->out_cap;
                kept.push_back(r);
            }
            batch.swap(kept);
        }
        const int nb = (int)batch.size();
        int rc = JPK_OK;
        if (nb == 1 && batch[0] != &req) { batch[0]->alone = true; turned_away.push_back(batch[0]); batch.clear(); }
        else if (nb == 1) {
            rc = jpk_dev_ans_decode(ctx, req.d_in, req.in_len, req.d_out, req.out_cap, &req.out_len);       // the single-block path, as ever
            req.status = rc;
        } else {
            jpk_ctx *cc = nullptr;
            if (g_combiner_fail_next.load() > 0 && g_combiner_fail_next.fetch_sub(1) > 0) rc = JPK_E_ALLOC;     // (test hook)
            else rc = batch_ctx_acquire(ctx->device, &cc);
            uint64_t generation;
            { CtxPool &p = pool(); std::lock_guard<std::mutex> g(p.mu); generation = p.generation; }
            if (rc == JPK_OK) {
                std::vector<const uint8_t *> ins((size_t)nb);
                std::vector<uint8_t *> outs((size_t)nb);
                std::vector<int32_t> il((size_t)nb), oc((size_t)nb), ol((size_t)nb), st((size_t)nb);
                for (int b = 0; b < nb && rc == JPK_OK; b++) {
                    ins[b] = batch[b]->d_in; il[b] = batch[b]->in_len; outs[b] = batch[b]->d_out; oc[b] = batch[b]->out_cap;
                    if (hipStreamWaitEvent(cc->stream, batch[b]->staged, 0) != hipSuccess) rc = JPK_E_DEVICE;
                }
                if (rc == JPK_OK) rc = jpk_ans_decode_batch(cc, nb, ins.data(), il.data(), outs.data(), oc.data(), ol.data(), st.data(), 0);   // synchronises cc->stream
                if (rc != JPK_OK) (void)hipStreamSynchronize(cc->stream);      // nothing of the failed pass may still touch the requests' buffers
                for (int b = 0; b < nb && rc == JPK_OK; b++) { batch[b]->out_len = ol[b]; batch[b]->status = st[b]; }
                batch_ctx_release(ctx->device, cc, generation);
            }
            // The pass as a WHOLE failed (no combiner context, its arena for N blocks did not fit beside the callers' own arenas, a
            // stream error): that says nothing about any single request -- each would have gone through on the single-block path
            // with its own context.  Every thread retries its own request alone (per-block statuses of a pass that ran stand).
            if (rc != JPK_OK)
                for (int b = 0; b < nb; b++) { batch[b]->alone = true; batch[b]->status = JPK_OK; batch[b]->out_len = 0; }
        }
        {
            std::lock_guard<std::mutex> lk(cb.mu);
            cb.last_batch = nb;
            for (DecReq *r : batch) r->done = true;
            for (DecReq *r : turned_away) r->done = true;
        }
        cb.cv.notify_all();
    }
    if (req.alone) {
        req.status = jpk_dev_ans_decode(ctx, req.d_in, req.in_len, req.d_out, req.out_cap, &req.out_len);
    }
    if (req.status != JPK_OK) return req.status;
    if (req.out_len > 0) JPK_HIP(hipMemcpyAsync(out, ctx->stage_res, (size_t)req.out_len, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    *out_len = req.out_len;
    return JPK_OK;
}

namespace {
// LEB128 "with carry" (utils.cpp:70-90), host side; returns bytes consumed or -1
int leb_host(uint32_t *v, const uint8_t *b, int64_t avail)
{
    static const uint32_t C[4] = {127u, 16510u, 2113661u, 270549116u};
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
}  // namespace

extern "C" int jpk_ans_decoded_size(const uint8_t *in, int32_t in_len, int64_t *decoded_len, int32_t *chunks)
{
    if (!decoded_len || in_len < 0 || (in_len > 0 && !in)) return JPK_E_ARG;
    int64_t ip = 0, total = 0;
    int32_t nch = 0;
    while (ip < in_len) {
        int64_t fsum = 0;
        uint32_t v = 0, olen = 0;
        for (int s = 0; s < 259; s++) {                 // 256 frequencies, olen, clen, rlen (ans.cpp:272-302)
            const int n = leb_host(&v, in + ip, (int64_t)in_len - ip);
            if (n < 0) return JPK_E_CORRUPT;
            ip += n;
            if (s < 256) { if (v > (uint32_t)JPK_ANS_CHUNK) return JPK_E_CORRUPT; fsum += v; }
            else if (s == 256) { if (v > (uint32_t)JPK_ANS_CHUNK || (int64_t)v != fsum) return JPK_E_CORRUPT; total += v; olen = v; }
            else if (s == 257) { if (v < 16 || (int64_t)v > (int64_t)in_len - ip) return JPK_E_CORRUPT; fsum = v; }   // clen, payload follows rlen
            else { if (v > olen) return JPK_E_CORRUPT; }     // RLE0 never has more symbols than bytes ("rle mismatch!", rle.cpp:73)
        }
        if (fsum > (int64_t)in_len - ip) return JPK_E_CORRUPT;
        ip += fsum;                                      // skip the payload
        nch++;
    }
    *decoded_len = total;
    if (chunks) *chunks = nch;
    return JPK_OK;
}

extern "C" int jpk_block_compress(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    return staged(jpk_dev_block_compress, in, in_len, out, out_cap, out_len, false);
}

extern "C" int jpk_block_decompress(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    return staged(jpk_dev_block_decompress, in, in_len, out, out_cap, out_len, false);
}

extern "C" int jpk_rank_encode(uint8_t *t, int32_t *freq256, int32_t len)
{
    if (!freq256 || len < 0 || (len > 0 && !t)) return JPK_E_ARG;
    jpk_ctx *ctx;
    JPK_TRY(tls_ctx(&ctx));
    JPK_HIP(hipSetDevice(ctx->device));
    JPK_TRY(buf_ensure(ctx, &ctx->stage_in, &ctx->stage_in_cap, (size_t)len + 2048));
    uint8_t *d_t = ctx->stage_in + 1024;
    int32_t *d_f = (int32_t *)ctx->stage_in;
    if (len) JPK_HIP(hipMemcpyAsync(d_t, t, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    JPK_TRY(jpk_rank_encode_device(ctx, d_t, d_f, len));
    if (len) JPK_HIP(hipMemcpyAsync(t, d_t, (size_t)len, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipMemcpyAsync(freq256, d_f, 1024, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    return JPK_OK;
}

extern "C" int jpk_rank_decode(uint8_t *ranks, const int32_t *freq256, int32_t len)
{
    if (!freq256 || len < 0 || (len > 0 && !ranks)) return JPK_E_ARG;
    jpk_ctx *ctx;
    JPK_TRY(tls_ctx(&ctx));
    JPK_HIP(hipSetDevice(ctx->device));
    JPK_TRY(buf_ensure(ctx, &ctx->stage_in, &ctx->stage_in_cap, (size_t)len + 2048));
    uint8_t *d_t = ctx->stage_in + 1024;
    int32_t *d_f = (int32_t *)ctx->stage_in;
    if (len) JPK_HIP(hipMemcpyAsync(d_t, ranks, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    JPK_HIP(hipMemcpyAsync(d_f, freq256, 1024, hipMemcpyHostToDevice, ctx->stream));
    JPK_TRY(jpk_rank_decode_device(ctx, d_t, d_f, len));
    if (len) JPK_HIP(hipMemcpyAsync(ranks, d_t, (size_t)len, hipMemcpyDeviceToHost, ctx->stream));
    JPK_HIP(hipStreamSynchronize(ctx->stream));
    return JPK_OK;
}

extern "C" int jpk_checksum(const uint8_t *in, int32_t in_len, uint32_t *crc)
{
    if (!crc || in_len < 0 || (in_len > 0 && !in)) return JPK_E_ARG;
    jpk_ctx *ctx;
    JPK_TRY(tls_ctx(&ctx));
    JPK_HIP(hipSetDevice(ctx->device));
    JPK_TRY(buf_ensure(ctx, &ctx->stage_in, &ctx->stage_in_cap, (size_t)in_len + 64));
    if (in_len) JPK_HIP(hipMemcpyAsync(ctx->stage_in, in, (size_t)in_len, hipMemcpyHostToDevice, ctx->stream));
    return jpk_dev_checksum(ctx, ctx->stage_in, in_len, crc);
}

namespace {
thread_local int32_t tls_block_size = 0;
thread_local int32_t tls_consumed = 0;
int jam_write_tramp(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    return jpk_dev_jam_block_write(ctx, d_in, in_len, tls_block_size, d_out, out_cap, out_len);
}
int jam_read_tramp(jpk_ctx *ctx, const uint8_t *d_in, int32_t in_len, uint8_t *d_out, int32_t out_cap, int32_t *out_len)
{
    return jpk_dev_jam_block_read(ctx, d_in, in_len, d_out, out_cap, out_len, &tls_consumed);
}
}  // namespace

extern "C" int jpk_jam_block_write(const uint8_t *in, int32_t in_len, int32_t block_size, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    tls_block_size = block_size;
    return staged(jam_write_tramp, in, in_len, out, out_cap, out_len, false);
}

extern "C" int jpk_jam_block_read(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, int32_t *consumed)
{
    if (in_len >= JPK_JAM_HEADER_BYTES && in) {      // stage only this frame, not the rest of the stream
        int32_t csize;
        memcpy(&csize, in + 7, 4);
        if (csize < 0 || csize > JPK_MAX_BLOCKSIZE || (int64_t)csize + JPK_JAM_HEADER_BYTES > in_len) return JPK_E_CORRUPT;
        in_len = csize + JPK_JAM_HEADER_BYTES;
    }
    tls_consumed = 0;
    int rc = staged(jam_read_tramp, in, in_len, out, out_cap, out_len, false);
    if (rc == JPK_OK && consumed) *consumed = tls_consumed;
    return rc;
}

// One frame of the stock CLI: header (jampack.cpp:140-164), GPU stages, host pre-stage decoders in the order of
// Jampack::Decomp() (jampack.cpp:47-57), crc check (jampack.cpp:58-59).
extern "C" int jpk_jam_cli_block_read(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len, int32_t *consumed)
{
    if (!in || !out || !out_len || in_len < 0 || out_cap < 0) return JPK_E_ARG;
    if (in_len < JPK_JAM_HEADER_BYTES) return JPK_E_CORRUPT;
    uint32_t crc;
    int32_t csize, block_size;
    memcpy(&crc, in + 3, 4);
    memcpy(&csize, in + 7, 4);
    memcpy(&block_size, in + 11, 4);
    if (memcmp(in, "JAM", 3) != 0 || !jam_block_size_ok(block_size) || csize < 0 || csize > JPK_MAX_BLOCKSIZE) return JPK_E_CORRUPT;
    if ((int64_t)csize + JPK_JAM_HEADER_BYTES > in_len) return JPK_E_CORRUPT;
    const int64_t cap64 = (int64_t)((double)block_size * 1.05) + 4096;             // the reference's stage buffers, jampack.cpp:156
    if (cap64 > 0x7fffffff) return JPK_E_ARG;
    const int32_t cap = (int32_t)cap64;
    // the two 1.05 x BlockSize stage buffers of the reference (jampack.cpp:156-159), kept per thread across frames
    static thread_local std::vector<uint8_t> a, b;
    try { if (a.size() < (size_t)cap) a.resize((size_t)cap); if (b.size() < (size_t)cap) b.resize((size_t)cap); } catch (...) { return JPK_E_ALLOC; }
    int32_t n = 0;
    JPK_TRY(jpk_block_decompress(in + JPK_JAM_HEADER_BYTES, csize, a.data(), cap, &n));     // Ans::Decode + InverseBwt
    int32_t m = 0;
    JPK_TRY(jpk_lz77_decompress(a.data(), n, b.data(), cap, &m));                            // Lz->Decompress
    JPK_TRY(jpk_lpx_decode(b.data(), m, a.data()));                                          // LocalModel->Decode
    JPK_TRY(jpk_filters_decode(a.data(), m, b.data(), cap, &n));                             // Filter->Decode
    JPK_TRY(jpk_lz77_decompress(b.data(), n, out, out_cap, &m));                             // Lz->Decompress
    if (jpk_checksum_host(out, m) != crc) return JPK_E_CORRUPT;                              // "Detected corrupt block!"
    *out_len = m;
    if (consumed) *consumed = csize + JPK_JAM_HEADER_BYTES;
    return JPK_OK;
}
