// prestage.cpp -- host-side decoders of the three stages the stock CLI runs in front of the BWT
// (Jampack::Decomp, jampack.cpp:47-60: Lz77::Decompress, Lpx::Decode, Filters::Decode, Lz77::Decompress), so that
// frames written by an unmodified `jampack c` decode end to end: rANS decode + inverse BWT on the GPU, these on the
// host where SURVEY.md section 8f (row 4) puts them -- byte-serial state machines with no data parallelism.
// Only the decoder side exists here; the encoders carry float heuristics and stay with the reference.
//
// Unlike the reference (which trusts its input outside NDEBUG builds, lz77.cpp:697-701) every read and write is
// bounds checked and a bad stream gives JPK_E_CORRUPT / JPK_E_CAPACITY.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/jampack_abi.h"

namespace {

// LEB128 "with carry" (Utils::DecodeLeb128, utils.cpp:73-90): big-endian 7-bit groups, the last byte has bit 7 set,
// a d-byte prefix adds the class offset of its length.  Returns bytes consumed or -1.
int leb_read(const uint8_t *b, int64_t avail, int32_t *v)
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
    *v = (int32_t)x;
    return d + 1;
}

}  // namespace

// Lz77::Decompress (lz77.cpp:678-714).  Token (lz77.cpp:75-98): one byte = match length class (5 bits) | literal
// count class (3 bits), then the offset, then the extensions of a saturated class; match lengths are stored minus 4.
// Offset 0 ends the LZ code: the rest of the input is copied through.
extern "C" int jpk_lz77_decompress(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    if (!out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !in) || (out_cap > 0 && !out)) return JPK_E_ARG;
    int64_t pos = 0, op = 0;
    while (pos < in_len) {
        const uint32_t token = in[pos++];
        int32_t off = 0;
        int64_t len = (int64_t)(token >> 3), lit = (int64_t)(token & 7u);   // 64-bit: the extensions are attacker-controlled int32
        int n = leb_read(in + pos, in_len - pos, &off);
        if (n < 0) return JPK_E_CORRUPT;
        pos += n;
        if (len == 31) {
            int32_t e = 0;
            n = leb_read(in + pos, in_len - pos, &e);
            if (n < 0 || e < 0) return JPK_E_CORRUPT;
            pos += n;
            len += e;
        }
        len += 4;                                                      // MIN_MATCH, lz77.hpp:33
        if (lit == 7) {
            int32_t e = 0;
            n = leb_read(in + pos, in_len - pos, &e);
            if (n < 0 || e < 0) return JPK_E_CORRUPT;
            pos += n;
            lit += e;
        }
        if (off == 0) {                                                // end marker: raw remainder
            const int64_t rest = in_len - pos;
            if (op + rest > out_cap) return JPK_E_CAPACITY;
            memcpy(out + op, in + pos, (size_t)rest);
            op += rest;
            break;
        }
        if (off < 0 || lit > in_len - pos) return JPK_E_CORRUPT;
        if (lit + len > (int64_t)out_cap - op) return JPK_E_CAPACITY;
        memcpy(out + op, in + pos, (size_t)lit);
        op += lit;
        pos += lit;
        if (off > op) return JPK_E_CORRUPT;
        const uint8_t *src = out + op - off;                           // may overlap the destination: byte order matters
        for (int64_t k = 0; k < len; k++) out[op + k] = src[k];
        op += len;
    }
    *out_len = (int32_t)op;
    return JPK_OK;
}

namespace {

// Lpx: localized prefix model (lpx.hpp:12-24, lpx.cpp:11-52).  Three tables (context orders 1..3) of 256 records
// keyed by the leading prefix byte; the decoder mirrors the encoder's table walk exactly.
struct PrefixRecord {
    uint32_t cxt, pos, hits, miss;
    int32_t threshold;
};
constexpr int LPX_MAX_THRESHOLD = 128, LPX_MIN_THRESHOLD = 4;
constexpr uint32_t LPX_MAX_RECORD = 64u << 10;

struct LpxState {
    PrefixRecord table[3][256];
    uint32_t cxt = 0;
    int order = 3;
    LpxState()
    {
        memset(table, 0, sizeof table);
        for (auto &t : table)
            for (auto &r : t) r.threshold = LPX_MAX_THRESHOLD >> 1;
    }
    // lpx.cpp:11-52.  Note the reference re-indexes the table with the *updated* order for the threshold adjustments.
    void update(uint32_t pos)
    {
        const uint32_t lp = (cxt >> (order * 8)) & 0xffu;
        const uint32_t ls = cxt & ((1u << (order * 8)) - 1u);
        PrefixRecord *r = &table[order - 1][lp];
        const int32_t distance = (int32_t)(pos - r->pos);
        const int32_t lower = LPX_MIN_THRESHOLD;
        int32_t upper;
        if (r->hits < (uint32_t)LPX_MAX_THRESHOLD) upper = distance > LPX_MIN_THRESHOLD ? distance : LPX_MIN_THRESHOLD;
        else { const int32_t a = distance >> order, b = LPX_MAX_THRESHOLD >> order; upper = a < b ? a : b; }
        const int32_t bound = (distance <= lower) ? lower : (distance > upper ? upper : distance);
        if (pos <= (uint32_t)order) return;
        if (r->cxt == ls) {
            r->pos = pos - (uint32_t)order;
            r->hits++;
            r->miss = 0;
            if (r->hits > (uint32_t)((r->threshold << order) << 3) && order > 1 && order <= 3) order--;
            r = &table[order - 1][lp];
            if (r->hits > (uint32_t)(r->threshold << 1) && r->miss == 0) r->threshold += (bound - r->threshold) >> order;
        } else {
            r->hits >>= 2;
            r->miss++;
            r->cxt = ls;
            if (r->miss > (uint32_t)(r->threshold * r->threshold * order) && order >= 1 && order < 3) order++;
            r = &table[order - 1][lp];
            if (r->miss > (uint32_t)r->threshold) r->threshold += (LPX_MAX_THRESHOLD - r->threshold) >> (4 - order);
        }
    }
};

// Lpx::DecodeBlock (lpx.cpp:101-144): inside a predicted stretch the stream holds prediction XOR byte
void lpx_decode_part(const uint8_t *in, uint8_t *out, int64_t len)
{
    LpxState *st = new LpxState();
    for (int64_t i = 0; i < len;) {
        const PrefixRecord &r = st->table[st->order - 1][st->cxt & 0xffu];
        const uint32_t dist = (uint32_t)i - r.pos;
        if (r.hits > (uint32_t)r.threshold && dist < LPX_MAX_RECORD && dist <= (uint32_t)i) {
            uint8_t err;
            do {
                err = in[i];
                out[i] = out[i - dist] ^ in[i];
                st->update((uint32_t)i);
                st->cxt = (st->cxt << 8) | out[i];
                i++;
            } while (err == 0 && i < len);
        } else {
            out[i] = in[i];
            st->update((uint32_t)i);
            st->cxt = (st->cxt << 8) | out[i];
            i++;
        }
    }
    delete st;
}

}  // namespace

// Lpx::Decode (lpx.cpp:158-169): the block is cut into parts of len / 4 bytes (a fifth, shorter one when len is not a
// multiple of 4), each decoded with a fresh model.  The reference loops forever for 0 < len < 4 (part size 0); no
// encoder output can be that short, so such inputs are passed through as one part.
extern "C" int jpk_lpx_decode(const uint8_t *in, int32_t len, uint8_t *out)
{
    if (len < 0 || (len > 0 && (!in || !out))) return JPK_E_ARG;
    const int64_t part = len / 4;
    if (part == 0) { if (len) lpx_decode_part(in, out, len); return JPK_OK; }
    for (int64_t i = 0; i < len; i += part) lpx_decode_part(in + i, out + i, (i + part < len) ? part : len - i);
    return JPK_OK;
}

// Filters::Decode (filters.cpp:442-490): per 64 KiB block two header bytes (filter type, channel width), width 0 =
// raw.  Types: 0 delta and 1 adaptive linear prediction on de-interleaved channels, 2 in-place delta per channel.
extern "C" int jpk_filters_decode(const uint8_t *in, int32_t in_len, uint8_t *out, int32_t out_cap, int32_t *out_len)
{
    if (!out_len || in_len < 0 || out_cap < 0 || (in_len > 0 && !in) || (out_cap > 0 && !out)) return JPK_E_ARG;
    constexpr int64_t FBS = 64 << 10;
    std::vector<uint8_t> dbuf((size_t)FBS);
    int64_t i = 0, op = 0;
    while (i < in_len) {
        if (i + 2 > in_len) return JPK_E_CORRUPT;
        const int type = in[i], width = in[i + 1];
        i += 2;
        if (type >= 3 || width > 32) return JPK_E_CORRUPT;            // "unsupported configuration", filters.cpp:455
        const int64_t len = (i + FBS < in_len) ? FBS : in_len - i;
        if (op + len > out_cap) return JPK_E_CAPACITY;
        const uint8_t *src = in + i;
        uint8_t *dst = out + op;
        if (width == 0) {
            memcpy(dst, src, (size_t)len);
        } else if (type == 2) {                                         // InlineUndelta, filters.cpp: running sum per channel in place
            uint8_t prev[32] = {0};
            int64_t k = len % width;
            memcpy(dst, src, (size_t)k);
            for (; k < len; k += width)
                for (int j = 0; j < width; j++) { dst[k + j] = (uint8_t)(src[k + j] + prev[j]); prev[j] = dst[k + j]; }
        } else {
            if (type == 0) {                                            // DeltaDecode: running sum over the whole block
                uint8_t prev = 0;
                for (int64_t k = 0; k < len; k++) { prev = (uint8_t)(src[k] + prev); dbuf[(size_t)k] = prev; }
            } else {                                                    // LpcDecode: x = w + 2 p1 - p2 - err, w += (err - w) >> 6
                int32_t weight = 0;
                uint8_t p1 = 0, p2 = 0;
                for (int64_t k = 0; k < len; k++) {
                    const uint8_t err = src[k];
                    const uint8_t cur = (uint8_t)(weight + (((int32_t)p1 - (int32_t)p2) + (int32_t)p1) - (int32_t)err);
                    dbuf[(size_t)k] = cur;
                    weight += ((int32_t)err - weight) >> 6;
                    p2 = p1;
                    p1 = cur;
                }
            }
            int64_t p = 0;                                              // Unreorder: channel c holds bytes c, c + width, ...
            for (int c = 0; c < width; c++)
                for (int64_t j = c; j < len; j += width) dst[j] = dbuf[(size_t)p++];
        }
        op += len;
        i += len;
    }
    *out_len = (int32_t)op;
    return JPK_OK;
}

// Checksum::IntegrityCheck on the host (checksum.cpp:12-36), for buffers that are already there
extern "C" uint32_t jpk_checksum_host(const uint8_t *p, int32_t size)
{
    const uint32_t prime = 0x9E3779B1u;
    uint32_t S[4] = {3u, 0u, 0u, 0u};
    uint32_t j = 0;
    const uint32_t n = size > 0 ? (uint32_t)size : 0u;
    while ((uint64_t)j + 16 < n) {
        for (int k = 0; k < 4; k++) {
            const uint8_t *q = p + j + 4 * k;
            const uint32_t w = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
            S[k] ^= (w + (1u << (S[k] & 7))) * prime;
        }
        j += 16;
    }
    for (; j < n; j++) S[0] ^= ((uint32_t)p[j] + (1u << (S[0] & 7))) * prime;
    return S[0] ^ S[1] ^ S[2] ^ S[3];
}
