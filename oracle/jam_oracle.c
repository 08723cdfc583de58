/*
 * jam_oracle.c -- CPU restatement of Jampack's block hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * This file is the parity oracle for the HIP kernels in jampack_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the shipped library
 * (libjampack_amd.so) never links, loads or calls anything in oracle/.
 *
 * It is a from-scratch restatement (plain C99) of the reference algorithm; every function cites
 * the reference file:line it follows.  Parity status: PINNED -- tests/test_oracle_vs_ref.py diffs
 * every function here byte-for-byte against the reference itself compiled from /root/reference
 * (oracle/_ref/libjamref.so, built by oracle/Makefile) and tests/test_oracle_golden.py checks it
 * against the committed tests/golden/ vectors that were generated from that reference build.
 *
 * Path restated:
 *   forward BWT      bwt.cpp:22-65      (suffix array: any correct SA gives identical bytes; the
 *                                        reference calls divsufsort, divsufsort.cpp:1721)
 *   inverse BWT      bwt.cpp:72-282
 *   sorted-rank      rank.cpp:15-151
 *   RLE0             rle.cpp:22-74
 *   models           model.cpp:60-235, tables.hpp:10-30
 *   rANS             rans_byte.hpp:56-154
 *   chunk driver     ans.cpp:113-302, LEB128 utils.cpp:22-90
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BWT_UNITS 120            /* format.hpp:26 */
#define TRAILER   (BWT_UNITS * 4)
#define CHUNK     (1 << 20)      /* ans.hpp:21 StackSize */
#define PROB_BITS 16
#define PROB_SCALE 65536
#define RANS_L    (1u << 23)     /* rans_byte.hpp:50 */

enum { ORC_OK = 0, ORC_E_ALLOC = -1, ORC_E_CORRUPT = -2, ORC_E_CAPACITY = -3 };

/* ------------------------------------------------------------------------------------------ */
/* Suffix array: prefix doubling with two stable counting sorts per round (O(n log n)).        */
/* Plain suffix order: a proper prefix sorts first (what divsufsort produces).                 */
/* ------------------------------------------------------------------------------------------ */
int orc_suffix_array(const uint8_t *T, int32_t n, int32_t *SA)
{
    if (n <= 0) return ORC_OK;
    int32_t *rk = (int32_t *)malloc((size_t)n * 4);
    int32_t *nr = (int32_t *)malloc((size_t)n * 4);
    int32_t *tmp = (int32_t *)malloc((size_t)n * 4);
    int32_t *cnt = (int32_t *)malloc(((size_t)n + 2) * 4);
    if (!rk || !nr || !tmp || !cnt) { free(rk); free(nr); free(tmp); free(cnt); return ORC_E_ALLOC; }

    /* round 0: order by first byte, rank = start of the byte's bucket */
    int32_t c256[257];
    memset(c256, 0, sizeof c256);
    for (int32_t i = 0; i < n; i++) c256[T[i] + 1]++;
    for (int k = 0; k < 256; k++) c256[k + 1] += c256[k];
    for (int32_t i = 0; i < n; i++) rk[i] = c256[T[i]];
    {
        int32_t pos[256];
        for (int k = 0; k < 256; k++) pos[k] = c256[k];
        for (int32_t i = 0; i < n; i++) SA[pos[T[i]]++] = i;
    }

    for (int64_t h = 1;; h <<= 1) {
        /* stable sort of the current order by second key (rank of suffix i+h, 0 if past the end) */
        memset(cnt, 0, ((size_t)n + 2) * 4);
        for (int32_t i = 0; i < n; i++) {
            int32_t k2 = (i + h < n) ? rk[i + h] + 1 : 0;
            cnt[k2 + 1]++;
        }
        for (int32_t k = 0; k <= n; k++) cnt[k + 1] += cnt[k];
        for (int32_t j = 0; j < n; j++) {
            int32_t i = SA[j];
            int32_t k2 = (i + h < n) ? rk[i + h] + 1 : 0;
            tmp[cnt[k2]++] = i;
        }
        /* stable sort by first key */
        memset(cnt, 0, ((size_t)n + 2) * 4);
        for (int32_t i = 0; i < n; i++) cnt[rk[i] + 1]++;
        for (int32_t k = 0; k < n; k++) cnt[k + 1] += cnt[k];
        for (int32_t j = 0; j < n; j++) {
            int32_t i = tmp[j];
            SA[cnt[rk[i]]++] = i;
        }
        /* re-rank */
        int32_t heads = 1;
        nr[SA[0]] = 0;
        for (int32_t j = 1; j < n; j++) {
            int32_t a = SA[j - 1], b = SA[j];
            int32_t ka = (a + h < n) ? rk[a + h] + 1 : 0;
            int32_t kb = (b + h < n) ? rk[b + h] + 1 : 0;
            if (rk[a] == rk[b] && ka == kb) nr[b] = nr[a];
            else { nr[b] = j; heads++; }
        }
        int32_t *sw = rk; rk = nr; nr = sw;
        if (heads == n || h >= n) break;
    }
    free(rk); free(nr); free(tmp); free(cnt);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Forward BWT -- bwt.cpp:22-65.  out must hold len + 480 bytes.  When len < 120 the 480       */
/* trailer bytes are left untouched (bwt.cpp:35 `if(nlen > 0)`).                               */
/* ------------------------------------------------------------------------------------------ */
int orc_bwt_forward(const uint8_t *T, int32_t len, uint8_t *out, int32_t *out_len)
{
    *out_len = len + TRAILER;
    int32_t rem = len % BWT_UNITS, nlen = len - rem;
    for (int32_t i = 0; i < rem; i++) out[nlen + i] = T[nlen + i];
    if (nlen <= 0) return ORC_OK;

    int32_t *SA = (int32_t *)malloc((size_t)nlen * 4);
    if (!SA) return ORC_E_ALLOC;
    int rc = orc_suffix_array(T, nlen, SA);
    if (rc) { free(SA); return rc; }

    int32_t step = nlen / BWT_UNITS;
    int32_t ind[BWT_UNITS];
    memset(ind, 0, sizeof ind);
    for (int32_t i = 0; i < nlen; i++)
        if (SA[i] % step == 0) ind[SA[i] / step] = i;      /* bwt.cpp:46-48 */
    int32_t idx = ind[0];
    out[0] = T[nlen - 1];
    for (int32_t i = 0; i < idx; i++) out[i + 1] = T[SA[i] - 1];
    for (int32_t i = idx + 1; i < nlen; i++) out[i] = T[SA[i] - 1];
    for (int k = 0; k < BWT_UNITS; k++) {
        int32_t v = ind[k] + 1;                              /* bwt.cpp:57-61 */
        memcpy(out + len + 4 * k, &v, 4);
    }
    free(SA);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Inverse BWT -- bwt.cpp:72-282 with a single chain (the chain count never changes T).        */
/* ------------------------------------------------------------------------------------------ */
int orc_bwt_inverse(const uint8_t *B, int32_t len_with_trailer, uint8_t *T, int32_t *out_len)
{
    int32_t len = len_with_trailer - TRAILER;
    if (len < 0) return ORC_E_CORRUPT;
    *out_len = len;
    int32_t rem = len % BWT_UNITS, nlen = len - rem;
    for (int32_t i = 0; i < rem; i++) T[nlen + i] = B[nlen + i];
    if (nlen <= 0) return ORC_OK;

    int32_t idx;
    memcpy(&idx, B + len, 4);
    if (idx < 1 || idx > nlen) return ORC_E_CORRUPT;

    int32_t count[257];
    memset(count, 0, sizeof count);
    for (int32_t i = 0; i < nlen; i++) count[B[i] + 1]++;
    for (int k = 1; k < 257; k++) count[k] += count[k - 1];
    int32_t *Map = (int32_t *)malloc((size_t)nlen * 4);
    if (!Map) return ORC_E_ALLOC;
    for (int32_t i = 0; i < idx; i++) Map[count[B[i]]++] = i;          /* bwt.cpp:171-174 */
    for (int32_t i = idx; i < nlen; i++) Map[count[B[i]]++] = i + 1;

    int32_t p = idx;
    for (int32_t i = 0; i < nlen; i++) {                                /* bwt.cpp:261-276 */
        if (p < 1 || p > nlen) { free(Map); return ORC_E_CORRUPT; }
        p = Map[p - 1];
        T[i] = B[p - (p >= idx)];
    }
    free(Map);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Sorted-rank coding -- rank.cpp:15-151                                                        */
/* ------------------------------------------------------------------------------------------ */
/* rank.cpp:15-39: symbols by descending frequency, ties -> smaller byte value first. */
int orc_sorted_map(const int32_t *freq, uint8_t *map)
{
    int32_t f[256];
    int n = 0;
    memcpy(f, freq, sizeof f);
    for (;;) {
        int best = -1, mx = 0;
        for (int s = 0; s < 256; s++)
            if (f[s] > mx) { mx = f[s]; best = s; }
        if (best < 0) break;
        map[n++] = (uint8_t)best;
        f[best] = 0;
    }
    return n;
}

/* rank.cpp:45-90 -- in place; returns Freq[256]. */
int orc_rank_encode(uint8_t *T, int32_t *freq, int32_t len)
{
    uint8_t *out = (uint8_t *)malloc(len > 0 ? (size_t)len : 1);
    if (!out) return ORC_E_ALLOC;
    memset(freq, 0, 256 * 4);
    uint8_t list[256];           /* MTF list: list[r] = symbol at rank r */
    int nseen = 0;
    for (int32_t i = 0; i < len; i++) {
        if (freq[T[i]]++ == 0) list[nseen++] = T[i];       /* first-appearance order */
    }
    uint8_t smap[256];
    int32_t bucket[256];
    int ns = orc_sorted_map(freq, smap);
    int32_t pos = 0;
    for (int k = 0; k < ns; k++) { bucket[smap[k]] = pos; pos += freq[smap[k]]; }

    for (int32_t i = 0; i < len; i++) {
        uint8_t s = T[i];
        int r = 0;
        while (list[r] != s) r++;
        out[bucket[s]++] = (uint8_t)r;
        for (; r > 0; r--) list[r] = list[r - 1];
        list[0] = s;
    }
    if (len > 0) memcpy(T, out, (size_t)len);
    free(out);
    return ORC_OK;
}

/* rank.cpp:96-151 -- in place. */
int orc_rank_decode(uint8_t *R, const int32_t *freq, int32_t len)
{
    int64_t total = 0;
    int uniq = 0;
    for (int s = 0; s < 256; s++) {
        if (freq[s] < 0) return ORC_E_CORRUPT;
        total += freq[s];
        if (freq[s] > 0) uniq++;
    }
    if (total != len) return ORC_E_CORRUPT;                /* rank.cpp:104-108 */
    if (len == 0) return ORC_OK;
    uint8_t *T = (uint8_t *)malloc((size_t)len);
    if (!T) return ORC_E_ALLOC;

    uint8_t smap[256], list[256];
    int32_t bpos[256], bend[256];
    memset(list, 0, sizeof list);
    orc_sorted_map(freq, smap);
    int32_t pos = 0;
    for (int k = 0; k < uniq; k++) {
        uint8_t s = smap[k];
        list[R[pos]] = s;               /* first stored rank of s = its first-appearance index */
        bpos[s] = pos + 1;
        pos += freq[s];
        bend[s] = pos;
    }
    uint8_t sym = list[0];
    for (int32_t i = 0; i < len; i++) {
        T[i] = sym;
        if (bpos[sym] < bend[sym]) {
            int r = R[bpos[sym]++];
            if (r > 0) {
                for (int k = 0; k < r; k++) list[k] = list[k + 1];
                list[r] = sym;
                sym = list[0];
            }
        } else if (uniq > 0) {
            uniq--;                                         /* rank.cpp:140-147 */
            int k = 0;
            do { list[k] = list[k + 1]; } while (++k < uniq);
            sym = list[0];
        }
    }
    memcpy(R, T, (size_t)len);
    free(T);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* RLE0 -- rle.cpp:22-74                                                                        */
/* ------------------------------------------------------------------------------------------ */
int32_t orc_rle_encode(const uint8_t *in, uint16_t *out, int32_t len)
{
    int32_t o = 0;
    for (int32_t i = 0; i < len;) {
        if (in[i] == 0) {
            int32_t run = 1;
            while (i + run < len && in[i + run] == 0) run++;
            i += run;
            uint32_t L = (uint32_t)run + 1;
            int msb = 31;
            while (!(L >> msb)) msb--;
            while (msb--) out[o++] = (L >> msb) & 1;
        } else {
            out[o++] = (uint16_t)(in[i++] + 1);
        }
    }
    return o;
}

/* returns decoded length or ORC_E_CORRUPT if it does not equal real_len (rle.cpp:72) */
int32_t orc_rle_decode(const uint16_t *in, uint8_t *out, int32_t rlen, int32_t real_len)
{
    int32_t o = 0;
    for (int32_t i = 0; i < rlen;) {
        if (in[i] > 1) {
            if (o >= real_len) return ORC_E_CORRUPT;
            out[o++] = (uint8_t)(in[i++] - 1);
        } else {
            uint32_t v = 1;
            while (i < rlen && in[i] <= 1) v = (v << 1) | in[i++];
            v -= 1;
            if ((int64_t)o + v > real_len) return ORC_E_CORRUPT;
            memset(out + o, 0, v);
            o += (int32_t)v;
        }
    }
    return o == real_len ? o : ORC_E_CORRUPT;
}

/* ------------------------------------------------------------------------------------------ */
/* Symbol split -- tables.hpp:10-30                                                             */
/* ------------------------------------------------------------------------------------------ */
static const int EXPO[9] = {0, 2, 4, 8, 16, 32, 64, 128, 257};
static inline int sym_class(int s) /* tables.hpp Log[] */
{
    int e = 0;
    while (e < 7 && s >= EXPO[e + 1]) e++;
    return e;
}

/* ------------------------------------------------------------------------------------------ */
/* Models -- model.cpp                                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int A; int32_t cdf[9]; } Adaptive;          /* alphabets 8, 2, 2 */
typedef struct { int A; int32_t cdf[130]; int32_t f[129]; int seen, expn; } Quasi; /* 4..129 */

static void uniform_cdf(int A, int32_t *cdf) /* model.cpp:85-95, 221-231 */
{
    int32_t scale = PROB_SCALE / A;
    cdf[0] = 0;
    for (int i = 0; i < A; i++) cdf[i + 1] = cdf[i] + scale + (i == 0 ? PROB_SCALE - scale * A : 0);
}
static void ad_reset(Adaptive *m, int A) { m->A = A; uniform_cdf(A, m->cdf); }
static void ad_update(Adaptive *m, int sym) /* model.cpp:60-77, mix row closed form :99-112 */
{
    for (int i = 1; i < m->A; i++) {
        int32_t mix = (i <= sym) ? i : i + PROB_SCALE - m->A;
        m->cdf[i] += (mix - m->cdf[i]) >> 5; /* arithmetic shift of a possibly negative value */
    }
}
static void qs_reset(Quasi *m, int A) /* model.cpp:209-235 */
{
    m->A = A; uniform_cdf(A, m->cdf);
    memset(m->f, 0, sizeof m->f);
    m->seen = 0; m->expn = 8;
}
static void qs_update(Quasi *m, int sym) /* model.cpp:160-204 */
{
    m->f[sym] += 16;
    if (++m->seen > m->expn) {
        int32_t total = 0;
        int lg = 0;
        for (int i = 0; i < m->A; i++) total += m->f[i];
        while ((total >> lg) + m->A > PROB_SCALE) lg++;
        uint32_t t2 = 0;
        for (int i = 0; i < m->A; i++) { m->f[i] = (m->f[i] >> lg) + 1; t2 += (uint32_t)m->f[i]; }
        uint32_t t3 = 0;
        for (int i = 0; i < m->A; i++) {
            m->f[i] = (int32_t)(((uint32_t)PROB_SCALE * (uint32_t)m->f[i]) / t2); /* unsigned 32-bit, model.cpp:183 */
            t3 += (uint32_t)m->f[i];
        }
        m->f[0] += (int32_t)(PROB_SCALE - t3);
        m->cdf[0] = 0;
        for (int i = 0; i < m->A; i++) m->cdf[i + 1] = m->cdf[i] + m->f[i];
        memset(m->f, 0, sizeof m->f);
        m->seen = 0;
        m->expn = (m->expn < 65536) ? m->expn << 1 : 65536;
    }
}

typedef struct { Adaptive ex; Adaptive mp[2]; Quasi ms[6]; } Models;
static void models_reset(Models *M) /* ans.cpp:136-140 */
{
    ad_reset(&M->ex, 8);
    for (int c = 0; c < 2; c++) ad_reset(&M->mp[c], EXPO[c + 1] - EXPO[c]);
    for (int c = 0; c < 6; c++) qs_reset(&M->ms[c], EXPO[c + 3] - EXPO[c + 2]);
}

/* Model pass of one chunk: rle symbols -> 2*rlen packed pairs (low | freq<<16) -- ans.cpp:152-187.
 * freq <= 65535 always (every alphabet has >= 2 symbols of freq >= 1). */
int orc_model_pairs(const uint16_t *rle, int32_t rlen, uint32_t *pairs)
{
    Models M;
    models_reset(&M);
    for (int32_t i = 0; i < rlen; i++) {
        int s = rle[i];
        if (s > 256) return ORC_E_CORRUPT;
        int e = sym_class(s), m = s - EXPO[e];
        uint32_t lo = (uint32_t)M.ex.cdf[e], fr = (uint32_t)(M.ex.cdf[e + 1] - M.ex.cdf[e]);
        if (fr == 0 || fr > 65535) return ORC_E_CORRUPT;
        pairs[2 * i] = lo | (fr << 16);
        ad_update(&M.ex, e);
        if (e < 2) {
            Adaptive *a = &M.mp[e];
            lo = (uint32_t)a->cdf[m]; fr = (uint32_t)(a->cdf[m + 1] - a->cdf[m]);
            ad_update(a, m);
        } else {
            Quasi *q = &M.ms[e - 2];
            lo = (uint32_t)q->cdf[m]; fr = (uint32_t)(q->cdf[m + 1] - q->cdf[m]);
            qs_update(q, m);
        }
        if (fr == 0 || fr > 65535) return ORC_E_CORRUPT;
        pairs[2 * i + 1] = lo | (fr << 16);
    }
    return ORC_OK;
}

/* 4-way interleaved rANS over packed pairs, reverse order -- ans.cpp:189-208, rans_byte.hpp:62-110.
 * Writes the stream forward into out (capacity cap); returns its size or a negative error. */
int32_t orc_rans_encode_pairs(const uint32_t *pairs, int32_t npairs, uint8_t *out, int32_t cap)
{
    size_t tcap = (size_t)npairs * 2 + 16;
    uint8_t *tmp = (uint8_t *)malloc(tcap);
    if (!tmp) return ORC_E_ALLOC;
    uint8_t *p = tmp + tcap;
    uint32_t R[4] = {RANS_L, RANS_L, RANS_L, RANS_L};
    for (int32_t j = npairs; j > 0; j--) {
        uint32_t lo = pairs[j - 1] & 0xffff, fr = pairs[j - 1] >> 16;
        uint32_t x = R[3];
        uint32_t xmax = ((RANS_L >> PROB_BITS) << 8) * fr;
        while (x >= xmax) { *--p = (uint8_t)x; x >>= 8; }
        x = ((x / fr) << PROB_BITS) + (x % fr) + lo;
        R[3] = R[2]; R[2] = R[1]; R[1] = R[0]; R[0] = x;
    }
    for (int k = 3; k >= 0; k--) {
        p -= 4;
        p[0] = (uint8_t)R[k]; p[1] = (uint8_t)(R[k] >> 8); p[2] = (uint8_t)(R[k] >> 16); p[3] = (uint8_t)(R[k] >> 24);
    }
    int32_t sz = (int32_t)(tmp + tcap - p);
    if (sz > cap) { free(tmp); return ORC_E_CAPACITY; }
    memcpy(out, p, (size_t)sz);
    free(tmp);
    return sz;
}

/* ------------------------------------------------------------------------------------------ */
/* LEB128 "with carry" -- utils.cpp:22-90, utils.hpp:19-22                                      */
/* ------------------------------------------------------------------------------------------ */
static const int32_t LEBC[4] = {127, 16510, 2113661, 270549116};
int orc_leb_encode(int32_t v, uint8_t *b)
{
    int n = 1;
    while (n < 5 && v >= LEBC[n - 1]) n++;
    if (n > 1) v -= LEBC[n - 2];
    for (int k = 0; k < n; k++) b[k] = (uint8_t)((v >> (7 * (n - 1 - k))) & 0x7f);
    b[n - 1] |= 0x80;
    return n;
}
/* returns bytes consumed, or ORC_E_CORRUPT when no terminator within 5 bytes / avail */
int orc_leb_decode(int32_t *v, const uint8_t *b, int32_t avail)
{
    int d = 0;
    uint32_t x = 0;
    while (d < avail && !(b[d] & 0x80)) {
        if (d >= 4) return ORC_E_CORRUPT;
        x = (x << 7) | b[d++];
    }
    if (d >= avail) return ORC_E_CORRUPT;
    x = (x << 7) | (b[d] & 0x7f);
    if (d > 0) x += (uint32_t)LEBC[d - 1];
    *v = (int32_t)x;
    return d + 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Ans::Encode -- ans.cpp:113-234.  Clobbers `in` (rank coding is in place, rank.cpp:88).      */
/* ------------------------------------------------------------------------------------------ */
int orc_ans_encode(uint8_t *in, int32_t len, uint8_t *out, int32_t cap, int32_t *out_len)
{
    uint16_t *rle = (uint16_t *)malloc(CHUNK * 2);
    uint32_t *pairs = (uint32_t *)malloc((size_t)CHUNK * 2 * 4);
    uint8_t *pay = (uint8_t *)malloc((size_t)CHUNK * 4 + 16);
    if (!rle || !pairs || !pay) { free(rle); free(pairs); free(pay); return ORC_E_ALLOC; }
    int32_t ip = 0, op = 0, rc = ORC_OK;
    while (ip < len) {
        int32_t clen = (ip + CHUNK < len) ? CHUNK : len - ip;
        int32_t freq[256];
        if ((rc = orc_rank_encode(in + ip, freq, clen))) break;
        int32_t rlen = orc_rle_encode(in + ip, rle, clen);
        if ((rc = orc_model_pairs(rle, rlen, pairs))) break;
        int32_t csize = orc_rans_encode_pairs(pairs, 2 * rlen, pay, CHUNK * 4 + 16);
        if (csize < 0) { rc = csize; break; }
        uint8_t hdr[259 * 5];
        int hp = 0;
        for (int s = 0; s < 256; s++) hp += orc_leb_encode(freq[s], hdr + hp);   /* ans.cpp:272-285 */
        hp += orc_leb_encode(clen, hdr + hp);
        hp += orc_leb_encode(csize, hdr + hp);
        hp += orc_leb_encode(rlen, hdr + hp);
        if ((int64_t)op + hp + csize > cap) { rc = ORC_E_CAPACITY; break; }
        memcpy(out + op, hdr, (size_t)hp); op += hp;
        memcpy(out + op, pay, (size_t)csize); op += csize;
        ip += clen;
    }
    *out_len = op;
    free(rle); free(pairs); free(pay);
    return rc;
}

/* rANS + model decode of one chunk payload -> rle symbols -- ans.cpp:30-92 */
int orc_rans_decode_chunk(const uint8_t *pay, int32_t clen, int32_t rlen, uint16_t *rle)
{
    if (clen < 16) return ORC_E_CORRUPT;
    Models M;
    models_reset(&M);
    const uint8_t *p = pay, *end = pay + clen;
    uint32_t R[4];
    for (int k = 0; k < 4; k++) { R[k] = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); p += 4; }
    for (int32_t i = 0; i < rlen; i++) {
        int e, m;
        for (int half = 0; half < 2; half++) {
            uint32_t x = R[0], range = x & 0xffff;
            const int32_t *cdf;
            int A, sym = 0;
            if (half == 0) { cdf = M.ex.cdf; A = 8; }
            else if (e < 2) { cdf = M.mp[e].cdf; A = 2; }
            else { cdf = M.ms[e - 2].cdf; A = M.ms[e - 2].A; }
            while (sym + 1 < A && (uint32_t)cdf[sym + 1] <= range) sym++;
            uint32_t lo = (uint32_t)cdf[sym], fr = (uint32_t)(cdf[sym + 1] - cdf[sym]);
            x = fr * (x >> PROB_BITS) + range - lo;
            while (x < RANS_L) {
                if (p >= end) return ORC_E_CORRUPT;
                x = (x << 8) | *p++;
            }
            R[0] = R[1]; R[1] = R[2]; R[2] = R[3]; R[3] = x;
            if (half == 0) { e = sym; ad_update(&M.ex, e); }
            else {
                m = sym;
                if (e < 2) ad_update(&M.mp[e], m); else qs_update(&M.ms[e - 2], m);
            }
        }
        rle[i] = (uint16_t)(EXPO[e] + m);
    }
    if (R[0] != RANS_L || R[1] != RANS_L || R[2] != RANS_L || R[3] != RANS_L) return ORC_E_CORRUPT; /* ans.cpp:91 */
    return ORC_OK;
}

/* Ans::Decode -- ans.cpp:236-270 (+ header ans.cpp:287-302) */
int orc_ans_decode(const uint8_t *in, int32_t len, uint8_t *out, int32_t cap, int32_t *out_len)
{
    uint16_t *rle = (uint16_t *)malloc(CHUNK * 2);
    if (!rle) return ORC_E_ALLOC;
    int32_t ip = 0, op = 0, rc = ORC_OK;
    while (ip < len) {
        int32_t freq[256], olen, clen, rlen;
        int n;
        for (int s = 0; s < 256 && rc == ORC_OK; s++) {
            n = orc_leb_decode(&freq[s], in + ip, len - ip);
            if (n < 0) rc = n; else ip += n;
        }
        if (rc) break;
        if ((n = orc_leb_decode(&olen, in + ip, len - ip)) < 0) { rc = n; break; } ip += n;
        if ((n = orc_leb_decode(&clen, in + ip, len - ip)) < 0) { rc = n; break; } ip += n;
        if ((n = orc_leb_decode(&rlen, in + ip, len - ip)) < 0) { rc = n; break; } ip += n;
        if (olen < 0 || olen > CHUNK || rlen < 0 || rlen > CHUNK || clen < 0 || clen > len - ip) { rc = ORC_E_CORRUPT; break; }
        if ((int64_t)op + olen > cap) { rc = ORC_E_CAPACITY; break; }
        if ((rc = orc_rans_decode_chunk(in + ip, clen, rlen, rle))) break;
        if (orc_rle_decode(rle, out + op, rlen, olen) < 0) { rc = ORC_E_CORRUPT; break; }
        if ((rc = orc_rank_decode(out + op, freq, olen))) break;
        ip += clen;
        op += olen;
    }
    *out_len = op;
    free(rle);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* SURVEY section 8f rows 1 and 3 (block container): checksum.cpp:12-36 and the 15-byte block     */
/* header of Jampack::CompWriteBlock / DecompReadBlock (jampack.cpp:122-164).                    */
/* ------------------------------------------------------------------------------------------ */
/* checksum.cpp:12-36: four lanes, S = {3,0,0,0}; 16 bytes per round while j + 16 < size, then the
 * remaining bytes one at a time into lane 0. */
uint32_t orc_checksum(const uint8_t *p, int32_t size)
{
    const uint32_t prime = 0x9E3779B1u;
    uint32_t S[4] = {3u, 0u, 0u, 0u};
    uint32_t j = 0;
    while ((uint64_t)j + 16 < (uint64_t)(uint32_t)size) {
        for (int k = 0; k < 4; k++) {
            const uint8_t *q = p + j + 4 * k;
            uint32_t w = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
            S[k] ^= (w + (1u << (S[k] & 7))) * prime;
        }
        j += 16;
    }
    while (j < (uint32_t)size) {
        S[0] ^= ((uint32_t)p[j] + (1u << (S[0] & 7))) * prime;
        j++;
    }
    return S[0] ^ S[1] ^ S[2] ^ S[3];
}

/* jampack.cpp:128-132: "JAM" | crc (4, native LE) | compressed size (4) | BlockSize (4) */
int orc_block_header(uint32_t crc, int32_t comp_size, int32_t block_size, uint8_t *out15)
{
    out15[0] = 'J'; out15[1] = 'A'; out15[2] = 'M';
    memcpy(out15 + 3, &crc, 4);
    memcpy(out15 + 7, &comp_size, 4);
    memcpy(out15 + 11, &block_size, 4);
    return 15;
}
