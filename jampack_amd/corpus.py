"""Deterministic synthetic corpora standing in for enwik6/8/9 and Silesia (none are available offline).

Everything is driven by a counter-based splitmix64 generator written in numpy, so a (kind, nbytes, seed)
triple gives the same bytes on every machine.  ``JAMPACK_CORPUS_DIR`` may point at real files
(enwik8, enwik9, silesia.tar); ``load_or_make`` prefers them when present.

Workloads:
  C1 enwik6-like : text, 1 000 000 B, seed 6
  C2/C3 enwik8-like : text, 100 000 000 B, seed 8   -> blocks 67 108 864 + 32 891 136
  C4 enwik9-like : text, 1 000 000 000 B, seed 9
  C5 silesia-like: mixed, 211 938 580 B, seed 5
"""
from __future__ import annotations

import functools
import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """count 64-bit outputs of splitmix64 seeded with `seed`, outputs start .. start+count-1."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, start: int, count: int) -> np.ndarray:
    return (splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)


def _vocab(seed: int, nwords: int = 50000):
    lens = (2 + (splitmix64(seed ^ 0x5EED, 0, nwords) % np.uint64(10))).astype(np.int64)
    starts = np.concatenate(([0], np.cumsum(lens)))[:-1]
    total = int(lens.sum())
    p = np.arange(1, 27, dtype=np.float64) ** -0.8
    cdf = np.cumsum(p / p.sum())
    u = _uniform(seed ^ 0xABCD, 0, total)
    letters = _LETTERS[np.minimum(np.searchsorted(cdf, u), 25)]
    return letters, starts, lens


@functools.lru_cache(maxsize=4)
def _phrases(seed: int, nphr: int = 200000):
    """phrase book: each phrase is 1..8 Zipf-chosen vocabulary words; gives text multi-word repeats like real prose."""
    letters, vstart, vlen = _vocab(seed)
    nw = len(vlen)
    pz = np.arange(1, nw + 1, dtype=np.float64) ** -1.05
    zcdf = np.cumsum(pz / pz.sum())
    pwords = (1 + (splitmix64(seed ^ 0x9A5E, 0, nphr) % np.uint64(8))).astype(np.int64)
    tw = int(pwords.sum())
    ids = np.minimum(np.searchsorted(zcdf, _uniform(seed ^ 0x1D5, 0, tw)), nw - 1)
    wl = vlen[ids] + 1                       # word + separator
    ends = np.cumsum(wl)
    begs = ends - wl
    tot = int(ends[-1])
    pos = np.arange(tot, dtype=np.int64) - np.repeat(begs, wl)
    is_sep = pos == np.repeat(vlen[ids], wl)
    src = np.where(is_sep, 0, np.repeat(vstart[ids], wl) + pos)
    pbytes = np.where(is_sep, np.uint8(32), letters[src]).astype(np.uint8)
    # phrase extents in pbytes; every 8th phrase ends its last separator with a newline
    wend = np.cumsum(pwords)
    pend = ends[wend - 1]
    pbeg = np.concatenate(([0], pend[:-1]))
    nl = pend[7::8] - 1
    pbytes[nl] = 10
    return pbytes, pbeg, pend - pbeg


def text(nbytes: int, seed: int) -> np.ndarray:
    """Zipf-phrase text: 50 000-word vocabulary, 200 000-phrase book, phrase choice Zipf s=0.8."""
    if nbytes <= 0:
        return np.zeros(0, dtype=np.uint8)
    pbytes, pbeg, plen = _phrases(seed)
    nphr = len(plen)
    pz = np.arange(1, nphr + 1, dtype=np.float64) ** -0.8
    zcdf = np.cumsum(pz / pz.sum())
    out = np.empty(nbytes, dtype=np.uint8)
    filled = 0
    widx = 0
    batch = 1 << 19
    while filled < nbytes:
        u = _uniform(seed, widx, batch)
        ids = np.minimum(np.searchsorted(zcdf, u), nphr - 1)
        ln = plen[ids]
        ends = np.cumsum(ln)
        tot = int(ends[-1])
        src = np.arange(tot, dtype=np.int64) - np.repeat(ends - ln, ln) + np.repeat(pbeg[ids], ln)
        take = min(tot, nbytes - filled)
        out[filled:filled + take] = pbytes[src[:take]]
        filled += take
        widx += batch
    return out


SURVEY_PAGE = 4 << 20


@functools.lru_cache(maxsize=4)
def _survey_tables(seed: int):
    letters, vstart, vlen = _vocab(seed)
    nw = len(vlen)
    pz = np.arange(1, nw + 1, dtype=np.float64) ** -1.05
    return letters, vstart, vlen, np.cumsum(pz / pz.sum())


def _survey_page(seed: int, page: int) -> np.ndarray:
    """page `page` (SURVEY_PAGE bytes) of the survey text stream: pages are generated independently from (seed, page), so
    any byte range of a long stream -- one 64 MiB block of the 1 GB workload -- costs only its own pages."""
    letters, vstart, vlen, zcdf = _survey_tables(seed)
    nw = len(vlen)
    words = SURVEY_PAGE // 6                       # mean word + separator is 7.5 bytes: always enough
    pseed = (seed * 0x9E3779B1 + 0x51 + page * 0x632BE5AB) & 0xFFFFFFFFFFFFFFFF
    ids = np.minimum(np.searchsorted(zcdf, _uniform(pseed, 0, words)), nw - 1)
    wl = vlen[ids] + 1
    ends = np.cumsum(wl)
    nwords = int(np.searchsorted(ends, SURVEY_PAGE)) + 1
    ids, wl, ends = ids[:nwords], wl[:nwords], ends[:nwords]
    begs = ends - wl
    tot = int(ends[-1])
    pos = np.arange(tot, dtype=np.int64) - np.repeat(begs, wl)
    is_sep = pos == np.repeat(vlen[ids], wl)
    src = np.where(is_sep, 0, np.repeat(vstart[ids], wl) + pos)
    seg = np.where(is_sep, np.uint8(32), letters[src]).astype(np.uint8)
    seg[ends[199999::200000] - 1] = 10             # newline every 200 000 words
    return seg[:SURVEY_PAGE]


def text_survey(nbytes: int, seed: int, start: int = 0) -> np.ndarray:
    """SURVEY.md section 8d text model (no phrase book): 50 000-word vocabulary (lengths U[2,11], letters p ~ rank^-0.8),
    word choice Zipf s=1.05, separator space, newline every 200 000 words.  Bytes [start, start + nbytes) of the stream."""
    if nbytes <= 0:
        return np.zeros(0, dtype=np.uint8)
    out = np.empty(nbytes, dtype=np.uint8)
    p0, p1 = start // SURVEY_PAGE, (start + nbytes - 1) // SURVEY_PAGE
    for p in range(p0, p1 + 1):
        page = _survey_page(seed, p)
        lo = max(start, p * SURVEY_PAGE)
        hi = min(start + nbytes, (p + 1) * SURVEY_PAGE)
        out[lo - start: hi - start] = page[lo - p * SURVEY_PAGE: hi - p * SURVEY_PAGE]
    return out


# ---- "wide" text: the survey text model with an enwik8-like BYTE ALPHABET ------------------------------------------------------
# SURVEY 8d's text model has 28 distinct bytes (lower-case letters, space, newline); real enwik8 has 205 (capitals, digits,
# punctuation, wiki / XML markup, UTF-8 lead and continuation bytes) at an order-0 entropy of 5.08 bits.  The suffix sort's first key
# depends on the alphabet (bwt_fwd.hip k_key_plan: 11 bytes for 28 values, 7 above 128), so the bench line carries this variant beside
# the headline: same vocabulary size, same Zipf word choice, same page scheme, but words come capitalised / upper-case / as numbers /
# with accented letters / in two- and three-byte scripts, and the separator is drawn from a table of punctuation and markup.
_Q3 = b"'" * 3
_WIDE_SEPS = [(b" ", 5200), (b", ", 560), (b". ", 460), (b"\n", 170), (b" (", 60), (b") ", 60), (b"; ", 30), (b": ", 80), (b" - ", 50), (b"'s ", 80),
              (b'" ', 40), (b' "', 40), (b"]] ", 160), (b" [[", 160), (b"|", 110), (b"'' ", 50), (b" ''", 50), (b"\n\n", 60), (b"\n* ", 40), (b"\n== ", 12),
              (b" ==\n", 12), (b"&quot;", 30), (b"&amp;", 8), (b"&lt;", 14), (b"&gt;", 14), (b" <", 14), (b"> ", 14), (b"/", 40), (b"=", 30), (b"{{", 20),
              (b"}} ", 20), (b"#", 8), (b"%", 6), (b"+", 6), (b"_", 30), (b"! ", 8), (b"? ", 10), (b"@", 2), (b"~", 2), (b"^", 1), (b"`", 1), (b"$", 3),
              (b"\\", 1), (b"\t", 4), (b"]", 6), (b"[", 6), (b"}", 2), (b"{", 2), (b"*", 6), (_Q3, 25), (b"-", 60), (b".", 40), (b",", 10), (b":", 30),
              (b"</", 10), (b"/>", 6), (b"\xe2\x80\x93", 12), (b"\xe2\x80\x94 ", 8), (b"\xc2\xa0", 6), (b"\xc2\xb0", 2), (b"\xe2\x80\x99", 6)]


@functools.lru_cache(maxsize=2)
def _wide_tables(seed: int):
    """piece pool of the wide text: pieces 0..nw-1 = the vocabulary's words in their wide spelling, nw.. = the separators;
    returns (pool bytes, piece start, piece length, word Zipf cdf, separator cdf, nw)"""
    letters, vstart, vlen = _vocab(seed)
    nw = len(vlen)
    h = splitmix64(seed ^ 0x71DE, 0, nw)
    kind = (h % np.uint64(1000)).astype(np.int64)
    kind[:40] = 999                                         # the most frequent words stay plain lower case
    r1 = ((h >> np.uint64(16)) % np.uint64(1 << 20)).astype(np.int64)
    pieces = []
    for k in range(nw):
        w = letters[vstart[k]: vstart[k] + vlen[k]]
        kd, r = int(kind[k]), int(r1[k])
        if kd < 150:                                        # Capitalised
            b = bytes([w[0] - 32]) + w[1:].tobytes()
        elif kd < 175:                                      # UPPER CASE
            b = (w - 32).astype(np.uint8).tobytes()
        elif kd < 260:                                      # a number of 1..4 digits; four digits are years
            nd = 1 + r % 4
            b = bytes(48 + ((r >> (3 * i)) % 10) for i in range(nd))
            if nd == 4:
                b = (b"19" if r & 1 else b"20") + b[2:]
        elif kd < 300:                                      # one accented letter (two-byte UTF-8, lead 0xC3..0xC5)
            i = r % len(w)
            b = w[:i].tobytes() + bytes([0xC3 + (r >> 8) % 3, 0x80 + (r >> 10) % 64]) + w[i + 1:].tobytes()
        elif kd < 320:                                      # a word in a two-byte script (leads 0xC6..0xDF: Greek, Cyrillic, Hebrew, Arabic, ...)
            lead = 0xC6 + (r >> 4) % 26
            b = b"".join(bytes([lead, 0x80 + (int(c) * 7 + r) % 64]) for c in w[: 2 + r % 5])
        elif kd < 335:                                      # a word in a three-byte script (leads 0xE0..0xEF)
            lead = 0xE0 + (r >> 3) % 16
            b = b"".join(bytes([lead, 0x80 + (int(c) * 5 + r) % 64, 0x80 + (int(c) * 11 + (r >> 7)) % 64]) for c in w[: 1 + r % 3])
        elif kd < 350:                                      # mixed case identifier with a digit
            b = bytes([w[0] - 32]) + w[1:].tobytes() + bytes([48 + r % 10])
        else:
            b = w.tobytes()
        pieces.append(b)
    sep_w = np.array([w for _, w in _WIDE_SEPS], dtype=np.float64)
    for sb, _ in _WIDE_SEPS:
        pieces.append(sb)
    plen = np.array([len(b) for b in pieces], dtype=np.int64)
    pstart = np.concatenate(([0], np.cumsum(plen)))[:-1]
    pool = np.frombuffer(b"".join(pieces), dtype=np.uint8).copy()
    pz = np.arange(1, nw + 1, dtype=np.float64) ** -1.05
    return pool, pstart, plen, np.cumsum(pz / pz.sum()), np.cumsum(sep_w / sep_w.sum()), nw


def _wide_page(seed: int, page: int) -> np.ndarray:
    pool, pstart, plen, zcdf, scdf, nw = _wide_tables(seed)
    words = SURVEY_PAGE // 5
    pseed = (seed * 0x9E3779B1 + 0x77 + page * 0x632BE5AB) & 0xFFFFFFFFFFFFFFFF
    wid = np.minimum(np.searchsorted(zcdf, _uniform(pseed, 0, words)), nw - 1)
    sid = nw + np.minimum(np.searchsorted(scdf, _uniform(pseed ^ 0x5E9, 0, words)), len(scdf) - 1)
    ids = np.empty(2 * words, dtype=np.int64)
    ids[0::2] = wid
    ids[1::2] = sid
    ln = plen[ids]
    ends = np.cumsum(ln)
    npieces = int(np.searchsorted(ends, SURVEY_PAGE)) + 1
    ids, ln, ends = ids[:npieces], ln[:npieces], ends[:npieces]
    tot = int(ends[-1])
    src = np.arange(tot, dtype=np.int64) - np.repeat(ends - ln, ln) + np.repeat(pstart[ids], ln)
    seg = pool[src]
    # a sentence starts with a capital: a lower-case letter behind ". " or a newline
    low = (seg >= 97) & (seg <= 122)
    after = np.zeros(tot, dtype=bool)
    after[2:] = ((seg[:-2] == 46) & (seg[1:-1] == 32))
    after[1:] |= seg[:-1] == 10
    seg = np.where(low & after, seg - 32, seg).astype(np.uint8)
    return seg[:SURVEY_PAGE]


def text_wide(nbytes: int, seed: int, start: int = 0) -> np.ndarray:
    """the survey text model over an enwik8-like byte alphabet (about 200 distinct bytes, order-0 entropy about 5 bits)"""
    if nbytes <= 0:
        return np.zeros(0, dtype=np.uint8)
    out = np.empty(nbytes, dtype=np.uint8)
    p0, p1 = start // SURVEY_PAGE, (start + nbytes - 1) // SURVEY_PAGE
    for p in range(p0, p1 + 1):
        page = _wide_page(seed, p)
        lo = max(start, p * SURVEY_PAGE)
        hi = min(start + nbytes, (p + 1) * SURVEY_PAGE)
        out[lo - start: hi - start] = page[lo - p * SURVEY_PAGE: hi - p * SURVEY_PAGE]
    return out


def random_bytes(nbytes: int, seed: int) -> np.ndarray:
    n8 = (nbytes + 7) // 8
    return splitmix64(seed, 0, n8).view(np.uint8)[:nbytes].copy()


def alphabet(nbytes: int, seed: int, symbols: bytes) -> np.ndarray:
    tab = np.frombuffer(symbols, dtype=np.uint8)
    r = random_bytes(nbytes, seed)
    return tab[r % len(tab)]


def geometric(nbytes: int, seed: int) -> np.ndarray:
    """bytes with P(v) ~ 2^-(v/8): exercises every exponent class of the entropy stage."""
    u = _uniform(seed, 0, nbytes)
    v = np.floor(-np.log2(np.maximum(u, 1e-300)) * 8.0)
    return np.minimum(v, 255).astype(np.uint8)


def samples16(nbytes: int, seed: int) -> np.ndarray:
    n = (nbytes + 1) // 2
    steps = (splitmix64(seed, 0, n) % np.uint64(65)).astype(np.int64) - 32
    wav = (np.cumsum(steps) & 0xFFFF).astype(np.uint16)
    return wav.view(np.uint8)[:nbytes].copy()


def runs(nbytes: int, seed: int) -> np.ndarray:
    """long runs of 0x00 / 0xFF with occasional other bytes."""
    out = np.empty(nbytes, dtype=np.uint8)
    nr = max(1, nbytes // 512)
    rl = 1 + (splitmix64(seed, 0, nr) % np.uint64(2048)).astype(np.int64)
    val = splitmix64(seed ^ 0x77, 0, nr)
    vals = np.where(val % np.uint64(16) == 0, (val >> np.uint64(8)) % np.uint64(256),
                    np.where(val % np.uint64(2) == 0, 0, 255)).astype(np.uint8)
    seq = np.repeat(vals, rl)
    while len(seq) < nbytes:
        seq = np.concatenate((seq, seq))
    out[:] = seq[:nbytes]
    return out


def repeated(nbytes: int, seed: int, period: int = 1 << 20) -> np.ndarray:
    """one text segment of `period` bytes repeated: LCP stress for suffix sorting."""
    period = max(1, min(period, nbytes))
    seg = text(period, seed)
    reps = (nbytes + period - 1) // period
    return np.tile(seg, reps)[:nbytes].copy()


def silesia_like(nbytes: int, seed: int) -> np.ndarray:
    parts = [("text", 0.40), ("samples16", 0.15), ("random", 0.15), ("dna", 0.10), ("runs", 0.10), ("repeat", 0.10)]
    out = []
    left = nbytes
    for k, (kind, frac) in enumerate(parts):
        n = left if k == len(parts) - 1 else int(nbytes * frac)
        out.append(make(kind, n, seed + 17 * k))
        left -= n
    return np.concatenate(out)


def make(kind: str, nbytes: int, seed: int) -> np.ndarray:
    if nbytes <= 0:
        return np.zeros(0, dtype=np.uint8)
    if kind == "text":
        return text(nbytes, seed)
    if kind == "text_survey":
        return text_survey(nbytes, seed)
    if kind == "text_wide":
        return text_wide(nbytes, seed)
    if kind == "random":
        return random_bytes(nbytes, seed)
    if kind == "dna":
        return alphabet(nbytes, seed, b"ACGT")
    if kind == "two":
        return alphabet(nbytes, seed, b"ab")
    if kind == "zero":
        return np.zeros(nbytes, dtype=np.uint8)
    if kind == "geometric":
        return geometric(nbytes, seed)
    if kind == "samples16":
        return samples16(nbytes, seed)
    if kind == "runs":
        return runs(nbytes, seed)
    if kind == "repeat":
        return repeated(nbytes, seed)
    if kind == "repeat4k":
        return repeated(nbytes, seed, 4096)
    if kind == "silesia":
        return silesia_like(nbytes, seed)
    raise ValueError(f"unknown corpus kind {kind!r}")


KINDS = ("text", "text_survey", "text_wide", "random", "dna", "two", "zero", "geometric", "samples16", "runs", "repeat", "repeat4k", "silesia")

# name -> (kind, bytes, seed, real file under JAMPACK_CORPUS_DIR).  The enwik workloads use the SURVEY 8d text model
# (ratio ~20 %, as enwik8's ~21 %); the *-phrase variants add a 200 000-phrase book (deeper repeats, ratio ~10 %) and are
# reported beside them.
WORKLOADS = {
    "enwik6": ("text_survey", 1_000_000, 6, "enwik6"),
    "enwik8": ("text_survey", 100_000_000, 8, "enwik8"),
    "enwik9": ("text_survey", 1_000_000_000, 9, "enwik9"),
    "enwik8-phrase": ("text", 100_000_000, 8, None),
    "enwik8-wide": ("text_wide", 100_000_000, 8, None),
    "silesia": ("silesia", 211_938_580, 5, "silesia.tar"),
}


def workload_bytes(name: str) -> int:
    return WORKLOADS[name][1]


def load_or_make(name: str, limit: int | None = None, seed_offset: int = 0, start: int = 0, count: int | None = None):
    """bytes [start, start+count) of a workload; returns (ndarray, source) where source is 'file:<path>' or 'synthetic'."""
    kind, nbytes, seed, fname = WORKLOADS[name]
    if limit is not None:
        nbytes = min(nbytes, limit)
    start = min(start, nbytes)
    count = nbytes - start if count is None else min(count, nbytes - start)
    d = os.environ.get("JAMPACK_CORPUS_DIR")
    if d and seed_offset == 0 and fname:
        p = os.path.join(d, fname)
        if os.path.isfile(p):
            return np.fromfile(p, dtype=np.uint8, count=count, offset=start), f"file:{p}"
    cache = os.environ.get("JAMPACK_CORPUS_CACHE")      # directory: generated ranges are kept as raw files (repeated bench runs)
    cpath = os.path.join(cache, f"{name}_{seed + seed_offset}_{nbytes}_{start}_{count}.bin") if cache else None
    if cpath and os.path.isfile(cpath):
        return np.fromfile(cpath, dtype=np.uint8), "synthetic"
    if kind == "text_survey":
        out = text_survey(count, seed + seed_offset, start)
    elif kind == "text_wide":
        out = text_wide(count, seed + seed_offset, start)
    else:
        out = make(kind, nbytes, seed + seed_offset)[start:start + count]
    if cpath:
        os.makedirs(cache, exist_ok=True)
        out.tofile(cpath)
    return out, "synthetic"


SOURCE_ROOTS = ("/usr/lib/python3.10", "/usr/lib/python3/dist-packages", "/opt/rocm/include", "/usr/local/lib/python3.10/dist-packages")
SOURCE_SUFFIXES = (".py", ".pyi", ".h", ".hpp", ".c", ".inc", ".txt", ".md", ".rst", ".json", ".yaml", ".cmake")


def system_sources(nbytes: int):
    """REAL data for the checks that want some: the first `nbytes` of the source and text files of this image's Python and ROCm trees,
    in sorted path order (the GPU box runs the same image).  None if the trees hold less.  Not a benchmark corpus -- the numbers BASELINE.json
    names are quoted on the synthetic workloads above -- but real files have what generators lack: licence headers repeated thousands of
    times, generated tables, long runs of blanks, near-duplicate files."""
    parts, left = [], nbytes
    for root in SOURCE_ROOTS:
        for d, dirs, files in os.walk(root):
            dirs.sort()
            for f in sorted(files):
                if not f.endswith(SOURCE_SUFFIXES):
                    continue
                try:
                    a = np.fromfile(os.path.join(d, f), dtype=np.uint8)
                except OSError:
                    continue
                if len(a) == 0:
                    continue
                parts.append(a[:left])
                left -= len(parts[-1])
                if left == 0:
                    return np.concatenate(parts)
    return None


def system_binaries(nbytes: int, skip: int = 0):
    """`nbytes` of REAL machine code and tables: the largest shared libraries of this image's ROCm tree in sorted name order, from byte
    `skip` of their concatenation (ELF sections, gfx code objects, symbol and string tables, zero padding).  None if there is less."""
    root = "/opt/rocm/lib"
    try:
        names = sorted(f for f in os.listdir(root) if ".so" in f and not os.path.islink(os.path.join(root, f)))
    except OSError:
        return None
    parts, left = [], nbytes
    for f in names:
        p = os.path.join(root, f)
        try:
            sz = os.path.getsize(p)
        except OSError:
            continue
        if skip >= sz:
            skip -= sz
            continue
        a = np.fromfile(p, dtype=np.uint8, count=min(left, sz - skip), offset=skip)
        skip = 0
        if len(a) == 0:
            continue
        parts.append(a)
        left -= len(a)
        if left == 0:
            return np.concatenate(parts)
    return None


def block_ranges(total: int, block_size: int):
    """(start, length) of every block of a `total`-byte stream (jampack.cpp:205-213 reads BlockSize bytes per block)"""
    return [(o, min(block_size, total - o)) for o in range(0, total, block_size)]


def split_blocks(data: np.ndarray, block_size: int):
    return [data[i:i + block_size] for i in range(0, len(data), block_size)]
