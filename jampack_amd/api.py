"""Python mirror of the reference's operator interface for the block hot path, on top of the C ABI.

Reference interface mirrored (same names and argument meaning; errors raise JampackError instead of exit(-1)):
  BlockSort::Bwt::ForwardBwt / InverseBwt   bwt.hpp:13-18
  Ans::Encode / Ans::Decode                 ans.hpp:32-33
  Postcoder::Encode / Decode                rank.hpp:12-13
Host-buffer calls take/return numpy uint8 arrays; `Context` exposes the device-buffer entry points for data that
already lives in HBM (torch CUDA tensors or raw device pointers).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import CHUNK, TRAILER, JampackError, Stats, lib


def _np_u8(a) -> np.ndarray:
    return np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray, memoryview)) else a, dtype=np.uint8)


def _ptr(a: np.ndarray):
    return a.ctypes.data if a.size else None


def _chk(rc: int, what: str):
    if rc != 0:
        raise JampackError(rc, what)


def ans_capacity(n: int) -> int:
    """generous output bound for Ans::Encode of n bytes (the reference gives 1.05 x BlockSize, jampack.cpp:74)"""
    return int(n * 1.25) + 4096 + 1400 * (n // CHUNK + 1)


def init(device_mask: int = 0) -> int:
    """jpk_init: devices the host-buffer entry points may use (bit d = device d, 0 = all); returns how many were selected"""
    n = lib().jpk_init(device_mask)
    if n < 0:
        raise JampackError(n, "jpk_init")
    return n


def shutdown() -> None:
    lib().jpk_shutdown()


def release_idle() -> int:
    """destroys the batch entries' idle worker contexts and the multi-device slabs; returns the contexts destroyed"""
    return int(lib().jpk_release_idle())


def thread_device() -> int:
    """device the calling thread's pooled context lives on"""
    d = lib().jpk_thread_device()
    if d < 0:
        raise JampackError(d, "jpk_thread_device")
    return d


def ans_decoded_size(stream):
    """(decoded bytes, chunks) declared by the chunk headers of an Ans stream (host-side header walk)"""
    c = _np_u8(stream)
    n, k = C.c_int64(0), C.c_int32(0)
    _chk(lib().jpk_ans_decoded_size(_ptr(c), len(c), C.byref(n), C.byref(k)), "jpk_ans_decoded_size")
    return n.value, k.value


class Bwt:
    """BlockSort::Bwt (bwt.hpp:13-18)."""

    def ForwardBwt(self, block, out: np.ndarray | None = None) -> np.ndarray:
        t = _np_u8(block)
        if out is None:
            out = np.zeros(len(t) + TRAILER, dtype=np.uint8)
        n = C.c_int32(0)
        _chk(lib().jpk_bwt_forward(_ptr(t), len(t), out.ctypes.data, len(out), C.byref(n)), "ForwardBwt")
        return out[: n.value]

    def InverseBwt(self, bwt, threads: int = 1, gpu: bool = True) -> np.ndarray:
        b = _np_u8(bwt)
        out = np.zeros(max(len(b), 1), dtype=np.uint8)
        n = C.c_int32(0)
        _chk(lib().jpk_bwt_inverse(_ptr(b), len(b), out.ctypes.data, len(out), C.byref(n), threads, int(gpu)), "InverseBwt")
        return out[: n.value]


class Ans:
    """Ans (ans.hpp:15-45)."""

    def Encode(self, data, cap: int | None = None) -> np.ndarray:
        x = np.array(_np_u8(data), copy=True)       # the ABI may clobber its input like the reference (rank.cpp:88)
        cap = ans_capacity(len(x)) if cap is None else cap
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_int32(0)
        _chk(lib().jpk_ans_encode(_ptr(x), len(x), out.ctypes.data, cap, C.byref(n)), "Ans::Encode")
        return out[: n.value]

    def Decode(self, data, cap: int, threads: int = 1) -> np.ndarray:
        c = _np_u8(data)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_int32(0)
        _chk(lib().jpk_ans_decode(_ptr(c), len(c), out.ctypes.data, cap, C.byref(n), threads), "Ans::Decode")
        return out[: n.value]


class Postcoder:
    """Postcoder (rank.hpp:9-16)."""

    def Encode(self, t):
        r = np.array(_np_u8(t), copy=True)
        f = np.zeros(256, dtype=np.int32)
        _chk(lib().jpk_rank_encode(_ptr(r), f.ctypes.data, len(r)), "Postcoder::Encode")
        return r, f

    def Decode(self, ranks, freq) -> np.ndarray:
        r = np.array(_np_u8(ranks), copy=True)
        f = np.ascontiguousarray(freq, dtype=np.int32)
        _chk(lib().jpk_rank_decode(_ptr(r), f.ctypes.data, len(r)), "Postcoder::Decode")
        return r


def block_compress(block, cap: int | None = None) -> np.ndarray:
    """ForwardBwt + Ans::Encode (jampack.cpp:40-41) with the BWT image kept in HBM."""
    t = _np_u8(block)
    cap = ans_capacity(len(t) + TRAILER) if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_int32(0)
    _chk(lib().jpk_block_compress(_ptr(t), len(t), out.ctypes.data, cap, C.byref(n)), "block_compress")
    return out[: n.value]


def block_decompress(comp, cap: int) -> np.ndarray:
    """Ans::Decode + InverseBwt (jampack.cpp:49-50)."""
    c = _np_u8(comp)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_int32(0)
    _chk(lib().jpk_block_decompress(_ptr(c), len(c), out.ctypes.data, cap, C.byref(n)), "block_decompress")
    return out[: n.value]


JAM_HEADER = 15
MIN_BLOCKSIZE = 1 << 20        # format.hpp:21
MAX_BLOCKSIZE = 1000 << 20     # format.hpp:22


class Checksum:
    """Mirror of `class Checksum` (checksum.hpp:12-16)."""

    def IntegrityCheck(self, buf) -> int:
        t = _np_u8(buf)
        crc = C.c_uint32(0)
        _chk(lib().jpk_checksum(_ptr(t), len(t), C.byref(crc)), "Checksum::IntegrityCheck")
        return crc.value


def jam_block_write(block, block_size: int, cap: int | None = None) -> np.ndarray:
    """One framed block: crc (jampack.cpp:31) + the 15-byte header of CompWriteBlock (jampack.cpp:122-135) +
    the block_compress payload."""
    t = _np_u8(block)
    cap = JAM_HEADER + ans_capacity(len(t) + TRAILER) if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_int32(0)
    _chk(lib().jpk_jam_block_write(_ptr(t), len(t), block_size, out.ctypes.data, cap, C.byref(n)), "jam_block_write")
    return out[: n.value]


def jam_block_read(stream, cap: int):
    """DecompReadBlock + Decomp (jampack.cpp:140-164, 47-60) for the frame at the start of `stream`.
    Returns (block bytes, bytes consumed); raises JampackError(CORRUPT) on a bad header or crc."""
    c = _np_u8(stream)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n, used = C.c_int32(0), C.c_int32(0)
    _chk(lib().jpk_jam_block_read(_ptr(c), len(c), out.ctypes.data, cap, C.byref(n), C.byref(used)), "jam_block_read")
    return out[: n.value], used.value


def jam_compress(data, block_size: int = 8 << 20) -> np.ndarray:
    """Jampack::Compress's block loop (jampack.cpp:186-254) over an in-memory buffer: consecutive frames of
    block_size input bytes (DEFAULT_BLOCKSIZE 8 MiB, format.hpp:20)."""
    t = _np_u8(data)
    frames = [jam_block_write(t[o: o + block_size], block_size) for o in range(0, len(t), block_size)]
    return np.concatenate(frames) if frames else np.zeros(0, dtype=np.uint8)


def jam_decompress(stream) -> np.ndarray:
    """Jampack::Decompress's block loop (jampack.cpp:262-336): frames until the stream ends."""
    c = _np_u8(stream)
    out, o = [], 0
    while o < len(c):
        if len(c) - o < JAM_HEADER:
            raise JampackError(-3, "jam_decompress: truncated header")
        bs = int(np.frombuffer(c[o + 11: o + 15].tobytes(), dtype="<i4")[0])
        if not (MIN_BLOCKSIZE <= bs <= MAX_BLOCKSIZE):
            raise JampackError(-3, "jam_decompress: Refusing to read from corrupt header!")
        blk, used = jam_block_read(c[o:], bs)
        out.append(blk)
        o += used
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)


class Lz77:
    """Decoder side of `class Lz77` (lz77.hpp:21-22); host code."""

    def Decompress(self, buf, cap: int) -> np.ndarray:
        t = _np_u8(buf)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_int32(0)
        _chk(lib().jpk_lz77_decompress(_ptr(t), len(t), out.ctypes.data, cap, C.byref(n)), "Lz77::Decompress")
        return out[: n.value]


class Lpx:
    """Decoder side of `class Lpx` (lpx.hpp:31-32); host code."""

    def Decode(self, buf) -> np.ndarray:
        t = _np_u8(buf)
        out = np.zeros(max(len(t), 1), dtype=np.uint8)
        _chk(lib().jpk_lpx_decode(_ptr(t), len(t), out.ctypes.data), "Lpx::Decode")
        return out[: len(t)]


class Filters:
    """Decoder side of `class Filters` (filters.hpp:43-44); host code."""

    def Decode(self, buf, cap: int) -> np.ndarray:
        t = _np_u8(buf)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_int32(0)
        _chk(lib().jpk_filters_decode(_ptr(t), len(t), out.ctypes.data, cap, C.byref(n)), "Filters::Decode")
        return out[: n.value]


def checksum_host(buf) -> int:
    t = _np_u8(buf)
    return int(lib().jpk_checksum_host(_ptr(t), len(t)))


def jam_cli_block_read(stream, cap: int):
    """One frame written by an unmodified `jampack c`: the whole Jampack::Decomp() (jampack.cpp:47-60), entropy decode and
    inverse BWT on the GPU, the pre-stage decoders on the host.  Returns (block bytes, bytes consumed)."""
    c = _np_u8(stream)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n, used = C.c_int32(0), C.c_int32(0)
    _chk(lib().jpk_jam_cli_block_read(_ptr(c), len(c), out.ctypes.data, cap, C.byref(n), C.byref(used)), "jam_cli_block_read")
    return out[: n.value], used.value


def jam_cli_decompress(stream) -> np.ndarray:
    """`jampack d` over an in-memory .jam stream (Jampack::Decompress's block loop, jampack.cpp:262-336)."""
    c = _np_u8(stream)
    out, o = [], 0
    while o < len(c):
        if len(c) - o < JAM_HEADER:
            raise JampackError(-3, "jam_cli_decompress: truncated header")
        bs = int(np.frombuffer(c[o + 11: o + 15].tobytes(), dtype="<i4")[0])
        if not (MIN_BLOCKSIZE <= bs <= MAX_BLOCKSIZE):
            raise JampackError(-3, "jam_cli_decompress: Refusing to read from corrupt header!")
        blk, used = jam_cli_block_read(c[o:], bs)
        out.append(blk)
        o += used
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)


def _dptr(x):
    """device pointer of a torch CUDA tensor or a raw int"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr() if x.numel() else None


class Context:
    """jpk_ctx: one HBM arena + stream per context.  `stream` is a raw hipStream_t (e.g.
    torch.cuda.current_stream().cuda_stream) or None for a private stream."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        _chk(lib().jpk_ctx_create(C.byref(self._h), device, stream), "jpk_ctx_create")

    def close(self):
        if self._h:
            lib().jpk_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, max_block_bytes: int):
        _chk(lib().jpk_ctx_reserve(self._h, max_block_bytes), "jpk_ctx_reserve")

    def stats(self) -> Stats:
        s = Stats()
        _chk(lib().jpk_ctx_stats(self._h, C.byref(s)), "jpk_ctx_stats")
        return s

    def profile_enable(self, mode: int = 2):
        """per-kernel HIP-event timing on the context's stream: 1 = on, 2 = on + reset, 0 = off + reset"""
        _chk(lib().jpk_ctx_profile(self._h, mode), "jpk_ctx_profile")

    def profile_table(self):
        """[{name, ms, launches, units}] for every timed kernel class with at least one launch"""
        out = []
        for i in range(lib().jpk_ctx_profile_count()):
            ms, ln, un = C.c_double(0), C.c_int64(0), C.c_int64(0)
            _chk(lib().jpk_ctx_profile_get(self._h, i, C.byref(ms), C.byref(ln), C.byref(un)), "jpk_ctx_profile_get")
            if ln.value:
                out.append({"id": i, "name": lib().jpk_ctx_profile_name(i).decode(), "ms": ms.value, "launches": ln.value, "units": un.value})
        return out

    def _io(self, fn, what, d_in, in_len, d_out, out_cap) -> int:
        n = C.c_int32(0)
        _chk(fn(self._h, _dptr(d_in), in_len, _dptr(d_out), out_cap, C.byref(n)), what)
        return n.value

    def bwt_forward(self, d_in, in_len, d_out, out_cap) -> int:
        return self._io(lib().jpk_dev_bwt_forward, "jpk_dev_bwt_forward", d_in, in_len, d_out, out_cap)

    def bwt_inverse(self, d_in, in_len, d_out, out_cap) -> int:
        return self._io(lib().jpk_dev_bwt_inverse, "jpk_dev_bwt_inverse", d_in, in_len, d_out, out_cap)

    def bwt_inverse_chains120(self, d_in, in_len, d_out, out_cap):
        """the reference's 120-chain chase (comparator); returns (bytes, chase kernel ms)"""
        n, ms = C.c_int32(0), C.c_float(0)
        _chk(lib().jpk_dev_bwt_inverse_chains120(self._h, _dptr(d_in), in_len, _dptr(d_out), out_cap, C.byref(n), C.byref(ms)), "jpk_dev_bwt_inverse_chains120")
        return n.value, ms.value

    def ans_encode(self, d_in, in_len, d_out, out_cap) -> int:
        return self._io(lib().jpk_dev_ans_encode, "jpk_dev_ans_encode", d_in, in_len, d_out, out_cap)

    def ans_decode(self, d_in, in_len, d_out, out_cap) -> int:
        return self._io(lib().jpk_dev_ans_decode, "jpk_dev_ans_decode", d_in, in_len, d_out, out_cap)

    def block_compress(self, d_in, in_len, d_out, out_cap) -> int:
        return self._io(lib().jpk_dev_block_compress, "jpk_dev_block_compress", d_in, in_len, d_out, out_cap)

    def block_decompress(self, d_in, in_len, d_out, out_cap) -> int:
        return self._io(lib().jpk_dev_block_decompress, "jpk_dev_block_decompress", d_in, in_len, d_out, out_cap)

    def _batch(self, fn, what, d_ins, in_lens, d_outs, out_caps):
        """(out_len list, status list) of a jpk_dev_blocks_* call: one pass over all blocks"""
        n = len(d_ins)
        P, I = C.c_void_p * n, C.c_int32 * n
        ins, outs = P(*[_dptr(x) for x in d_ins]), P(*[_dptr(x) for x in d_outs])
        il, oc, ol, st = I(*in_lens), I(*out_caps), I(), I()
        _chk(fn(self._h, n, ins, il, outs, oc, ol, st), what)
        return list(ol), list(st)

    def blocks_ans_decode(self, d_ins, in_lens, d_outs, out_caps):
        return self._batch(lib().jpk_dev_blocks_ans_decode, "jpk_dev_blocks_ans_decode", d_ins, in_lens, d_outs, out_caps)

    def blocks_decompress(self, d_ins, in_lens, d_outs, out_caps):
        return self._batch(lib().jpk_dev_blocks_decompress, "jpk_dev_blocks_decompress", d_ins, in_lens, d_outs, out_caps)

    def blocks_compress(self, d_ins, in_lens, d_outs, out_caps, in_flight: int = 0):
        """ForwardBwt + Ans::Encode of independent blocks in one call; the library keeps `in_flight` (0: 4) of them in flight on
        contexts of its own (jampack.cpp:205-224's OpenMP block loop).  Returns (out_len list, status list)."""
        n = len(d_ins)
        P, I = C.c_void_p * n, C.c_int32 * n
        ins, outs = P(*[_dptr(x) for x in d_ins]), P(*[_dptr(x) for x in d_outs])
        il, oc, ol, st = I(*in_lens), I(*out_caps), I(), I()
        _chk(lib().jpk_dev_blocks_compress(self._h, n, ins, il, outs, oc, ol, st, int(in_flight)), "jpk_dev_blocks_compress")
        return list(ol), list(st)

    def checksum(self, d_in, in_len) -> int:
        crc = C.c_uint32(0)
        _chk(lib().jpk_dev_checksum(self._h, _dptr(d_in), in_len, C.byref(crc)), "jpk_dev_checksum")
        return crc.value

    def jam_block_write(self, d_in, in_len, block_size, d_out, out_cap) -> int:
        n = C.c_int32(0)
        _chk(lib().jpk_dev_jam_block_write(self._h, _dptr(d_in), in_len, block_size, _dptr(d_out), out_cap, C.byref(n)), "jpk_dev_jam_block_write")
        return n.value

    def jam_block_read(self, d_in, in_len, d_out, out_cap):
        n, used = C.c_int32(0), C.c_int32(0)
        _chk(lib().jpk_dev_jam_block_read(self._h, _dptr(d_in), in_len, _dptr(d_out), out_cap, C.byref(n), C.byref(used)), "jpk_dev_jam_block_read")
        return n.value, used.value

    def rank_encode(self, d_t, d_freq, n):
        _chk(lib().jpk_dev_rank_encode(self._h, _dptr(d_t), _dptr(d_freq), n), "jpk_dev_rank_encode")

    def rank_decode(self, d_r, d_freq, n):
        _chk(lib().jpk_dev_rank_decode(self._h, _dptr(d_r), _dptr(d_freq), n), "jpk_dev_rank_decode")

    def suffix_array(self, d_t, n, d_sa):
        _chk(lib().jpk_dev_suffix_array(self._h, _dptr(d_t), n, _dptr(d_sa)), "jpk_dev_suffix_array")

    def sort_pairs_u64(self, d_keys, d_vals, n, bit_lo=0, bit_hi=64):
        _chk(lib().jpk_dev_sort_pairs_u64(self._h, _dptr(d_keys), _dptr(d_vals), n, bit_lo, bit_hi), "jpk_dev_sort_pairs_u64")

    def exclusive_scan_u32(self, d_data, n) -> int:
        t = C.c_uint32(0)
        _chk(lib().jpk_dev_exclusive_scan_u32(self._h, _dptr(d_data), n, C.byref(t)), "jpk_dev_exclusive_scan_u32")
        return t.value

    def rle_encode(self, d_ranks, n, d_rle) -> int:
        r = C.c_int32(0)
        _chk(lib().jpk_dev_rle_encode(self._h, _dptr(d_ranks), n, _dptr(d_rle), C.byref(r)), "jpk_dev_rle_encode")
        return r.value

    def model_pairs(self, d_rle, rlen, d_pairs):
        _chk(lib().jpk_dev_model_pairs(self._h, _dptr(d_rle), rlen, _dptr(d_pairs)), "jpk_dev_model_pairs")


def multi_plan(device_mask: int, ndev_visible: int, nblocks: int):
    """jpk_debug_multi_plan: (devices taking part, owner device of every block) -- host logic only"""
    own = (C.c_int32 * max(nblocks, 1))()
    g = lib().jpk_debug_multi_plan(device_mask, ndev_visible, nblocks, own)
    _chk(g if g < 0 else 0, "jpk_debug_multi_plan")
    return g, list(own)[:nblocks]


def blocks_compress_multi(blocks, d_out, out_cap: int, device_mask: int = 0, in_flight: int = 0, check: bool = True):
    """jpk_blocks_compress_multi(_ex): host blocks -> compressed blocks gathered in block order into `d_out` (a device buffer on the first
    device of the mask; anything with data_ptr() or an int address), `in_flight` blocks in flight per device (0: the library's default).
    Returns (offsets [nblocks + 1], status [nblocks]); check=False: no exception for a failed block, returns (offsets, status, rc) --
    the blocks whose status is 0 are in d_out whatever happened to the others."""
    arrs = [_np_u8(b) for b in blocks]
    n = len(arrs)
    P, I = C.c_void_p * max(n, 1), C.c_int32 * max(n, 1)
    ins = P(*[a.ctypes.data if len(a) else None for a in arrs])
    lens = I(*[len(a) for a in arrs])
    off = (C.c_int64 * (n + 1))()
    st = I()
    rc = lib().jpk_blocks_compress_multi_ex(device_mask, n, ins, lens, _dptr(d_out), out_cap, off, st, in_flight)
    if not check:
        return list(off), list(st)[:n], int(rc)
    _chk(rc, "jpk_blocks_compress_multi")
    return list(off), list(st)[:n]


def blocks_decompress_multi(comp_blocks, raw_lens, d_out, out_cap: int, device_mask: int = 0, check: bool = True):
    """jpk_blocks_decompress_multi: host compressed blocks (+ the decompressed size of each) -> the blocks' bytes gathered in block order
    into `d_out` on the first device of the mask.  Returns (offsets [nblocks + 1], status [nblocks]); check=False: (offsets, status, rc),
    no exception -- the blocks whose status is 0 are in d_out whatever happened to the others."""
    arrs = [_np_u8(b) for b in comp_blocks]
    n = len(arrs)
    P, I = C.c_void_p * max(n, 1), C.c_int32 * max(n, 1)
    ins = P(*[a.ctypes.data if len(a) else None for a in arrs])
    lens = I(*[len(a) for a in arrs])
    raw = I(*[int(x) for x in raw_lens])
    off = (C.c_int64 * (n + 1))()
    st = I()
    rc = lib().jpk_blocks_decompress_multi(device_mask, n, ins, lens, raw, _dptr(d_out), out_cap, off, st)
    if not check:
        return list(off), list(st)[:n], int(rc)
    _chk(rc, "jpk_blocks_decompress_multi")
    return list(off), list(st)[:n]
