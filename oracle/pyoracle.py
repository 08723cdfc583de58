"""ctypes bindings for the parity checker.  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
jampack_amd/.  `Oracle` wraps oracle/libjamoracle.so (the C restatement, jam_oracle.c); `Ref` wraps
oracle/_ref/libjamref_hot.so / libjamref_cli.so (the real reference compiled from its own sources by oracle/Makefile).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TRAILER = 480
CHUNK = 1 << 20

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)


def _p(a, t):
    return a.ctypes.data_as(t)


def build(ref: bool = True):
    subprocess.check_call(["make", "-C", HERE, "oracle"] + (["ref"] if ref else []), stdout=subprocess.DEVNULL)


class OracleError(RuntimeError):
    pass


class Oracle:
    """The C restatement."""

    def __init__(self, path: str | None = None):
        path = path or os.path.join(HERE, "libjamoracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)

    @staticmethod
    def _chk(rc):
        if rc < 0:
            raise OracleError(f"oracle error {rc}")
        return rc

    def suffix_array(self, t: np.ndarray) -> np.ndarray:
        t = np.ascontiguousarray(t, dtype=np.uint8)
        sa = np.empty(len(t), dtype=np.int32)
        self._chk(self.lib.orc_suffix_array(_p(t, _u8p), C.c_int32(len(t)), _p(sa, _i32p)))
        return sa

    def bwt_forward(self, t: np.ndarray, prefill: int = 0) -> np.ndarray:
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.full(len(t) + TRAILER, prefill, dtype=np.uint8)
        n = C.c_int32(0)
        self._chk(self.lib.orc_bwt_forward(_p(t, _u8p), C.c_int32(len(t)), _p(out, _u8p), C.byref(n)))
        return out[: n.value]

    def bwt_inverse(self, b: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(b, dtype=np.uint8)
        out = np.zeros(max(len(b), 1), dtype=np.uint8)
        n = C.c_int32(0)
        self._chk(self.lib.orc_bwt_inverse(_p(b, _u8p), C.c_int32(len(b)), _p(out, _u8p), C.byref(n)))
        return out[: n.value]

    def sorted_map(self, freq: np.ndarray) -> np.ndarray:
        freq = np.ascontiguousarray(freq, dtype=np.int32)
        m = np.zeros(256, dtype=np.uint8)
        n = self.lib.orc_sorted_map(_p(freq, _i32p), _p(m, _u8p))
        return m[:n]

    def rank_encode(self, t: np.ndarray):
        r = np.array(t, dtype=np.uint8, copy=True)
        f = np.zeros(256, dtype=np.int32)
        self._chk(self.lib.orc_rank_encode(_p(r, _u8p), _p(f, _i32p), C.c_int32(len(r))))
        return r, f

    def rank_decode(self, r: np.ndarray, freq: np.ndarray) -> np.ndarray:
        t = np.array(r, dtype=np.uint8, copy=True)
        f = np.ascontiguousarray(freq, dtype=np.int32)
        self._chk(self.lib.orc_rank_decode(_p(t, _u8p), _p(f, _i32p), C.c_int32(len(t))))
        return t

    def rle_encode(self, r: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(r, dtype=np.uint8)
        out = np.zeros(len(r) + 1, dtype=np.uint16)
        n = self.lib.orc_rle_encode(_p(r, _u8p), _p(out, _u16p), C.c_int32(len(r)))
        return out[:n]

    def rle_decode(self, s: np.ndarray, real_len: int) -> np.ndarray:
        s = np.ascontiguousarray(s, dtype=np.uint16)
        out = np.zeros(real_len + 1, dtype=np.uint8)
        n = self._chk(self.lib.orc_rle_decode(_p(s, _u16p), _p(out, _u8p), C.c_int32(len(s)), C.c_int32(real_len)))
        return out[:n]

    def model_pairs(self, rle: np.ndarray) -> np.ndarray:
        rle = np.ascontiguousarray(rle, dtype=np.uint16)
        pairs = np.zeros(2 * len(rle) + 1, dtype=np.uint32)
        self._chk(self.lib.orc_model_pairs(_p(rle, _u16p), C.c_int32(len(rle)), _p(pairs, _u32p)))
        return pairs[: 2 * len(rle)]

    def rans_encode_pairs(self, pairs: np.ndarray) -> np.ndarray:
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
        out = np.zeros(2 * len(pairs) + 16, dtype=np.uint8)
        n = self._chk(self.lib.orc_rans_encode_pairs(_p(pairs, _u32p), C.c_int32(len(pairs)), _p(out, _u8p), C.c_int32(len(out))))
        return out[:n]

    def rans_decode_chunk(self, payload: np.ndarray, rlen: int) -> np.ndarray:
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        rle = np.zeros(rlen + 1, dtype=np.uint16)
        self._chk(self.lib.orc_rans_decode_chunk(_p(payload, _u8p), C.c_int32(len(payload)), C.c_int32(rlen), _p(rle, _u16p)))
        return rle[:rlen]

    def leb_encode(self, v: int) -> bytes:
        b = np.zeros(8, dtype=np.uint8)
        n = self.lib.orc_leb_encode(C.c_int32(v), _p(b, _u8p))
        return bytes(b[:n])

    def leb_decode(self, b: bytes):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        v = C.c_int32(0)
        n = self._chk(self.lib.orc_leb_decode(C.byref(v), _p(a, _u8p), C.c_int32(len(a))))
        return v.value, n

    def ans_encode(self, b: np.ndarray, cap: int | None = None) -> np.ndarray:
        x = np.array(b, dtype=np.uint8, copy=True)
        cap = cap if cap is not None else int(len(x) * 1.25) + 4096 + 1400 * (len(x) // CHUNK + 1)
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_int32(0)
        self._chk(self.lib.orc_ans_encode(_p(x, _u8p), C.c_int32(len(x)), _p(out, _u8p), C.c_int32(cap), C.byref(n)))
        return out[: n.value]

    def ans_decode(self, c: np.ndarray, cap: int) -> np.ndarray:
        c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.zeros(cap + 1, dtype=np.uint8)
        n = C.c_int32(0)
        self._chk(self.lib.orc_ans_decode(_p(c, _u8p), C.c_int32(len(c)), _p(out, _u8p), C.c_int32(cap), C.byref(n)))
        return out[: n.value]

    def compress_block(self, t: np.ndarray) -> np.ndarray:
        return self.ans_encode(self.bwt_forward(t))

    def decompress_block(self, c: np.ndarray, cap: int) -> np.ndarray:
        return self.bwt_inverse(self.ans_decode(c, cap + TRAILER))

    def checksum(self, t: np.ndarray) -> int:
        t = np.ascontiguousarray(t, dtype=np.uint8)
        self.lib.orc_checksum.restype = C.c_uint32
        return int(self.lib.orc_checksum(_p(t, _u8p), C.c_int32(len(t))))

    def block_header(self, crc: int, comp_size: int, block_size: int) -> bytes:
        b = np.zeros(16, dtype=np.uint8)
        self.lib.orc_block_header(C.c_uint32(crc), C.c_int32(comp_size), C.c_int32(block_size), _p(b, _u8p))
        return bytes(b[:15])


class Ref:
    """The real reference.  `lib` = oracle/_ref/libjamref_hot.so (hot-path translation units, no macro stand-ins);
    `cli` = oracle/_ref/libjamref_cli.so (adds the pre-stages and the Jampack class, built with the __min/__max stand-ins),
    loaded only by the methods that need it.  Errors inside the reference call exit(-1) (format.cpp:6-10)."""

    HOT = os.path.join(HERE, "_ref", "libjamref_hot.so")
    CLI = os.path.join(HERE, "_ref", "libjamref_cli.so")

    def __init__(self, path: str | None = None):
        path = path or self.HOT
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self._cli = None

    @property
    def cli(self):
        if self._cli is None:
            if not os.path.exists(self.CLI):
                raise FileNotFoundError(self.CLI)
            self._cli = C.CDLL(self.CLI)
        return self._cli

    @staticmethod
    def available() -> bool:
        return os.path.exists(Ref.HOT)

    @staticmethod
    def cli_available() -> bool:
        return os.path.exists(Ref.CLI)

    def set_threads(self, n: int):
        """OpenMP team size of the reference (divsufsort.cpp:1493 and the decode loops pick it up)"""
        self.lib.ref_set_threads(C.c_int32(n))

    def bwt_forward(self, t: np.ndarray, prefill: int = 0) -> np.ndarray:
        t = np.array(t, dtype=np.uint8, copy=True)
        out = np.full(len(t) + TRAILER, prefill, dtype=np.uint8)
        n = C.c_int32(0)
        self.lib.ref_bwt_forward(_p(t, _u8p), C.c_int32(len(t)), _p(out, _u8p), C.byref(n))
        return out[: n.value]

    def bwt_inverse(self, b: np.ndarray, threads: int = 1) -> np.ndarray:
        b = np.array(b, dtype=np.uint8, copy=True)
        out = np.zeros(max(len(b), 1), dtype=np.uint8)
        n = C.c_int32(0)
        self.lib.ref_bwt_inverse(_p(b, _u8p), C.c_int32(len(b)), _p(out, _u8p), C.byref(n), C.c_int32(threads))
        return out[: n.value]

    def ans_encode(self, b: np.ndarray) -> np.ndarray:
        x = np.array(b, dtype=np.uint8, copy=True)
        cap = int(len(x) * 1.25) + 4096 + 1400 * (len(x) // CHUNK + 1)
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_int32(0)
        self.lib.ref_ans_encode(_p(x, _u8p), C.c_int32(len(x)), _p(out, _u8p), C.byref(n))
        return out[: n.value]

    def ans_decode(self, c: np.ndarray, cap: int, threads: int = 1) -> np.ndarray:
        c = np.array(c, dtype=np.uint8, copy=True)
        out = np.zeros(cap + 1, dtype=np.uint8)
        n = C.c_int32(0)
        self.lib.ref_ans_decode(_p(c, _u8p), C.c_int32(len(c)), _p(out, _u8p), C.byref(n), C.c_int32(threads))
        return out[: n.value]

    def rank_encode(self, t: np.ndarray):
        r = np.array(t, dtype=np.uint8, copy=True)
        f = np.zeros(256, dtype=np.int32)
        self.lib.ref_rank_encode(_p(r, _u8p), _p(f, _i32p), C.c_int32(len(r)))
        return r, f

    def rank_decode(self, r: np.ndarray, freq: np.ndarray) -> np.ndarray:
        t = np.array(r, dtype=np.uint8, copy=True)
        f = np.array(freq, dtype=np.int32, copy=True)
        self.lib.ref_rank_decode(_p(t, _u8p), _p(f, _i32p), C.c_int32(len(t)))
        return t

    def rle_encode(self, r: np.ndarray) -> np.ndarray:
        r = np.concatenate((np.asarray(r, dtype=np.uint8), np.ones(1, dtype=np.uint8)))  # rle.cpp:31 reads in[len]
        out = np.zeros(len(r) + 1, dtype=np.uint16)
        n = self.lib.ref_rle_encode(_p(r, _u8p), _p(out, _u16p), C.c_int32(len(r) - 1))
        return out[:n]

    def rle_decode(self, s: np.ndarray, real_len: int) -> np.ndarray:
        s = np.concatenate((np.asarray(s, dtype=np.uint16), np.full(1, 2, dtype=np.uint16)))  # rle.cpp:65 reads in[len]
        out = np.zeros(real_len + 1, dtype=np.uint8)
        n = self.lib.ref_rle_decode(_p(s, _u16p), _p(out, _u8p), C.c_int32(len(s) - 1), C.c_int32(real_len))
        return out[:n]

    def leb_encode(self, v: int) -> bytes:
        b = np.zeros(8, dtype=np.uint8)
        n = self.lib.ref_leb_encode(C.c_int32(v), _p(b, _u8p))
        return bytes(b[:n])

    def leb_decode(self, b: bytes):
        a = np.frombuffer(b + b"\x80", dtype=np.uint8).copy()
        v = C.c_int32(0)
        n = self.lib.ref_leb_decode(C.byref(v), _p(a, _u8p))
        return v.value, n

    def checksum(self, t: np.ndarray) -> int:
        t = np.array(t, dtype=np.uint8, copy=True)
        self.lib.ref_checksum.restype = C.c_uint32
        return int(self.lib.ref_checksum(_p(t, _u8p), C.c_int32(len(t))))

    # ---- pre-stages and the whole block codec of the stock CLI ----
    def _stage(self, fn, t, cap, *extra):
        x = np.array(t, dtype=np.uint8, copy=True)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = fn(_p(x, _u8p), C.c_int32(len(x)), _p(out, _u8p), *extra)
        if n < 0:
            raise OracleError(f"reference stage failed ({n})")
        return out[:n]

    def lz77_compress(self, t, match_finder: int = 0, block_size: int = 8 << 20) -> np.ndarray:
        return self._stage(self.cli.ref_lz77_compress, t, int(len(t) * 1.05) + 4096, C.c_int32(match_finder), C.c_int32(block_size))

    def lz77_decompress(self, t, cap: int) -> np.ndarray:
        return self._stage(self.cli.ref_lz77_decompress, t, cap + 64)

    def lpx_encode(self, t) -> np.ndarray:
        return self._stage(self.cli.ref_lpx_encode, t, len(t) + 64)

    def lpx_decode(self, t) -> np.ndarray:
        return self._stage(self.cli.ref_lpx_decode, t, len(t) + 64)

    def filters_encode(self, t, filters: int = 1) -> np.ndarray:
        return self._stage(self.cli.ref_filters_encode, t, int(len(t) * 1.05) + 4096, C.c_int32(filters))

    def filters_decode(self, t, cap: int) -> np.ndarray:
        return self._stage(self.cli.ref_filters_decode, t, cap + 64)

    def jam_comp_block(self, t, block_size: int = 1 << 20, match_finder: int = 0, filters: int = 1) -> np.ndarray:
        """one frame exactly as `jampack c` writes it: Comp() with all six stages + CompWriteBlock"""
        x = np.array(t, dtype=np.uint8, copy=True)
        cap = 15 + int(block_size * 1.05)
        out = np.zeros(cap, dtype=np.uint8)
        n = self.cli.ref_jam_comp_block(_p(x, _u8p), C.c_int32(len(x)), C.c_int32(block_size), C.c_int32(match_finder), C.c_int32(filters),
                                        _p(out, _u8p), C.c_int32(cap))
        if n < 0:
            raise OracleError("reference frame does not fit")
        return out[:n]

    def jam_decomp_block(self, frame, cap: int) -> np.ndarray:
        f = np.array(frame, dtype=np.uint8, copy=True)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = self.cli.ref_jam_decomp_block(_p(f, _u8p), C.c_int32(len(f)), _p(out, _u8p), C.c_int32(cap))
        if n < 0:
            raise OracleError(f"reference block decode failed ({n})")
        return out[:n]

    def divsufsort(self, t: np.ndarray) -> np.ndarray:
        t = np.ascontiguousarray(t, dtype=np.uint8)
        sa = np.zeros(len(t), dtype=np.int32)
        rc = self.lib.ref_divsufsort(_p(t, _u8p), _p(sa, _i32p), C.c_int32(len(t)))
        if rc != 0:
            raise RuntimeError(f"divsufsort rc={rc}")
        return sa

    def compress_block(self, t: np.ndarray) -> np.ndarray:
        return self.ans_encode(self.bwt_forward(t))

    def decompress_block(self, c: np.ndarray, cap: int, threads: int = 1) -> np.ndarray:
        return self.bwt_inverse(self.ans_decode(c, cap + TRAILER, threads), threads)
