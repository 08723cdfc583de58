"""CPU-side checks: the C ABI library loads and exports every symbol include/jampack_abi.h declares, the
product path refuses to run without a GPU (no fallback), host-side helpers.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "jampack_abi.h")).read()
    return sorted(set(re.findall(r"JPK_API[^;(]*?\b(jpk_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import jampack_amd
    lib = ctypes.CDLL(jampack_amd.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/jampack_abi.h but not exported"


def test_python_binding_covers_the_abi():
    import jampack_amd
    from jampack_amd._lib import _SIGS
    missing = set(_declared_symbols()) - set(_SIGS)
    assert not missing, f"ctypes signatures missing for {missing}"


def test_no_cpu_fallback_without_gpu():
    import jampack_amd
    if jampack_amd.lib().jpk_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(jampack_amd.JampackError) as e:
        jampack_amd.Bwt().ForwardBwt(np.zeros(1000, dtype=np.uint8))
    assert e.value.status == -6          # JPK_E_NODEVICE: the product path never routes through the oracle
    with pytest.raises(jampack_amd.JampackError):
        jampack_amd.Context(0)


def test_product_code_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "jampack_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in src and "jam_oracle" not in src and "libjamref" not in src, f"{f} references oracle/"


def test_argument_validation():
    import jampack_amd
    from jampack_amd import lib
    n = ctypes.c_int32(0)
    buf = (ctypes.c_uint8 * 16)()
    assert lib().jpk_bwt_forward(buf, -1, buf, 16, ctypes.byref(n)) in (-1, -6)
    assert lib().jpk_bwt_forward(buf, 8, buf, 16, ctypes.byref(n)) == -2     # capacity: needs len + 480
    assert lib().jpk_ctx_create(None, 0, None) == -1
    assert lib().jpk_strerror(-3).decode().startswith("corrupt")
    # the batch entries refuse a missing context before anything else (no GPU call is made)
    for name in ("jpk_dev_blocks_ans_decode", "jpk_dev_blocks_decompress"):
        assert getattr(lib(), name)(None, 0, None, None, None, None, None, None) == -1
    assert lib().jpk_dev_blocks_compress(None, 0, None, None, None, None, None, None, 4) == -1


def test_shim_headers_compile_against_reference_call_pattern(tmp_path):
    """jampack.cpp-style pipeline compiles and links against the shim + C ABI (no GPU needed to link)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "jampack_amd", "csrc", "shim")], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "jampack_amd", "csrc", "shim", "jam_block_pipeline"))


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree absent (GPU box)")
def test_integration_md_build_recipe_links_the_real_reference(tmp_path):
    """Performs INTEGRATION.md section 1 literally: a scratch copy of the reference tree, its bwt.hpp / ans.hpp /
    rank.hpp replaced by the shim headers, built with the documented g++ line (read from the document, so the two cannot
    drift apart).  The unmodified main.cpp + jampack.cpp + pre-stages must link against shim.cpp + libjampack_amd.so."""
    import glob
    import shutil
    import subprocess
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```\n\s*(g\+\+ -std=c\+\+14.*?)```", doc, re.S)
    assert m, "INTEGRATION.md lost its build line"
    cmdline = " ".join(m.group(1).replace("\\\n", " ").split())
    assert "divsufsort.cpp" in cmdline, "lz77.cpp:141 calls divsufsort(): the recipe must keep divsufsort.cpp"
    for f in glob.glob(os.path.join(REFERENCE, "*.[ch]pp")):
        shutil.copy(f, tmp_path)
    shim = os.path.join(ROOT, "jampack_amd", "csrc", "shim")
    for h in ("bwt.hpp", "ans.hpp", "rank.hpp"):
        shutil.copy(os.path.join(shim, h), tmp_path)
    os.symlink(os.path.join(ROOT, "jampack_amd"), tmp_path / "jampack_amd")
    os.symlink(os.path.join(ROOT, "include"), tmp_path / "include")
    r = subprocess.run(cmdline + " -o jampack_gpu", shell=True, cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run(["nm", "-D", "--undefined-only", str(tmp_path / "jampack_gpu")], capture_output=True, text=True).stdout
    for sym in ("jpk_bwt_forward", "jpk_bwt_inverse", "jpk_ans_encode", "jpk_ans_decode", "jpk_ans_decoded_size"):
        assert sym in out, f"the drop-in binary does not bind {sym}"
    assert "divsufsort" not in out            # defined in the binary, from the reference's own divsufsort.cpp


def test_device_mask_api_without_gpu():
    import jampack_amd
    if jampack_amd.lib().jpk_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert jampack_amd.lib().jpk_init(0) == -6           # JPK_E_NODEVICE, never a silent CPU path
    assert jampack_amd.lib().jpk_thread_device() == -6
    jampack_amd.lib().jpk_shutdown()                     # harmless with nothing to destroy
    assert jampack_amd.lib().jpk_release_idle() == 0     # nothing idle either


def test_bench_refuses_a_gpus_flag_that_disagrees_with_the_launcher():
    """`--gpus N` must never silently run one GPU"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_survey_corpus_is_seekable():
    from jampack_amd import corpus
    whole = corpus.text_survey(9_000_000, 9)
    part = corpus.text_survey(2_500_000, 9, start=3_999_000)         # straddles a page boundary
    assert np.array_equal(part, whole[3_999_000:6_499_000])
    d, src = corpus.load_or_make("enwik9", start=14 * (64 << 20), count=64 << 20)
    assert len(d) == 60_475_904 and src == "synthetic"               # the 15th block of the 1 GB stream
    assert corpus.block_ranges(1_000_000_000, 64 << 20)[-1] == (939_524_096, 60_475_904)


def test_corpus_is_deterministic():
    from jampack_amd import corpus
    a = corpus.make("text", 100000, 8)
    b = corpus.make("text", 100000, 8)
    assert np.array_equal(a, b) and len(a) == 100000
    assert not np.array_equal(a, corpus.make("text", 100000, 9))
    blocks = corpus.split_blocks(np.zeros(100_000_000, dtype=np.uint8), 64 << 20)
    assert [len(x) for x in blocks] == [67108864, 32891136]


def test_encoder_launch_grouping_follows_the_load_of_its_own_device():
    """the blocks-in-flight count that drives the encoder's launch grouping is per DEVICE.  Faked two-device
    load through the host-logic probes (no device call): three blocks on device 0 leave a block on device 1 alone."""
    from jampack_amd import lib
    l = lib()
    nch = 65                                   # a 64 MiB block: 64 full chunks + the trailer chunk
    assert l.jpk_debug_compress_inflight(0, 0) == 0 and l.jpk_debug_compress_inflight(1, 0) == 0
    # the hooks that change live state are refused unless the process asks for them
    os.environ.pop("JPK_DEBUG_HOOKS", None)
    assert l.jpk_debug_compress_inflight(0, 1) == -1 and l.jpk_debug_compress_inflight(0, 0) == 0 and l.jpk_debug_combiner_fail_next(1) == -1
    os.environ["JPK_DEBUG_HOOKS"] = "1"
    assert l.jpk_debug_enc_groups(0, nch) == 4 and l.jpk_debug_enc_groups(1, nch) == 4        # alone: four graded groups
    try:
        assert [l.jpk_debug_compress_inflight(0, 1) for _ in range(3)] == [1, 2, 3]
        assert l.jpk_debug_enc_groups(0, nch) == 1           # a fourth block on device 0: the machine is full, one group
        assert l.jpk_debug_enc_groups(1, nch) == 4           # ... while a block on device 1 is still alone on ITS GPU
        assert l.jpk_debug_compress_inflight(1, 1) == 1
        assert l.jpk_debug_enc_groups(1, nch) == 2           # beside one other block: two groups
        assert l.jpk_debug_compress_inflight(1, -1) == 0
    finally:
        while l.jpk_debug_compress_inflight(0, 0) > 0:
            l.jpk_debug_compress_inflight(0, -1)
    assert l.jpk_debug_enc_groups(0, 7) == 1                 # fewer than 8 chunks: never split
    assert l.jpk_debug_compress_inflight(64, 1) == -1 and l.jpk_debug_enc_groups(-1, nch) == -1


def test_reserve_follows_the_stage_layouts():
    """jpk_ctx_reserve takes the maximum of the four stages' own planning passes (DESIGN.md section 3 quotes these factors)"""
    from jampack_amd import lib
    n = 64 << 20
    per_byte = [lib().jpk_debug_arena_bytes(n, st) / n for st in range(5)]
    assert 53 < per_byte[0] < 56          # forward BWT: radix ping-pong 24 n + ISA 4 n + active list 8 n + BWT bytes n + carried BWT bytes 3 n + run lengths 4 n + tables
                                          # + the groups' depths 2 x 4 n (variable-length keys; + n above 2^28 bytes, where the slots' depths keep a buffer of their own; 46.3 n with fixed-width keys)
    assert 40 < per_byte[1] < 47          # rANS encode sized for text (0.55 RLE0 symbols per byte at ~75 bytes per symbol + ranks n + RLE0 symbols 2 n)
    assert 9 < per_byte[2] < 11           # inverse BWT
    assert 3 <= per_byte[3] < 3.2         # rANS decode bound
    assert 68 < per_byte[4] < 78          # rANS encode, densest data (every byte a symbol): records 32 n, states' low halves 4 n + emit masks 0.5 n (round 6; 8 n + a 4 n frequency sidecar before), exponent histories 14 n, ...
    assert lib().jpk_debug_arena_bytes(n, 5) == -1 and lib().jpk_debug_arena_bytes(-1, 0) == -1


def test_decoded_size_rejects_more_rle_symbols_than_bytes():
    """RLE0 never has more symbols than bytes; a header that claims so is refused before any buffer is sized
    by it (the reference ends such a chunk in "rle mismatch!", rle.cpp:73)."""
    import jampack_amd
    from jampack_amd.api import ans_decoded_size

    def leb(v):
        for nb, c in ((1, 0), (2, 127), (3, 16510), (4, 2113661)):
            if v < (127, 16510, 2113661, 270549116)[nb - 1]:
                v -= c
                b = [(v >> (7 * (nb - 1 - k))) & 0x7F for k in range(nb)]
                b[-1] |= 0x80
                return bytes(b)
        raise ValueError(v)

    def chunk(olen, clen, rlen):
        freq = [olen] + [0] * 255
        return b"".join(leb(f) for f in freq) + leb(olen) + leb(clen) + leb(rlen) + bytes(clen)

    ok = np.frombuffer(chunk(1, 16, 1), dtype=np.uint8)
    assert ans_decoded_size(ok) == (1, 1)
    bad = np.frombuffer(chunk(1, 16, 1 << 20), dtype=np.uint8)
    with pytest.raises(jampack_amd.JampackError) as e:
        ans_decoded_size(bad)
    assert e.value.status == -3


def test_multi_device_calls_on_disjoint_device_sets_run_side_by_side():
    """the multi-device entries hold the mutexes of the devices of THEIR mask (ascending), not one process-wide mutex:
    two files on two halves of a node compress concurrently, calls that share a device queue.  Host logic through the probe that takes
    exactly those mutexes and holds them (no device call)."""
    import threading
    import time
    from jampack_amd import lib

    def span(masks, hold_ms=300):
        t0 = time.perf_counter()
        th = [threading.Thread(target=lib().jpk_debug_multi_lock_probe, args=(m, hold_ms)) for m in masks]
        [t.start() for t in th]
        [t.join() for t in th]
        return time.perf_counter() - t0

    assert lib().jpk_debug_multi_lock_probe(0b10110, 0) == 3
    assert span([0b00001111, 0b11110000]) < 0.5               # disjoint halves of an 8-GPU node: side by side
    assert span([0b0011, 0b1100, 0b110000]) < 0.5
    assert span([0b00001111, 0b00011000]) >= 0.58             # they share device 3: one after the other
    assert span([0b1, 0b1, 0b1]) >= 0.88
    assert span([0b0101, 0b1010, 0b0110]) >= 0.58             # the third needs a device of each of the first two (no deadlock: ascending order)
    assert lib().jpk_debug_multi_lock_probe(1, -1) == -1


def test_multi_device_ownership_and_order():
    """jpk_blocks_compress_multi's plan (host logic): block b on the (b mod G)-th device of the mask, the first device is the root;
    a mask that names no visible device is an error; the header declares the entry points and the library exports them"""
    import subprocess
    from jampack_amd import api, lib
    g, own = api.multi_plan(0, 8, 15)                     # BASELINE config 4: 15 blocks over 8 GPUs
    assert g == 8 and own == [b % 8 for b in range(15)]
    g, own = api.multi_plan(0b10100100, 8, 7)             # devices 2, 5, 7
    assert g == 3 and own == [2, 5, 7, 2, 5, 7, 2]
    g, own = api.multi_plan(0b1, 1, 4)
    assert g == 1 and own == [0, 0, 0, 0]
    assert lib().jpk_debug_multi_plan(0b100, 2, 3, (ctypes.c_int32 * 3)()) == -6        # device 2 of 2 visible: none
    assert lib().jpk_debug_multi_plan(0, 8, -1, None) == -1
    # every rank's blocks come back in block order: the offsets of block b follow those of block b - 1 whatever device ran it
    hdr = open(os.path.join(ROOT, "include", "jampack_abi.h")).read()
    assert "jpk_blocks_compress_multi" in hdr and "jpk_debug_multi_plan" in hdr
    # no link-time dependency on RCCL: it is loaded at first use
    out = subprocess.run(["ldd", os.path.join(ROOT, "jampack_amd", "libjampack_amd.so")], capture_output=True, text=True).stdout
    assert "rccl" not in out


def test_pair_round_schedule_on_recorded_lists():
    """which rounds of a suffix sort are PAIR rounds (bwt_fwd.hip PairSchedule, host logic): a round from the third on behind a doubling round that
    left >= 90 % of its list, two doubling rounds apart -- and six, then fourteen, behind a pair round that left more than half of its list; round 2
    already when round 1 left 99 %.  The lists are the ones the GPU recorded (profiles/r06_real_files_pair_rounds.txt, tools/pair_yield.py): the
    schedule must pick the rounds it picked there, keep the rounds that resolve exact repeats, and stay away from real near-duplicate trees."""
    from jampack_amd import lib

    def pairs(n, lst, runs_heavy=0):
        L = (ctypes.c_uint32 * len(lst))(*lst)
        out = (ctypes.c_int32 * len(lst))()
        k = lib().jpk_debug_pair_schedule(n, len(lst), L, runs_heavy, out)
        got = [r for r in range(len(lst)) if out[r]]
        assert k == len(got)
        return got

    n = 67108800
    # real source trees: every round leaves 60-85 % of its list -- no pair round (with the round-5 threshold of 60 %: rounds 3, 6, 9, 12 ms wasted)
    assert pairs(n, [n, 58502091, 48754810, 40065036, 32193135, 24239432, 15533282, 10290246, 6549112, 4124712, 2326215, 1313193, 777774, 390726, 249406, 20642]) == []
    # the headline text: its list shrinks 2.5x and 500x -- never
    assert pairs(n, [n, 27281026, 48909]) == []
    # a block that holds a text twice: round 1 leaves everything, round 2 is the pair round that resolves it
    assert pairs(33554400, [33554400, 33554306, 33554306, 40764, 1722, 20]) == [2]
    # config 5 / the silesia-like mix (a list shaped like the recorded one): text and samples resolve, the repeated segment is what round 4 leaves as it was
    assert pairs(211938480, [211938480, 110000000, 40000000, 26000000, 21400000, 21193563]) == [5]
    # a 1 MiB period: the early pair round leaves 16 %, two doubling rounds, nothing left for another
    assert pairs(n, [n, 67108772, 67108772, 10964920, 3100000, 400000]) == [2]
    # real shared libraries twice: pair rounds 3 and 10 (the first leaves 93 %: the next one waits six rounds and resolves 89 %), then 13
    # (zero padding: the block counts as runs-heavy, so no early pair round at 2)
    assert pairs(n, [n, 67108695, 67108695, 67108695, 62202089, 62100000, 62000000, 61900000, 61850000, 61800000, 61745666, 6628360, 6500000, 6319522], runs_heavy=1) == [3, 10, 13]
    # the Fibonacci word: pair rounds resolve nothing -- 2, then 9, then not before 24 (nine of them, every third round, until the end of round 6)
    fib = [33554400] + [33554289] * 9 + [33547233] * 3 + [33525729] * 3 + [33095649, 31719393, 26214369, 15000000, 4000000, 100000]
    assert pairs(33554400, fib) == [2, 9]
    # mostly runs: the run rule is splitting those groups, no early pair round
    assert pairs(n, [n, 67000000, 66900000], runs_heavy=1) == []
    assert lib().jpk_debug_pair_schedule(0, 3, (ctypes.c_uint32 * 3)(), 0, (ctypes.c_int32 * 3)()) == -1


def test_group_plan_covers_every_block_once_and_in_order():
    """jpk_dev_blocks_compress' work list (host logic): small blocks (<= 16 MiB) in groups of consecutive blocks -- at most 256 blocks, at
    most the target size beyond the first block (a quarter of the small blocks' bytes, 8..64 MiB) --, a larger block alone"""
    from jampack_amd import lib
    MiB = 1 << 20

    def plan(lens):
        n = len(lens)
        L = (ctypes.c_int32 * max(n, 1))(*lens)
        F = (ctypes.c_int32 * max(n, 1))()
        K = (ctypes.c_int32 * max(n, 1))()
        t = lib().jpk_debug_group_plan(n, L, F, K)
        assert t >= 0
        return [(F[i], K[i]) for i in range(t)]

    def check(lens, tasks):
        nxt = 0
        for f, k in tasks:
            assert f == nxt and k >= 1
            nxt += k
            if k > 1:
                assert k <= 256 and all(lens[i] <= 16 * MiB for i in range(f, f + k))
            if any(lens[i] > 16 * MiB for i in range(f, f + k)):
                assert k == 1
        assert nxt == len(lens)

    assert plan([]) == []
    for lens in ([MiB] * 256, [8 * MiB] * 32, [64 * MiB] * 4, [0, 1, 119, 120, 121, MiB - 1, MiB, 8 * MiB, 17 * MiB, 5, 3 * MiB + 61], [300] * 1000, [16 * MiB, 16 * MiB + 1] * 5):
        t = plan(lens)
        check(lens, t)
    assert plan([MiB] * 256) == [(0, 64), (64, 64), (128, 64), (192, 64)]           # 256 MiB of 1 MiB blocks: four groups of 64 MiB
    assert plan([8 * MiB] * 32) == [(8 * i, 8) for i in range(4)]
    assert plan([MiB] * 64) == [(16 * i, 16) for i in range(4)]                       # a 64 MiB stream: a quarter each
    assert plan([64 * MiB] * 4) == [(i, 1) for i in range(4)]                         # large blocks go alone
    assert all(k <= 256 for _, k in plan([300] * 1000))
    assert lib().jpk_debug_group_plan(2, (ctypes.c_int32 * 2)(5, -1), (ctypes.c_int32 * 2)(), (ctypes.c_int32 * 2)()) == -1


def test_stats_struct_layout_matches_the_header(tmp_path):
    """jpk_stats as the C compiler lays it out from include/jampack_abi.h against the ctypes mirror (jampack_amd/_lib.py): size and the
    offset of every field -- the bench reads the rounds, the key depth of round 0 and the arena size through it"""
    import subprocess
    from jampack_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in _lib.Stats._fields_]
    src = tmp_path / "layout.c"
    body = "".join(f'    printf("{f} %zu\\n", offsetof(jpk_stats, {f}));\n' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "jampack_abi.h"\nint main(void)\n{\n    printf("sizeof %zu\\n", sizeof(jpk_stats));\n' + body + "    return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(out["sizeof"]) == ctypes.sizeof(_lib.Stats)
    for f in fields:
        assert int(out[f]) == getattr(_lib.Stats, f).offset, f
    assert "sa_key_depth" in fields
