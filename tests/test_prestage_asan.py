"""The host pre-stage decoders (jampack_amd/csrc/prestage.cpp) under AddressSanitizer + UBSan, fed with valid golden
streams and with mutated / truncated / random ones: every outcome must be JPK_OK or a negative status, never a memory
error.  CPU build only (g++); the file has no HIP dependency."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, os, sys
import numpy as np
lib = C.CDLL(%(lib)r)
gold = %(gold)r
z = np.load(os.path.join(gold, "golden_cli.npz"))
man = json.load(open(os.path.join(gold, "golden_cli_manifest.json")))
i32 = C.c_int32
def run(stage, enc, cap):
    enc = np.ascontiguousarray(enc, dtype=np.uint8)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = i32(0)
    p = enc.ctypes.data_as(C.c_void_p) if len(enc) else None
    if stage == "lz77":
        rc = lib.jpk_lz77_decompress(p, i32(len(enc)), out.ctypes.data_as(C.c_void_p), i32(cap), C.byref(n))
    elif stage == "filters":
        rc = lib.jpk_filters_decode(p, i32(len(enc)), out.ctypes.data_as(C.c_void_p), i32(cap), C.byref(n))
    else:
        out = np.zeros(max(len(enc), 1), dtype=np.uint8)
        rc = lib.jpk_lpx_decode(p, i32(len(enc)), out.ctypes.data_as(C.c_void_p)); n = i32(len(enc))
    return rc, out[: max(n.value, 0)]
rng = np.random.default_rng(7)
ok = bad = 0
for c in man["stages"]:
    enc = z[c["name"]]
    rc, out = run(c["stage"], enc, c["n"] + 64)
    assert rc == 0 and len(out) == c["n"], (c["name"], rc)
    for trial in range(60):
        m = enc.copy()
        kind = trial %% 4
        if kind == 0:                                   # flip a few bytes
            for _ in range(1 + trial // 8): m[rng.integers(len(m))] ^= np.uint8(1 << rng.integers(8))
        elif kind == 1: m = m[: rng.integers(len(m))]   # truncate
        elif kind == 2: m = np.concatenate([m[: rng.integers(len(m))], rng.integers(0, 256, 64, dtype=np.uint8)])
        else: m = rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8)
        for cap in (c["n"] + 64, 1000, 0):              # roomy, tight and empty output buffers
            rc, _ = run(c["stage"], m, cap)
            assert rc <= 0
            ok += rc == 0; bad += rc < 0
assert lib.jpk_checksum_host(None, i32(0)) == 3
print("asan-ok", ok, bad)
"""


def test_prestage_decoders_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    lib = str(tmp_path / "libprestage_asan.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-DJPK_BUILD", os.path.join(ROOT, "jampack_amd", "csrc", "prestage.cpp"), "-o", lib])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    code = CHILD % {"lib": lib, "gold": os.path.join(ROOT, "tests", "golden")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
