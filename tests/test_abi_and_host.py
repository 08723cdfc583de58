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


def test_shim_headers_compile_against_reference_call_pattern(tmp_path):
    """jampack.cpp-style pipeline compiles and links against the shim + C ABI (no GPU needed to link)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "jampack_amd", "csrc", "shim")], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "jampack_amd", "csrc", "shim", "jam_block_pipeline"))


def test_corpus_is_deterministic():
    from jampack_amd import corpus
    a = corpus.make("text", 100000, 8)
    b = corpus.make("text", 100000, 8)
    assert np.array_equal(a, b) and len(a) == 100000
    assert not np.array_equal(a, corpus.make("text", 100000, 9))
    blocks = corpus.split_blocks(np.zeros(100_000_000, dtype=np.uint8), 64 << 20)
    assert [len(x) for x in blocks] == [67108864, 32891136]
