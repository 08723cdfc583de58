"""The oracle restatement under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on the
pool).  Runs a child interpreter with libasan preloaded over the edge-case sizes the reference handles."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle.pyoracle import Oracle
from jampack_amd import corpus
o = Oracle(%r)
for kind in ("text", "zero", "random", "geometric", "runs", "two"):
    for n in (0, 1, 119, 120, 121, 1207, 70000, (1 << 20) + 480):
        t = corpus.make(kind, n, 5)
        b = o.bwt_forward(t)
        e = o.ans_encode(b)
        assert np.array_equal(o.ans_decode(e, len(b)), b)
        assert np.array_equal(o.bwt_inverse(b), t)
print("asan-ok")
"""


def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "libjamoracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    # import numpy/corpus without pulling the HIP library into the sanitized child
    code = CHILD % (ROOT, lib)
    code = code.replace("from jampack_amd import corpus", "import importlib.util as u; s = u.spec_from_file_location('corpus', %r); corpus = u.module_from_spec(s); s.loader.exec_module(corpus)" % os.path.join(ROOT, "jampack_amd", "corpus.py"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan-ok" in r.stdout, r.stderr[-3000:]
