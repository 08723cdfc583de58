import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """the .so files are git-ignored build products: build them when a fresh checkout runs the tests"""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "jampack_amd", "libjampack_amd.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "jampack_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libjamoracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"], stdout=subprocess.DEVNULL)


_ensure_built()

# torch bundles its own HIP runtime: let it initialise first so that libjampack_amd.so (linked against the system
# ROCm) and torch share one runtime instance in the GPU tests that use both (torch tensors as device buffers)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
try:
    import torch
    torch.cuda.is_available()
except Exception:  # pragma: no cover - torch is optional for the CPU-only oracle tests
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libjamref.so not built (reference tree absent)")
    return Ref()
