"""helpers shared by the golden-vector tests (CPU oracle tests and GPU parity tests)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def manifest():
    with open(os.path.join(GOLD, "golden_manifest.json")) as f:
        return json.load(f)


_small = None


def small():
    global _small
    if _small is None:
        _small = dict(np.load(os.path.join(GOLD, "golden_small.npz")))
    return _small


def case_input(case):
    from jampack_amd import corpus
    if case["raw"]:
        return small()[case["name"] + ".in"]
    t = corpus.make(case["kind"], case["n"], case["seed"])
    assert sha(t) == case["input_sha256"], "corpus generator drifted from the golden fixtures"
    return t


def cases(raw=None, max_n=None):
    out = []
    for c in manifest()["cases"]:
        if raw is not None and c["raw"] != raw:
            continue
        if max_n is not None and c["n"] > max_n:
            continue
        out.append(c)
    return out
