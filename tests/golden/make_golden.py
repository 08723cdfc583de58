"""Generates tests/golden/ from the REAL reference (oracle/_ref/libjamref.so, built from /root/reference
by oracle/Makefile).  Run in the build container only:  python tests/golden/make_golden.py

Fixtures are data: inputs (or their generator triple) and the reference's outputs.
  golden_small.npz     raw input / ForwardBwt / Ans::Encode / Postcoder::Encode / RLE bytes for inputs <= 64 KiB
  golden_manifest.json sha256 of every stage output for all cases (incl. 1 MB .. 3 MB inputs)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from jampack_amd import corpus  # noqa: E402
from oracle.pyoracle import Ref, TRAILER  # noqa: E402

SMALL = [(k, n, 11) for k in ("text", "random", "dna", "two", "zero", "geometric", "runs", "repeat4k")
         for n in (0, 1, 119, 120, 121, 1207, 65536)]
LARGE = [("text", 1_000_000, 6), ("text", (1 << 20) + 480, 7), ("text", 3_000_000, 8), ("silesia", 2_500_000, 5),
         ("repeat", 2_400_000, 9), ("geometric", 1_500_000, 3), ("random", 1_100_000, 4), ("zero", 1_300_000, 1),
         ("samples16", 1_200_000, 2)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    r = Ref()
    small = {}
    manifest = {"trailer_prefill": 0xAB, "cases": []}
    for kind, n, seed in SMALL + LARGE:
        t = corpus.make(kind, n, seed)
        bwt = r.bwt_forward(t, prefill=0xAB)
        ans = r.ans_encode(bwt)
        back = r.bwt_inverse(r.ans_decode(ans, len(bwt)))
        assert np.array_equal(back, t)
        chunk0 = bwt[: 1 << 20]
        ranks, freq = r.rank_encode(chunk0)
        rle = r.rle_encode(ranks)
        name = f"{kind}_{n}_{seed}"
        manifest["cases"].append({
            "name": name, "kind": kind, "n": n, "seed": seed, "input_sha256": sha(t), "bwt_sha256": sha(bwt),
            "bwt_len": int(len(bwt)), "ans_sha256": sha(ans), "ans_len": int(len(ans)),
            "rank0_sha256": sha(ranks), "freq0_sha256": sha(freq.astype("<i4")), "rle0_sha256": sha(rle.astype("<u2")),
            "rle0_len": int(len(rle)), "raw": n <= 65536, "crc": r.checksum(t)})
        if n <= 65536:
            small[name + ".in"] = t
            small[name + ".bwt"] = bwt
            small[name + ".ans"] = ans
            if n in (1207, 65536):
                small[name + ".rank"] = ranks
                small[name + ".freq"] = freq.astype("<i4")
                small[name + ".rle"] = rle.astype("<u2")
    leb = {}
    for v in (0, 1, 126, 127, 128, 16509, 16510, 16511, 2113660, 2113661, 2113662, 270549115, 270549116, 1 << 20, 2147483647):
        leb[str(v)] = r.leb_encode(v).hex()
    manifest["leb128"] = leb
    # Checksum::IntegrityCheck at the loop boundaries of checksum.cpp:18-33 (j + 16 < size) and odd lengths
    manifest["checksum"] = [{"kind": k, "n": n, "seed": 5, "crc": r.checksum(corpus.make(k, n, 5))}
                            for k in ("text", "random", "zero") for n in (0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 4095, 65551, 1048577, 5000011)]
    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **small)
    with open(os.path.join(HERE, "golden_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", len(manifest["cases"]), "cases")


if __name__ == "__main__":
    main()
