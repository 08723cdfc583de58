"""Generates tests/golden/golden_cli.npz + golden_cli_manifest.json from the REAL reference (oracle/_ref/libjamref.so):
  * pre-stage streams: Lz77::Compress / Lpx::Encode / Filters::Encode outputs of seeded corpus inputs
  * whole frames exactly as `jampack c` writes them (Jampack::Comp with all six stages + CompWriteBlock)
Run in the build container only:  python tests/golden/make_golden_cli.py
Fixtures are data: the inputs are regenerated from (kind, n, seed); only the reference's outputs are stored."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from jampack_amd import corpus  # noqa: E402
from oracle.pyoracle import Ref  # noqa: E402

STAGES = [("lz77", "repeat4k", 150_000, 61, 0), ("lz77", "text", 40_000, 62, 1), ("lz77", "runs", 60_000, 63, 8),
          ("lpx", "text", 120_000, 64, 0), ("lpx", "samples16", 50_001, 65, 0), ("lpx", "repeat4k", 70_003, 66, 0),
          ("filters", "samples16", 150_000, 67, 1), ("filters", "silesia", 140_000, 68, 1), ("filters", "geometric", 66_000, 69, 2)]
FRAMES = [("text", 250_000, 71, 0, 1), ("silesia", 300_000, 72, 0, 1), ("samples16", 140_000, 73, 0, 1), ("repeat4k", 200_000, 74, 0, 1),
          ("text", 60_000, 75, 1, 1), ("samples16", 50_000, 76, 8, 2), ("zero", 5_000, 77, 0, 1), ("random", 70_000, 78, 0, 1)]


def main():
    r = Ref()
    arrays, man = {}, {"stages": [], "frames": []}
    for stage, kind, n, seed, arg in STAGES:
        t = corpus.make(kind, n, seed)
        enc = {"lz77": lambda: r.lz77_compress(t, arg), "lpx": lambda: r.lpx_encode(t), "filters": lambda: r.filters_encode(t, arg)}[stage]()
        name = f"{stage}_{kind}_{n}_{seed}"
        arrays[name] = enc
        man["stages"].append({"name": name, "stage": stage, "kind": kind, "n": n, "seed": seed, "arg": arg, "enc_len": int(len(enc))})
    for kind, n, seed, mf, fl in FRAMES:
        t = corpus.make(kind, n, seed)
        frame = r.jam_comp_block(t, 1 << 20, mf, fl)
        assert np.array_equal(r.jam_decomp_block(frame, 1 << 20), t)
        name = f"frame_{kind}_{n}_{seed}"
        arrays[name] = frame
        man["frames"].append({"name": name, "kind": kind, "n": n, "seed": seed, "match_finder": mf, "filters": fl, "block_size": 1 << 20,
                              "frame_len": int(len(frame)), "crc": r.checksum(t)})
    # a two-block stream, second block short
    t = corpus.make("text", (1 << 20) + 77_777, 79)
    stream = np.concatenate([r.jam_comp_block(t[: 1 << 20], 1 << 20, 0, 1), r.jam_comp_block(t[1 << 20:], 1 << 20, 0, 1)])
    arrays["stream_text_2blocks"] = stream
    man["stream"] = {"name": "stream_text_2blocks", "kind": "text", "n": int(len(t)), "seed": 79, "block_size": 1 << 20}
    np.savez_compressed(os.path.join(HERE, "golden_cli.npz"), **arrays)
    json.dump(man, open(os.path.join(HERE, "golden_cli_manifest.json"), "w"), indent=1)
    print("wrote", len(arrays), "arrays,", sum(len(a) for a in arrays.values()), "bytes raw")


if __name__ == "__main__":
    main()
