#!/usr/bin/env python3
"""writes the first 100 MB of this image's source and text files (corpus.system_sources) where JAMPACK_CORPUS_DIR finds it as the enwik8
workload, so that `JAMPACK_CORPUS_DIR=<dir> python bench.py --no-extras` runs the timed loop on REAL files:   python tools/real_files.py <dir>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jampack_amd import corpus
d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/realcorpus"
os.makedirs(d, exist_ok=True)
a = corpus.system_sources(100_000_000)
if a is None:
    sys.exit("the source trees hold less than 100 MB")
a.tofile(os.path.join(d, "enwik8"))
print(f"{len(a)} bytes of real source files -> {d}/enwik8")
