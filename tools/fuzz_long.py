#!/usr/bin/env python3
"""the hostile-stream loop of tests/test_gpu_fuzz.py with many more rounds:  python tools/fuzz_long.py [ans|bwt|jam] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as t
which = sys.argv[1] if len(sys.argv) > 1 else "ans"
rounds = sys.argv[2] if len(sys.argv) > 2 else "1000"
r = subprocess.run(["timeout", "-k", "10", "1500", sys.executable, "-c", t._FUZZ, ROOT, which, rounds], capture_output=True, text=True)
print(r.returncode, r.stdout[-300:], r.stderr[-1500:] if r.returncode else "")
sys.exit(r.returncode)
