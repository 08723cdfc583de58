#!/bin/bash
# PCIe-inclusive rate of the reference's call pattern through the shim (jam_block_pipeline: 64 MiB blocks, one at a time,
# pageable host buffers, ForwardBwt and Ans::Encode as separate host-buffer calls) on the enwik8-like workload
cd "$(dirname "$0")/.."
python - <<'PY'
import sys
sys.path.insert(0, '.')
from jampack_amd import corpus
d, _ = corpus.load_or_make("enwik9", start=0, count=4 * (64 << 20))
d.tofile("/tmp/pipe_in.bin")
PY
make -C jampack_amd/csrc/shim > /dev/null
for i in 1 2; do jampack_amd/csrc/shim/jam_block_pipeline /tmp/pipe_in.bin 64; done
# the same file the way the multi-block loop of jampack.cpp drives it: 4 threads, each with its own Pipeline (8 blocks: the file twice as long)
python - <<'PY'
import sys
sys.path.insert(0, ".")
from jampack_amd import corpus
d, _ = corpus.load_or_make("enwik9", start=0, count=8 * (64 << 20))
d.tofile("/tmp/pipe_in8.bin")
PY
jampack_amd/csrc/shim/jam_block_pipeline /tmp/pipe_in8.bin 64 4 | tail -1
jampack_amd/csrc/shim/jam_block_pipeline /tmp/pipe_in8.bin 64 8 | tail -1
