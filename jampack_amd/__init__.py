"""jampack_amd -- MI355X (gfx950) implementation of Jampack's block hot path behind the reference's own
bwt.hpp / ans.hpp / rank.hpp interface.  The compute path is libjampack_amd.so (hand-written HIP kernels);
importing this package loads it and fails loudly if it has not been built."""
from . import corpus  # noqa: F401
from ._lib import ABI_SYMBOLS, CHUNK, LIB_PATH, TRAILER, JampackError, lib  # noqa: F401
from .api import (blocks_compress_multi, blocks_decompress_multi, multi_plan, Ans, Bwt, Checksum, Context, Postcoder, ans_capacity, block_compress, block_decompress,  # noqa: F401
                  jam_block_read, jam_block_write, jam_compress, jam_decompress, Lz77, Lpx, Filters, checksum_host,
                  jam_cli_block_read, jam_cli_decompress, init, shutdown, release_idle, thread_device, ans_decoded_size)

lib()  # no lazy fallback: the HIP extension must be present
