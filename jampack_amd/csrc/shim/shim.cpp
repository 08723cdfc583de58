// shim.cpp -- forwards the reference's stage classes to the C ABI (include/jampack_abi.h) and maps a non-zero
// status back to Error(), which is how every reference stage reports failure (format.cpp:6-10).
#include <stdio.h>

#include "../../../include/jampack_abi.h"
#include "ans.hpp"
#include "bwt.hpp"
#include "rank.hpp"

static void fail(const char *where, int rc)
{
	char msg[160];
	snprintf(msg, sizeof msg, "%s :: %s", where, jpk_strerror(rc));
	Error(msg);
}

// stage buffers are allocated as int(BlockSize * 1.05) by the caller (jampack.cpp:74-76, 157-159)
static int stage_capacity(const Options &Opt) { return (int)((double)Opt.BlockSize * 1.05); }

void BlockSort::Bwt::ForwardBwt(Buffer Input, Buffer Output)
{
	int n = 0;
	int rc = jpk_bwt_forward(Input.block, *Input.size, Output.block, *Input.size + JPK_TRAILER_BYTES, &n);
	if (rc) fail("Bwt", rc);
	*Output.size = n;
}

void BlockSort::Bwt::InverseBwt(Buffer Input, Buffer Output, Options Opt)
{
	int n = 0;
	int rc = jpk_bwt_inverse(Input.block, *Input.size, Output.block, *Input.size - JPK_TRAILER_BYTES, &n, (int)Opt.Threads, Opt.Gpu ? 1 : 0);
	if (rc) fail("Bwt", rc);
	*Input.size -= JPK_TRAILER_BYTES;      // the reference shrinks the caller's input size (bwt.cpp:77)
	*Output.size = n;
}

void Ans::Encode(Buffer Input, Buffer Output, Options Opt)
{
	int n = 0;
	int rc = jpk_ans_encode(Input.block, *Input.size, Output.block, stage_capacity(Opt), &n);
	if (rc) fail("Ans", rc);
	*Output.size = n;
}

void Ans::Decode(Buffer Input, Buffer Output, Options Opt)
{
	// Not stage_capacity(Opt): on the decompress path Options.BlockSize is the CLI option (default 8 MiB, main.cpp:60),
	// while the buffers were sized from the frame header (Jampack::BlockSize, jampack.cpp:146-159).  Like the reference
	// the shim trusts that the caller's buffer holds what the stream declares; the declared size itself is validated.
	int64_t need = 0;
	int rc = jpk_ans_decoded_size(Input.block, *Input.size, &need, 0);
	if (rc) fail("Ans", rc);
	if (need > (int64_t)((double)JPK_MAX_BLOCKSIZE * 1.05)) fail("Ans", JPK_E_CORRUPT);
	int n = 0;
	rc = jpk_ans_decode(Input.block, *Input.size, Output.block, (int)need, &n, (int)Opt.Threads);
	if (rc) fail("Ans", rc);
	*Output.size = n;
}

void Postcoder::Encode(unsigned char *T, int *Freq, int len)
{
	int rc = jpk_rank_encode(T, Freq, len);
	if (rc) fail("Postcoder", rc);
}

void Postcoder::Decode(unsigned char *RankArray, int *Freq, int len)
{
	int rc = jpk_rank_decode(RankArray, Freq, len);
	if (rc) fail("Postcoder", rc);
}
