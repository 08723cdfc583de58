// shim.cpp -- forwards the reference's stage classes to the C ABI (include/jampack_abi.h) and maps a non-zero
// status back to Error(), which is how every reference stage reports failure (format.cpp:6-10).
#include <malloc.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>

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

// Diagnostics for a maintainer who diffs a patched build against the stock one, compiled in with -DJPK_SHIM_DIAG only
// (tools/cli_dump.sh, tools/cli_guard.sh build such a shim); the product shim has none of it on its call path.
#ifdef JPK_SHIM_DIAG
// JPK_SHIM_TRACE=1: one line per stage call with the block checksum (checksum.cpp:12-36) of its input and output -- lets a
// maintainer diff a run of the patched program against the stock one stage by stage
static void trace(const char *stage, const unsigned char *in, int in_len, const unsigned char *out, int out_len)
{
	static const bool on = getenv("JPK_SHIM_TRACE") != NULL;
	if (on) fprintf(stderr, "[shim] %-10s in %9d %08x  out %9d %08x\n", stage, in_len, jpk_checksum_host(in, in_len), out_len, jpk_checksum_host(out, out_len));
	static const char *dump = getenv("JPK_SHIM_DUMP");          // directory: every stage input as <n>_<stage>.bin
	if (dump) {
		static int seq = 0;
		char path[512];
		snprintf(path, sizeof path, "%s/%03d_%s.bin", dump, seq++, stage);
		if (FILE *f = fopen(path, "wb")) { fwrite(in, 1, (size_t)in_len, f); fclose(f); }
	}
}

// JPK_SHIM_GUARD=<capacity bytes>: verifies that a stage call changes nothing but Output[0, n): the rest of Output up to the
// capacity and all of Input keep their bytes, and Output[0, n) does not change after the call has returned
struct Guard {
	long cap;
	std::vector<unsigned char> in0, out0;
	const unsigned char *in, *out;
	int in_len;
	const char *stage;
	Guard(const char *st, const unsigned char *i, int il, const unsigned char *o) : in(i), out(o), in_len(il), stage(st)
	{
		const char *e = getenv("JPK_SHIM_GUARD");
		cap = e ? atol(e) : 0;
		if (cap) { in0.assign(i, i + cap); out0.assign(o, o + cap); }
	}
	void check(int n, bool input_may_change)
	{
		if (!cap) return;
		if (!input_may_change && memcmp(in0.data(), in, (size_t)cap) != 0) fprintf(stderr, "[guard] %s modified its INPUT buffer\n", stage);
		if (memcmp(out0.data() + n, out + n, (size_t)(cap - n)) != 0) {
			long k = n; while (k < cap && out0[k] == out[k]) k++;
			fprintf(stderr, "[guard] %s wrote past its output length %d: first changed byte at %ld\n", stage, n, k);
		}
		const unsigned int c0 = jpk_checksum_host(out, n);
		usleep(3000);
		if (jpk_checksum_host(out, n) != c0) fprintf(stderr, "[guard] %s: output changed AFTER the call returned\n", stage);
	}
};

#else
static inline void trace(const char *, const unsigned char *, int, const unsigned char *, int) {}
struct Guard {
	Guard(const char *, const unsigned char *, int, const unsigned char *) {}
	void check(int, bool) {}
};
#endif

// stage buffers are allocated as int(BlockSize * 1.05) by the caller (jampack.cpp:74-76, 157-159)
static int stage_capacity(const Options &Opt) { return (int)((double)Opt.BlockSize * 1.05); }

void BlockSort::Bwt::ForwardBwt(Buffer Input, Buffer Output)
{
	int n = 0;
	Guard g("ForwardBwt", Input.block, *Input.size, Output.block);
	int rc = jpk_bwt_forward(Input.block, *Input.size, Output.block, *Input.size + JPK_TRAILER_BYTES, &n);
	if (rc) fail("Bwt", rc);
	g.check(n, false);
	*Output.size = n;
	trace("ForwardBwt", Input.block, *Input.size, Output.block, n);
}

void BlockSort::Bwt::InverseBwt(Buffer Input, Buffer Output, Options Opt)
{
	int n = 0;
	int rc = jpk_bwt_inverse(Input.block, *Input.size, Output.block, *Input.size - JPK_TRAILER_BYTES, &n, (int)Opt.Threads, Opt.Gpu ? 1 : 0);
	if (rc) fail("Bwt", rc);
	*Input.size -= JPK_TRAILER_BYTES;      // the reference shrinks the caller's input size (bwt.cpp:77)
	*Output.size = n;
	trace("InverseBwt", Input.block, *Input.size + JPK_TRAILER_BYTES, Output.block, n);
}

void Ans::Encode(Buffer Input, Buffer Output, Options Opt)
{
	int n = 0;
	Guard g("AnsEncode", Input.block, *Input.size, Output.block);
	int rc = jpk_ans_encode(Input.block, *Input.size, Output.block, stage_capacity(Opt), &n);
	if (rc) fail("Ans", rc);
	g.check(n, false);
	*Output.size = n;
	trace("AnsEncode", Input.block, *Input.size, Output.block, n);
}

// Capacity of a stage buffer of the caller.  Ans::Decode has no capacity argument and Options.BlockSize is not it on the
// decompress path (the CLI default, main.cpp:60, while the buffers follow the frame header, jampack.cpp:146-159).  Every stage
// buffer of the reference's call sites is the start of a heap block (calloc / realloc in jampack.cpp:74-76, 157-159), so the
// allocator itself knows the real capacity; -DJPK_SHIM_NO_HEAP_QUERY (callers with buffers that are not heap blocks) falls back
// to what the stream declares.
static long buffer_capacity(const unsigned char *block)
{
#if defined(__GLIBC__) && !defined(JPK_SHIM_NO_HEAP_QUERY)
	return block ? (long)malloc_usable_size((void *)block) : 0;
#else
	(void)block;
	return -1;
#endif
}

void Ans::Decode(Buffer Input, Buffer Output, Options Opt)
{
	// Not stage_capacity(Opt), see buffer_capacity().  The stream's own chunk headers declare the decoded size (validated by
	// the header walk); a stream that declares more than the caller's buffer holds is refused instead of overflowing it in the
	// final device-to-host copy (the reference would write past the buffer chunk by chunk).
	int64_t need = 0;
	int rc = jpk_ans_decoded_size(Input.block, *Input.size, &need, 0);
	if (rc) fail("Ans", rc);
	if (need > (int64_t)((double)JPK_MAX_BLOCKSIZE * 1.05)) fail("Ans", JPK_E_CORRUPT);
	const long cap = buffer_capacity(Output.block);
	if (cap >= 0 && need > (int64_t)cap) fail("Ans", JPK_E_CAPACITY);
	int n = 0;
	rc = jpk_ans_decode(Input.block, *Input.size, Output.block, (int)need, &n, (int)Opt.Threads);
	if (rc) fail("Ans", rc);
	*Output.size = n;
	trace("AnsDecode", Input.block, *Input.size, Output.block, n);
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
