// jam_block_pipeline.cpp -- the tail of Jampack::Comp() / Decomp() (jampack.cpp:30-58) written against the
// shim headers exactly as the reference writes it against its own: two heap buffers of 1.05 x BlockSize that are
// pointer-swapped between stages.  Usage: jam_block_pipeline <file> [blocksize_MiB]  -- compresses every block,
// decompresses it again, verifies, prints MB/s (wall clock, PCIe staging included).
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ans.hpp"
#include "bwt.hpp"

void Error(const char *string)
{
	printf("\n Error: %s \n", string);
	exit(-1);
}

struct Pipeline {
	Buffer Input, Output;
	Options Option;
	BlockSort::Bwt *Bwt = new BlockSort::Bwt();
	Ans *Entropy = new Ans();
	void SwapStreams() { Buffer t = Input; Input = Output; Output = t; }
	void Comp() { Bwt->ForwardBwt(Input, Output); SwapStreams(); Entropy->Encode(Input, Output, Option); }
	void Decomp() { Entropy->Decode(Input, Output, Option); SwapStreams(); Bwt->InverseBwt(Input, Output, Option); }
};

int main(int argc, char **argv)
{
	if (argc < 2) { printf("usage: %s file [blocksize_MiB]\n", argv[0]); return 2; }
	const int bs = (argc > 2 ? atoi(argv[2]) : 8) << 20;
	FILE *f = fopen(argv[1], "rb");
	if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
	Pipeline p;
	p.Option.BlockSize = bs; p.Option.MatchFinder = 0; p.Option.Threads = 8; p.Option.Filters = 0; p.Option.Gpu = true; p.Option.Multiblock = false;
	const int cap = (int)(bs * 1.05) + 4096;
	p.Input.block = (unsigned char *)calloc(cap, 1);
	p.Output.block = (unsigned char *)calloc(cap, 1);
	p.Input.size = (int *)calloc(1, sizeof(int));
	p.Output.size = (int *)calloc(1, sizeof(int));
	unsigned char *orig = (unsigned char *)malloc(bs);
	double tc = 0, td = 0, tc1 = 0, td1 = 0;      // all blocks / all but the first (which pays for the HBM arena and staging allocations)
	long long in_total = 0, out_total = 0, in_after_first = 0;
	int nblocks = 0;
	for (;;) {
		int n = (int)fread(orig, 1, bs, f);
		if (n <= 0) break;
		memcpy(p.Input.block, orig, n);
		*p.Input.size = n;
		auto t0 = std::chrono::steady_clock::now();
		p.Comp();
		auto t1 = std::chrono::steady_clock::now();
		const int csize = *p.Output.size;
		p.SwapStreams();                  // compressed block becomes the decoder's input
		p.Decomp();
		auto t2 = std::chrono::steady_clock::now();
		if (*p.Output.size != n || memcmp(p.Output.block, orig, n) != 0) Error("round trip mismatch!");
		tc += std::chrono::duration<double>(t1 - t0).count();
		td += std::chrono::duration<double>(t2 - t1).count();
		if (nblocks > 0) { tc1 += std::chrono::duration<double>(t1 - t0).count(); td1 += std::chrono::duration<double>(t2 - t1).count(); in_after_first += n; }
		nblocks++;
		in_total += n; out_total += csize;
	}
	fclose(f);
	printf("%lld -> %lld bytes, compress %.1f MB/s, decompress %.1f MB/s (PCIe staging included), round trip ok\n", in_total, out_total,
	       in_total / 1e6 / (tc > 0 ? tc : 1), in_total / 1e6 / (td > 0 ? td : 1));
	if (nblocks > 1)
		printf("steady state (blocks 2..%d, allocations done): compress %.1f MB/s, decompress %.1f MB/s\n", nblocks,
		       in_after_first / 1e6 / (tc1 > 0 ? tc1 : 1), in_after_first / 1e6 / (td1 > 0 ? td1 : 1));
	return 0;
}
