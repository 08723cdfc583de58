// jam_block_pipeline.cpp -- the tail of Jampack::Comp() / Decomp() (jampack.cpp:30-58) written against the
// shim headers exactly as the reference writes it against its own: two heap buffers of 1.05 x BlockSize that are
// pointer-swapped between stages.  Usage: jam_block_pipeline <file> [blocksize_MiB] [threads]  -- compresses every block,
// decompresses it again, verifies, prints MB/s (wall clock, PCIe staging included).  With threads > 1 the blocks are then
// processed again the way Jampack::Compress / Decompress do in multi-block mode (jampack.cpp:205-224, 286-317): one Pipeline
// (= one Jampack instance) per thread, the threads take the blocks of the file in turn; every thread borrows a GPU context of
// its own from the library, so `threads` blocks are in flight.
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
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
	void Setup(int bs)
	{
		Option.BlockSize = bs; Option.MatchFinder = 0; Option.Threads = 8; Option.Filters = 0; Option.Gpu = true; Option.Multiblock = false;
		const int cap = (int)(bs * 1.05) + 4096;
		Input.block = (unsigned char *)calloc(cap, 1);
		Output.block = (unsigned char *)calloc(cap, 1);
		Input.size = (int *)calloc(1, sizeof(int));
		Output.size = (int *)calloc(1, sizeof(int));
	}
};

// multi-block mode: `threads` pipelines take the blocks in turn; compressed blocks are kept, then decoded the same way
static void RunThreads(const std::vector<std::vector<unsigned char>> &blocks, int bs, int threads)
{
	const int nb = (int)blocks.size();
	std::vector<std::vector<unsigned char>> comp((size_t)nb);
	std::vector<Pipeline *> pipes;
	for (int t = 0; t < threads; t++) { pipes.push_back(new Pipeline()); pipes.back()->Setup(bs); }
	double secs[2] = {0, 0};
	for (int rep = 0; rep < 2; rep++) {            // rep 0 warms every thread's context (arena, staging buffers)
		for (int dir = 0; dir < 2; dir++) {
			std::atomic<int> next{0};
			std::atomic<int> bad{0};
			auto work = [&](Pipeline *p) {
				for (;;) {
					const int b = next.fetch_add(1);
					if (b >= nb) return;
					if (dir == 0) {
						memcpy(p->Input.block, blocks[(size_t)b].data(), blocks[(size_t)b].size());
						*p->Input.size = (int)blocks[(size_t)b].size();
						p->Comp();
						comp[(size_t)b].assign(p->Output.block, p->Output.block + *p->Output.size);
					} else {
						memcpy(p->Input.block, comp[(size_t)b].data(), comp[(size_t)b].size());
						*p->Input.size = (int)comp[(size_t)b].size();
						p->Decomp();
						if (*p->Output.size != (int)blocks[(size_t)b].size() || memcmp(p->Output.block, blocks[(size_t)b].data(), blocks[(size_t)b].size()) != 0) bad++;
					}
				}
			};
			auto t0 = std::chrono::steady_clock::now();
			std::vector<std::thread> th;
			for (int t = 1; t < threads; t++) th.emplace_back(work, pipes[(size_t)t]);
			work(pipes[0]);
			for (auto &x : th) x.join();
			secs[dir] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (bad) Error("round trip mismatch (threads)!");
		}
	}
	long long total = 0;
	for (auto &b : blocks) total += (long long)b.size();
	printf("%d threads (blocks in flight), %d blocks: compress %.1f MB/s, decompress %.1f MB/s (PCIe staging and host copies included), round trip ok\n",
	       threads, nb, total / 1e6 / secs[0], total / 1e6 / secs[1]);
}

int main(int argc, char **argv)
{
	if (argc < 2) { printf("usage: %s file [blocksize_MiB] [threads]\n", argv[0]); return 2; }
	const int bs = (argc > 2 ? atoi(argv[2]) : 8) << 20;
	FILE *f = fopen(argv[1], "rb");
	if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
	const int threads = argc > 3 ? atoi(argv[3]) : 1;
	Pipeline p;
	p.Setup(bs);
	unsigned char *orig = (unsigned char *)malloc(bs);
	std::vector<std::vector<unsigned char>> all_blocks;
	double tc = 0, td = 0, tc1 = 0, td1 = 0;      // all blocks / all but the first (which pays for the HBM arena and staging allocations)
	long long in_total = 0, out_total = 0, in_after_first = 0;
	int nblocks = 0;
	for (;;) {
		int n = (int)fread(orig, 1, bs, f);
		if (n <= 0) break;
		if (threads > 1) all_blocks.emplace_back(orig, orig + n);
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
	if (threads > 1 && !all_blocks.empty()) RunThreads(all_blocks, bs, threads);
	return 0;
}
