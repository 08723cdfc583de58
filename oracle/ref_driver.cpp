// TEST INFRASTRUCTURE ONLY -- not part of the shipped library.
//
// C-ABI driver around the *real* reference classes, compiled together with the reference's own
// sources where they lie under /root/reference (see oracle/Makefile, target `ref`).  The output
// (oracle/_ref/libjamref.so) is used (a) to pin the CPU restatement in oracle/jam_oracle.c,
// (b) to generate tests/golden/, (c) as bench.py's cpu_baseline (kind "reference").
// Nothing under jampack_amd/ links or loads it.
//
// Reference entry points wrapped here:
//   BlockSort::Bwt::ForwardBwt / InverseBwt   bwt.hpp:13-18, bwt.cpp:22-282
//   Ans::Encode / Ans::Decode                 ans.hpp:32-33, ans.cpp:113-270
//   Postcoder::Encode / Decode                rank.hpp:12-13, rank.cpp:45-151
//   RLE::encode / decode                      rle.hpp:9-10,  rle.cpp:22-74
//   Utils::EncodeLeb128 / DecodeLeb128        utils.cpp:22-90
//   divsufsort                                divsufsort.cpp:1721
//   Checksum::IntegrityCheck                  checksum.hpp:15, checksum.cpp:12-36
#include "bwt.hpp"
#include "ans.hpp"
#include "rank.hpp"
#include "rle.hpp"
#include "utils.hpp"
#include "divsufsort.hpp"
#include "checksum.hpp"

static Options make_opt(int threads)
{
	Options o;
	o.BlockSize = 64 << 20;
	o.MatchFinder = 0;
	o.Threads = threads < 1 ? 1 : (unsigned)threads;
	o.Filters = 0;
	o.Gpu = false;
	o.Multiblock = false;
	return o;
}

extern "C" {

// out must hold len + 480 bytes
int ref_bwt_forward(unsigned char* in, int len, unsigned char* out, int* out_len)
{
	int isz = len, osz = 0;
	Buffer I{in, &isz}, O{out, &osz};
	BlockSort::Bwt b;
	b.ForwardBwt(I, O);
	*out_len = osz;
	return 0;
}

int ref_bwt_inverse(unsigned char* in, int len_with_trailer, unsigned char* out, int* out_len, int threads)
{
	int isz = len_with_trailer, osz = 0;
	Buffer I{in, &isz}, O{out, &osz};
	BlockSort::Bwt b;
	b.InverseBwt(I, O, make_opt(threads));
	*out_len = osz;
	return isz; // the reference rewrites *Input.size (bwt.cpp:77)
}

int ref_ans_encode(unsigned char* in_clobbered, int len, unsigned char* out, int* out_len)
{
	int isz = len, osz = 0;
	Buffer I{in_clobbered, &isz}, O{out, &osz};
	Ans a;
	a.Encode(I, O, make_opt(1));
	*out_len = osz;
	return 0;
}

int ref_ans_decode(unsigned char* in, int len, unsigned char* out, int* out_len, int threads)
{
	int isz = len, osz = 0;
	Buffer I{in, &isz}, O{out, &osz};
	Ans a;
	a.Decode(I, O, make_opt(threads));
	*out_len = osz;
	return 0;
}

void ref_rank_encode(unsigned char* t, int* freq256, int len)
{
	Postcoder p;
	p.Encode(t, freq256, len);
}

void ref_rank_decode(unsigned char* ranks, int* freq256, int len)
{
	Postcoder p;
	p.Decode(ranks, freq256, len);
}

int ref_rle_encode(unsigned char* in, unsigned short* out, int len)
{
	RLE r;
	int l = len;
	r.encode(in, out, &l);
	return l;
}

int ref_rle_decode(unsigned short* in, unsigned char* out, int rlen, int real_len)
{
	RLE r;
	int l = rlen;
	r.decode(in, out, &l, real_len);
	return l;
}

int ref_leb_encode(int val, unsigned char* buf)
{
	Utils* u = new Utils;
	int n = u->EncodeLeb128(val, buf);
	delete u;
	return n;
}

int ref_leb_decode(int* val, unsigned char* buf)
{
	Utils* u = new Utils;
	int n = u->DecodeLeb128(val, buf);
	delete u;
	return n;
}

unsigned int ref_checksum(unsigned char* p, int size)
{
	Checksum c;
	Buffer b{p, &size};
	return c.IntegrityCheck(b);
}

int ref_divsufsort(const unsigned char* t, int* sa, int n)
{
	return divsufsort(t, sa, n);
}

} // extern "C"
