// bwt.hpp (shim) -- BlockSort::Bwt with the reference's signatures (bwt.hpp:13-18), implemented on MI355X
// by libjampack_amd.so.  Drop this header and shim.cpp in place of bwt.hpp / bwt.cpp / divsufsort.cpp.
#ifndef JPK_SHIM_BWT_H
#define JPK_SHIM_BWT_H

#include "format.hpp"

namespace BlockSort
{
	class Bwt
	{
		public:
		void ForwardBwt(Buffer Input, Buffer Output);
		void InverseBwt(Buffer Input, Buffer Output, Options Opt);
	};
}
#endif
