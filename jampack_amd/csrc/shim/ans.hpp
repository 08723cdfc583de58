// ans.hpp (shim) -- Ans with the reference's public signatures (ans.hpp:32-33), implemented on MI355X.
// Replaces ans.cpp, model.cpp, rle.cpp, rans_byte.hpp and the LEB128 part of utils.cpp.
#ifndef JPK_SHIM_ANS_H
#define JPK_SHIM_ANS_H

#include "format.hpp"
#include "rank.hpp"

class Ans
{
	public:
	void Encode(Buffer Input, Buffer Output, Options Opt);
	void Decode(Buffer Input, Buffer Output, Options Opt);
};
#endif
