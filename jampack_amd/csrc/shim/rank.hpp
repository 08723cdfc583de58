// rank.hpp (shim) -- Postcoder with the reference's signatures (rank.hpp:12-13), implemented on MI355X.
#ifndef JPK_SHIM_RANK_H
#define JPK_SHIM_RANK_H

#include "format.hpp"

class Postcoder
{
public:
	void Encode(unsigned char *T, int *Freq, int len);
	void Decode(unsigned char *RankArray, int *Freq, int len);
};
#endif
