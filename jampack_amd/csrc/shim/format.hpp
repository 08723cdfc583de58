// format.hpp (shim) -- the shared types of the reference's stage interface, re-declared layout-compatibly
// (reference: format.hpp:26, 32-54, 59).  Only what the BWT / entropy stages need.
#ifndef JPK_SHIM_FORMAT_H
#define JPK_SHIM_FORMAT_H

#include <stdint.h>

#define BWT_UNITS 120            // sampled BWT ranks stored behind every block (format.hpp:26)

typedef int Index;

struct Buffer {                  // passed by value between stages; the caller owns both pointers
    unsigned char *block;
    Index *size;
};

struct Options {                 // command-line options handed to every stage (format.hpp:46-54)
    Index BlockSize;
    unsigned int MatchFinder;
    unsigned int Threads;
    unsigned int Filters;
    bool Gpu;
    bool Multiblock;
};

extern void Error(const char *string);   // provided by the host program (format.cpp:6-10)

#endif
