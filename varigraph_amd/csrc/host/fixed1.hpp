// fixed1.hpp -- what `oss << std::fixed << std::setprecision(1) << x` writes (src/genotype.cpp:1579-1696 formats GQ, GPP and CAK that
// way; libstdc++ does vsnprintf("%.1f", (double)x): the exact binary value rounded to one decimal, ties to even), without the stream.
// Half a million VCF lines per sample carry four or five such numbers each: through an ostringstream they were most of the host's
// work behind the device (1.0 thread-second per chr20-scale sample).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>

namespace vgh {

// for a float, x * 10 is exact in double (24 + 4 significant bits), so the nearest-even integer of it IS printf's rounding
inline void append_fixed1(std::string& out, float x)
{
    const double a = std::fabs((double)x);
    if (!(a < 1e15)) {      // inf, nan, huge: the library's own words
        char buf[64];
        out.append(buf, (size_t)snprintf(buf, sizeof buf, "%.1f", (double)x));
        return;
    }
    const uint64_t n = (uint64_t)std::nearbyint(a * 10.0);
    char buf[24];
    char* const e = buf + sizeof buf;
    char* q = e;
    *--q = (char)('0' + n % 10);
    *--q = '.';
    uint64_t whole = n / 10;
    do {
        *--q = (char)('0' + whole % 10);
        whole /= 10;
    } while (whole);
    if (std::signbit(x)) *--q = '-';
    out.append(q, (size_t)(e - q));
}

// a long double's x * 10 is not exact in any type at hand: the library rounds it
inline void append_fixed1(std::string& out, long double x)
{
    char buf[64];
    out.append(buf, (size_t)snprintf(buf, sizeof buf, "%.1Lf", x));
}

inline void append_uint(std::string& out, uint64_t v)
{
    char buf[24];
    char* const e = buf + sizeof buf;
    char* q = e;
    do {
        *--q = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    out.append(q, (size_t)(e - q));
}

}  // namespace vgh
