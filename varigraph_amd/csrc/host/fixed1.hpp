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

// a long double's x * 10 is not exact in any floating type at hand, but it is in 128-bit integers: x = m 2^(e - 64) with a 64-bit m, so
// 10 x = (10 m) / 2^(64 - e) -- quotient and remainder are exact, and round-half-even of them is what the library's "%.1Lf" prints
// (it rounds the exact binary value).  The posterior of a call (GPP) is in [0, 1]: half a million snprintf calls per chr20-scale
// sample were 0.15 of its 0.4 thread-seconds of VCF text.  Anything outside [0, 1e15) takes the library.
inline void append_fixed1(std::string& out, long double x)
{
    if (!(x >= 0.0L && x < 1e15L) || std::signbit(x)) {      // (negative zero included: the library writes its sign)
        char buf[64];
        const int len = snprintf(buf, sizeof buf, "%.1Lf", x);
        if (len >= 0 && (size_t)len < sizeof buf) out.append(buf, (size_t)len);
        else if (len > 0) {                                     // thousands of digits: a long double reaches 1e4932
            std::string big((size_t)len + 1, '\0');
            snprintf(&big[0], big.size(), "%.1Lf", x);
            out.append(big.data(), (size_t)len);
        }
        return;
    }
    uint64_t n = 0;
    if (x != 0.0L) {
        int ex = 0;
        const long double fr = frexpl(x, &ex);                      // x = fr 2^ex, 0.5 <= fr < 1
        const uint64_t m = (uint64_t)ldexpl(fr, 64);               // exact: the 64-bit significand
        const int s = 64 - ex;                                      // x = m / 2^s; ex <= 50 here, so s >= 14
        const unsigned __int128 t = (unsigned __int128)m * 10u;     // < 2^68
        if (s >= 70) n = 0;                                         // 10 x < 2^68 / 2^70: below a quarter, rounds to 0
        else {
            const unsigned __int128 q = t >> s, rem = t & (((unsigned __int128)1 << s) - 1), half = (unsigned __int128)1 << (s - 1);
            n = (uint64_t)q;
            if (rem > half || (rem == half && (n & 1u))) ++n;
        }
    }
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
    out.append(q, (size_t)(e - q));
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
