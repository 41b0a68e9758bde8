// append_fixed1 / append_uint (csrc/host/fixed1.hpp) against the stream formatting they replace
#include <cstdio>
#include <cstring>
#include <iomanip>
#include <limits>
#include <random>
#include <sstream>

#include "fixed1.hpp"

static std::string by_stream(float x)
{
    std::ostringstream o;
    o << std::fixed << std::setprecision(1) << x;
    return o.str();
}
static std::string by_stream(long double x)
{
    std::ostringstream o;
    o << std::fixed << std::setprecision(1) << x;
    return o.str();
}

int main()
{
    size_t bad = 0, n = 0;
    auto check = [&](float x) {
        std::string a;
        vgh::append_fixed1(a, x);
        if (a != by_stream(x)) {
            if (bad < 10) std::printf("float %a: %s vs %s\n", (double)x, a.c_str(), by_stream(x).c_str());
            ++bad;
        }
        ++n;
    };
    // every tie and near-tie of the first decimals, both signs; powers of two; the ends of the range
    for (int k = -2000; k <= 2000; ++k)
        for (int d = -2; d <= 2; ++d) {
            const float x = (float)k / 20.0f;
            check(std::nextafterf(x, d < 0 ? -1e30f : 1e30f) * (d == 0 ? 0.0f : 1.0f) + (d == 0 ? x : 0.0f));
            float y = x;
            for (int s2 = 0; s2 < (d < 0 ? -d : d); ++s2) y = std::nextafterf(y, d < 0 ? -1e30f : 1e30f);
            check(y);
        }
    for (int e = -149; e <= 127; ++e) { check(std::ldexp(1.0f, e)); check(-std::ldexp(1.0f, e)); check(std::ldexp(1.5f, e)); }
    for (float x : {0.0f, -0.0f, 0.05f, -0.05f, 0.25f, 0.75f, 99.0f, 99.95f, 1e15f, 9.9e14f, std::numeric_limits<float>::max(), std::numeric_limits<float>::infinity(),
                    -std::numeric_limits<float>::infinity(), std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::denorm_min()})
        check(x);
    std::mt19937_64 rng(7);
    for (int i = 0; i < 3000000; ++i) {        // random bit patterns: every exponent
        uint32_t b = (uint32_t)rng();
        float x;
        std::memcpy(&x, &b, 4);
        check(x);
    }
    std::uniform_real_distribution<float> cov(0.0f, 300.0f);
    for (int i = 0; i < 3000000; ++i) check(cov(rng));         // the values the VCF holds
    for (long double x : {0.0L, 1.0L, 0.25L, 0.75L, 0.95L, 0.9999999999L, 0.04999999999L, 0.05L}) {
        std::string a;
        vgh::append_fixed1(a, x);
        if (a != by_stream(x)) ++bad;
        ++n;
    }
    {   // long double (the calls' posteriors): ties and near-ties of every tenth, random values in [0, 1] and beyond, tiny ones, the library's cases
        auto check_ld = [&](long double x) {
            std::string a;
            vgh::append_fixed1(a, x);
            if (a != by_stream(x)) {
                if (bad < 10) std::printf("long double %La: %s vs %s\n", x, a.c_str(), by_stream(x).c_str());
                ++bad;
            }
            ++n;
        };
        for (int k = 0; k <= 4000; ++k) {
            long double x = (long double)k / 20.0L;
            check_ld(x);
            long double lo = x, hi = x;
            for (int d = 0; d < 3; ++d) {
                lo = std::nextafterl(lo, -1.0L);
                hi = std::nextafterl(hi, 1e9L);
                if (lo >= 0) check_ld(lo);
                check_ld(hi);
            }
        }
        std::uniform_real_distribution<double> u01(0.0, 1.0);
        for (int i = 0; i < 2000000; ++i) {
            const long double x = (long double)u01(rng) + (long double)u01(rng) * 0x1p-53L;      // all 64 bits of the significand in play
            check_ld(x);
            check_ld(x * 0.01L);
            check_ld(x * 1234.5L);
        }
        for (int e = -16445; e <= 60; e += (e < -80 || e > 0) ? 37 : 1) { check_ld(std::ldexp(1.0L, e)); check_ld(std::ldexp(1.9999999999999999999L, e)); }
        for (long double x : {1e15L, 9.99999e14L, -0.0L, -1.0L, 1e300L, std::numeric_limits<long double>::infinity(), std::numeric_limits<long double>::quiet_NaN(),
                              std::numeric_limits<long double>::denorm_min(), 0.95L, 0.05L, 0.15L, 0.25L, 0.35L, 0.45L, 0.55L, 0.65L, 0.75L, 0.85L})
            check_ld(x);
    }
    for (uint64_t v : {0ull, 9ull, 10ull, 255ull, 18446744073709551615ull}) {
        std::string a;
        vgh::append_uint(a, v);
        if (a != std::to_string(v)) ++bad;
        ++n;
    }
    std::printf("%zu values, %zu different\n", n, bad);
    return bad != 0;
}
