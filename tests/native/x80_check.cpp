// csrc/vg_x80.h (extended precision in integer arithmetic, non-negative operands) against the x87 unit: products, sums and
// quotients of random and edge operands -- normal numbers over the whole exponent range, results that underflow gradually,
// denormal operands, zeros, significands of all ones / single bits / short patterns; the normalised-form variants (n80_*)
// on the same operands and along the chains.  Prints "<n> cases, <k> mismatches".
// Test infrastructure (tests/test_x80_cpu.py).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "vg_x80.h"

static long double to_ld(VgX80 v)
{
    unsigned char b[16];
    x80_store(b, v);
    long double x;
    memcpy(&x, b, sizeof x);
    return x;
}

static VgX80 from_ld(long double x)
{
    unsigned char b[16] = {0};
    memcpy(b, &x, 10);
    return x80_load(b);
}

static bool same(VgX80 a, VgX80 b) { return a.m == b.m && a.e == b.e; }

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 3000000;
    std::mt19937_64 rng(12345);
    auto significand = [&]() -> uint64_t {
        switch (rng() % 8) {
            case 0: return ~0ULL;
            case 1: return 1ULL << 63;
            case 2: return (1ULL << 63) | 1u;
            case 3: return (1ULL << 63) | (rng() & 0xFFFF);
            case 4: return ~(rng() & 0xFFFF);
            case 5: return (1ULL << 63) | (1ULL << (rng() % 63));
            default: return rng() | (1ULL << 63);
        }
    };
    auto value = [&]() -> VgX80 {
        VgX80 v;
        const unsigned kind = (unsigned)(rng() % 16);
        if (kind == 0) {
            v.m = 0;
            v.e = 0;
        } else if (kind == 1) {                       // denormal
            v.e = 0;
            v.m = significand() >> (1 + rng() % 63);
            if (v.m == 0) v.m = 1;
        } else if (kind < 5) {                        // close to the underflow boundary
            v.e = 1 + (uint32_t)(rng() % 140);
            v.m = significand();
        } else if (kind < 8) {                        // products of these land around the boundary
            v.e = 8100 + (uint32_t)(rng() % 300);
            v.m = significand();
        } else if (kind < 12) {                       // probabilities
            v.e = VG_X80_BIAS - (uint32_t)(rng() % 200);
            v.m = significand();
        } else {                                      // anything below 2^8
            v.e = 1 + (uint32_t)(rng() % (VG_X80_BIAS + 8));
            v.m = significand();
        }
        return v;
    };
    long bad = 0, cases = 0;
    for (long i = 0; i < n; ++i) {
        const VgX80 a = value();
        VgX80 b = value();
        if (rng() % 3 == 0 && a.e > 80 && b.e) b.e = a.e - 70 + (uint32_t)(rng() % 140);   // every alignment distance of a sum
        volatile long double x = to_ld(a), y = to_ld(b);
        {
            volatile long double p = x * y;
            if (!same(from_ld(p), x80_mul(a, b)) || !same(from_ld(p), n80_to(n80_mul(n80_from(a), n80_from(b))))) {
                if (bad++ < 10) printf("mul %016llx:%u * %016llx:%u\n", (unsigned long long)a.m, a.e, (unsigned long long)b.m, b.e);
            }
            ++cases;
        }
        {
            volatile long double s = x + y;
            if (!same(from_ld(s), x80_add(a, b)) || !same(from_ld(s), n80_to(n80_add(n80_from(a), n80_from(b)))) ||
                !same(from_ld(s), n80_to(n80_sum(n80_from(a), n80_from(b))))) {
                if (bad++ < 10) printf("add %016llx:%u + %016llx:%u\n", (unsigned long long)a.m, a.e, (unsigned long long)b.m, b.e);
            }
            ++cases;
        }
        if (b.m != 0 && (int)a.e - (int)b.e < 16000) {     // no overflow in the quotient
            volatile long double q = x / y;
            if (!same(from_ld(q), x80_div(a, b)) || !same(from_ld(q), n80_to(n80_div(n80_from(a), n80_from(b))))) {
                if (bad++ < 10) printf("div %016llx:%u / %016llx:%u\n", (unsigned long long)a.m, a.e, (unsigned long long)b.m, b.e);
            }
            ++cases;
        }
    }
    // the fused term  r + s * o  with r placed around the product (every alignment distance, both orders), and anywhere
    for (long i = 0; i < n; ++i) {
        const VgX80 sv = value(), o = value();
        if ((int)sv.e + (int)o.e > 2 * VG_X80_BIAS - 100) continue;
        VgX80 r = value();
        if (rng() % 4 && sv.m && o.m) {
            const int pe = (int)(sv.e ? sv.e : 1) + (int)(o.e ? o.e : 1) - VG_X80_BIAS + (int)(rng() % 140) - 70;
            r.e = pe < 1 ? 0 : (uint32_t)pe;
            r.m = r.e ? significand() : significand() >> (1 + rng() % 63);
        }
        volatile long double x = to_ld(r), y = to_ld(sv), z = to_ld(o);
        volatile long double p = y * z;
        volatile long double w = x + p;
        const VgN80 f = n80_muladd(n80_from(r), n80_from(sv), n80_from(o));
        if (!same(from_ld(w), n80_to(f))) {
            if (bad++ < 10)
                printf("muladd %016llx:%u + %016llx:%u * %016llx:%u\n", (unsigned long long)r.m, r.e, (unsigned long long)sv.m, sv.e,
                       (unsigned long long)o.m, o.e);
        }
        ++cases;
    }
    // chains as the recursion builds them: r += s * o over 120 terms
    for (long i = 0; i < n / 200; ++i) {
        VgX80 r = {0, 0};
        VgN80 rn = {0, 0}, rf = {0, 0};
        volatile long double rr = 0.0L;
        const VgX80 o = value();
        volatile long double oo = to_ld(o);
        for (int t = 0; t < 120; ++t) {
            const VgX80 s = value();
            volatile long double ss = to_ld(s);
            if ((int)s.e + (int)o.e > 2 * VG_X80_BIAS - 100) continue;
            r = x80_add(r, x80_mul(s, o));
            rn = n80_add(rn, n80_mul(n80_from(s), n80_from(o)));     // stays normalised along the chain
            rf = n80_muladd(rf, n80_from(s), n80_from(o));
            if (rf.m != rn.m || rf.e != rn.e) {
                if (bad++ < 10) printf("muladd term %d\n", t);
                rf = rn;
            }
            rr = rr + ss * oo;
        }
        if (!same(from_ld(rr), r) || !same(from_ld(rr), n80_to(rn))) ++bad;
        ++cases;
    }
    printf("%ld cases, %ld mismatches\n", cases, bad);
    return bad != 0;
}
