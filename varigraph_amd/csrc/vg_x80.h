// vg_x80.h -- x87 extended precision (the reference's `long double`: 64-bit significand with an explicit integer bit, 15-bit
// exponent) for NON-NEGATIVE finite values, in integer arithmetic, for host and device alike.
//
// The reference's HMM (src/genotype.cpp:1170-1380) multiplies and adds probabilities in `long double`; a result identical
// to its bits needs every operation rounded exactly as the x87 unit rounds it under the default control word: ONE
// round-to-nearest-even at the precision of the result, which for a result below 2^-16382 is the reduced precision of a
// denormal (gradual underflow -- emission products of a hundred k-mers do get there).  Everything here is a probability
// or a product / sum / quotient of probabilities: no signs, no infinities, no NaNs, no overflow.
//
//   x80_mul   64 x 64 -> 128-bit product of the significands, one rounding
//   x80_add   the smaller operand aligned into 128 bits with a sticky bit, one rounding
//   x80_div   128 / 64 restoring division of the significands, guard and sticky from the remainder, one rounding
//
//   n80_*     the same three on values kept normalised (VgN80) along a chain of operations, n80_muladd / n80_sum the
//             branch-free forms the device recursion runs (vgmi_hmm.hip); the general functions remain their fallback
//
// tests/native/x80_check.cpp holds each against the x87 unit on tens of millions of random and edge operands.
#ifndef VG_X80_H
#define VG_X80_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VG_X80_HD __host__ __device__ static inline
#else
#define VG_X80_HD static inline
#endif

struct VgX80 {
    uint64_t m;   // significand, bit 63 = integer bit (clear for a denormal)
    uint32_t e;   // biased exponent field: 0 for zero and denormals (whose exponent is that of field 1)
};

#define VG_X80_BIAS 16383

VG_X80_HD VgX80 x80_load(const void* p)      // the 10 significant bytes of an x86-64 long double (16 in memory)
{
    VgX80 v;
    uint16_t se;
    memcpy(&v.m, p, 8);
    memcpy(&se, (const char*)p + 8, 2);
    v.e = se & 0x7FFFu;
    return v;
}

VG_X80_HD void x80_store(void* p, VgX80 v)
{
    const uint16_t se = (uint16_t)v.e;
    memset(p, 0, 16);
    memcpy(p, &v.m, 8);
    memcpy((char*)p + 8, &se, 2);
}

VG_X80_HD void x80_mul64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo)
{
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    lo = (uint64_t)p;
    hi = (uint64_t)(p >> 64);
#endif
}

VG_X80_HD uint32_t x80_clz64(uint64_t x)   // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clzll((long long)x);
#else
    return (uint32_t)__builtin_clzll(x);
#endif
}

// (hi:lo) is a 128-bit significand whose bit 127 has the weight 2^(e_eff - BIAS); `sticky` says that non-zero bits lie below
// bit 0.  Rounds once, to nearest even, to 64 bits -- fewer when e_eff < 1 (denormal) -- and packs.
VG_X80_HD VgX80 x80_round_pack(uint64_t hi, uint64_t lo, int32_t e_eff, bool sticky)
{
    uint32_t extra = 0;
    if (e_eff < 1) {
        extra = (uint32_t)(1 - e_eff);
        e_eff = 1;
    }
    if (extra) {
        if (extra >= 128) {     // the whole significand lies below the guard bit of the smallest denormal: zero
            VgX80 r = {0, 0};
            return r;
        }
        if (extra >= 64) {
            const uint32_t s = extra - 64;
            sticky = sticky || lo != 0 || (s ? (hi << (64 - s)) != 0 : false);
            lo = s ? hi >> s : hi;
            hi = 0;
        } else {
            sticky = sticky || (lo << (64 - extra)) != 0;
            lo = (lo >> extra) | (hi << (64 - extra));
            hi >>= extra;
        }
    }
    const bool guard = (lo >> 63) != 0;
    sticky = sticky || (lo << 1) != 0;
    uint64_t m = hi;
    int32_t e = e_eff;
    if (guard && (sticky || (m & 1u))) {
        ++m;
        if (m == 0) {           // 2^64: one binade up
            m = 1ULL << 63;
            ++e;
        }
    }
    VgX80 r;
    r.m = m;
    r.e = (m >> 63) ? (uint32_t)e : 0u;   // a denormal that rounds up to the integer bit is the smallest normal (field 1)
    return r;
}

// significand with the integer bit set and the exponent it then has (may be < 1 for a denormal); v.m != 0
VG_X80_HD void x80_normal(VgX80 v, uint64_t& m, int32_t& e)
{
    e = v.e ? (int32_t)v.e : 1;
    m = v.m;
    if (!(m >> 63)) {
        const uint32_t s = x80_clz64(m);
        m <<= s;
        e -= (int32_t)s;
    }
}

VG_X80_HD VgX80 x80_mul(VgX80 a, VgX80 b)
{
    if (a.m == 0 || b.m == 0) {
        VgX80 z = {0, 0};
        return z;
    }
    uint64_t ma, mb, hi, lo;
    int32_t ea, eb;
    x80_normal(a, ma, ea);
    x80_normal(b, mb, eb);
    x80_mul64(ma, mb, hi, lo);      // in [2^126, 2^128)
    int32_t e = ea + eb - VG_X80_BIAS + 1;
    if (!(hi >> 63)) {
        hi = (hi << 1) | (lo >> 63);
        lo <<= 1;
        --e;
    }
    return x80_round_pack(hi, lo, e, false);
}

VG_X80_HD VgX80 x80_add(VgX80 a, VgX80 b)
{
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    int32_t ea = a.e ? (int32_t)a.e : 1, eb = b.e ? (int32_t)b.e : 1;
    if (ea < eb) {
        const VgX80 t = a;
        a = b;
        b = t;
        const int32_t te = ea;
        ea = eb;
        eb = te;
    }
    const uint32_t d = (uint32_t)(ea - eb);
    // b's significand, 64 bits below a's, shifted right by d
    uint64_t bh, bl;
    bool sticky = false;
    if (d == 0) {
        bh = b.m;
        bl = 0;
    } else if (d < 64) {
        bh = b.m >> d;
        bl = b.m << (64 - d);
    } else if (d < 128) {
        const uint32_t s = d - 64;
        bh = 0;
        bl = s ? b.m >> s : b.m;
        sticky = s ? (b.m << (64 - s)) != 0 : false;
    } else {
        bh = 0;
        bl = 0;
        sticky = true;
    }
    uint64_t hi = a.m + bh, lo = bl;
    const bool carry = hi < bh;
    int32_t e = ea;
    if (carry) {
        sticky = sticky || (lo & 1u);
        lo = (lo >> 1) | (hi << 63);
        hi = (hi >> 1) | (1ULL << 63);
        ++e;
    }
    // two denormals: bit 127 may still be clear; the weight of bit 127 is 2^(e - BIAS) either way
    return x80_round_pack(hi, lo, e, sticky);
}

VG_X80_HD VgX80 x80_div(VgX80 a, VgX80 b)    // b != 0
{
    if (a.m == 0) {
        VgX80 z = {0, 0};
        return z;
    }
    uint64_t ma, mb;
    int32_t ea, eb;
    x80_normal(a, ma, ea);
    x80_normal(b, mb, eb);
    uint64_t nh, nl;
    int32_t e = ea - eb + VG_X80_BIAS;
    if (ma >= mb) {         // quotient of the significands in [1, 2): numerator ma * 2^63
        nh = ma >> 1;
        nl = ma << 63;
    } else {                // in (1/2, 1): numerator ma * 2^64, one binade down
        nh = ma;
        nl = 0;
        --e;
    }
    uint64_t q = 0, r = nh;   // r < mb
    for (int i = 63; i >= 0; --i) {
        const bool top = (r >> 63) != 0;
        r = (r << 1) | ((nl >> i) & 1u);
        if (top || r >= mb) {
            r -= mb;
            q |= 1ULL << i;
        }
    }
    // the next quotient bit and whether anything lies below it
    const bool guard = (r >> 63) != 0 || (r << 1) >= mb;
    const uint64_t r2 = (r << 1) - (guard ? mb : 0);
    const uint64_t lo = ((uint64_t)guard << 63) | (r2 != 0 ? 1u : 0u);
    return x80_round_pack(q, lo, e, false);
}

// ---- the same arithmetic on values held NORMALISED ----------------------------------------------------------------------
// A chain of operations (the recursion: a hundred products and sums per lane and node) keeps its values as  m with bit 63 set,
// e the exponent field that significand would have -- below 1 for a denormal, whose low 1 - e bits are then zero -- so that the
// common case (a normal result, e >= 1) is one rounding at 64 bits with no shifts by variable denormal distances; a result
// that is not normal goes through x80_round_pack above.  n80_from / n80_to convert from and to the stored form, exactly.
struct VgN80 {
    uint64_t m;   // 0 for zero, else bit 63 set
    int32_t e;
};

VG_X80_HD VgN80 n80_from(VgX80 v)
{
    VgN80 r = {0, 0};
    if (v.m != 0) x80_normal(v, r.m, r.e);
    return r;
}

VG_X80_HD VgX80 n80_to(VgN80 v)
{
    VgX80 r = {0, 0};
    if (v.m != 0) {
        if (v.e >= 1) {
            r.m = v.m;
            r.e = (uint32_t)v.e;
        } else {
            r.m = v.m >> (uint32_t)(1 - v.e);     // the bits that leave are zero
        }
    }
    return r;
}

// (hi:lo), bit 127 set and of weight 2^(e - BIAS); bit 0 of lo may stand for any non-zero bits below it (it is never the guard)
VG_X80_HD VgN80 n80_round(uint64_t hi, uint64_t lo, int32_t e)
{
    VgN80 r;
    if (e >= 1) {
        const uint64_t inc = (lo | (hi & 1u)) > (1ULL << 63) ? 1u : 0u;     // guard and (sticky or odd)
        r.m = hi + inc;
        r.e = e;
        if (r.m < inc) {        // 2^64: one binade up
            r.m = 1ULL << 63;
            ++r.e;
        }
    } else {
        r = n80_from(x80_round_pack(hi, lo, e, false));
    }
    return r;
}

VG_X80_HD VgN80 n80_mul(VgN80 a, VgN80 b)
{
    if (a.m == 0 || b.m == 0) {
        VgN80 z = {0, 0};
        return z;
    }
    uint64_t hi, lo;
    x80_mul64(a.m, b.m, hi, lo);    // in [2^126, 2^128)
    int32_t e = a.e + b.e - VG_X80_BIAS + 1;
    if (!(hi >> 63)) {
        hi = (hi << 1) | (lo >> 63);
        lo <<= 1;
        --e;
    }
    return n80_round(hi, lo, e);
}

VG_X80_HD VgN80 n80_add(VgN80 a, VgN80 b)
{
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    if (a.e < b.e) {
        const VgN80 t = a;
        a = b;
        b = t;
    }
    const uint32_t d = (uint32_t)(a.e - b.e);
    // b's significand shifted right by d into 128 bits; beyond the guard bit only "something is there" matters
    uint64_t bh, bl;
    if (d < 64) {
        bh = b.m >> d;
        bl = (b.m << 1) << (63 - d);
    } else {
        bh = 0;
        bl = d == 64 ? b.m : 1u;
    }
    uint64_t hi = a.m + bh, lo = bl;
    int32_t e = a.e;
    if (hi < bh) {              // carry: bit 127 is the carry
        lo = (lo >> 1) | (lo & 1u) | (hi << 63);
        hi = (hi >> 1) | (1ULL << 63);
        ++e;
    }
    return n80_round(hi, lo, e);
}

VG_X80_HD VgN80 n80_div(VgN80 a, VgN80 b)    // b != 0
{
    if (a.m == 0) {
        VgN80 z = {0, 0};
        return z;
    }
    uint64_t nh, nl;
    int32_t e = a.e - b.e + VG_X80_BIAS;
    if (a.m >= b.m) {
        nh = a.m >> 1;
        nl = a.m << 63;
    } else {
        nh = a.m;
        nl = 0;
        --e;
    }
    uint64_t q = 0, r = nh;
    for (int i = 63; i >= 0; --i) {
        const bool top = (r >> 63) != 0;
        r = (r << 1) | ((nl >> i) & 1u);
        if (top || r >= b.m) {
            r -= b.m;
            q |= 1ULL << i;
        }
    }
    const bool guard = (r >> 63) != 0 || (r << 1) >= b.m;
    const uint64_t r2 = (r << 1) - (guard ? b.m : 0);
    return n80_round(q, ((uint64_t)guard << 63) | (r2 != 0 ? 1u : 0u), e);
}

// r + s * o  (the product rounded, then the sum rounded) -- the recursion's term.  Written without branches for the case
// that is all but universal there (the product and the sum are normal numbers): selects instead of per-lane jumps, which on
// the device cost more than the arithmetic they skip.  Anything else takes the two functions above.
VG_X80_HD VgN80 n80_muladd(VgN80 r, VgN80 s, VgN80 o)
{
    const uint64_t half = 1ULL << 63;
    const bool t_zero = s.m == 0 || o.m == 0;
    uint64_t hi, lo;
    x80_mul64(s.m, o.m, hi, lo);
    const bool up = (hi >> 63) != 0;                 // product of the significands in [2^127, 2^128)
    int32_t te = s.e + o.e - VG_X80_BIAS + (up ? 1 : 0);
    const uint64_t hi1 = (hi << 1) | (lo >> 63), lo1 = lo << 1;
    hi = up ? hi : hi1;
    lo = up ? lo : lo1;
    uint64_t inc = (lo | (hi & 1u)) > half ? 1u : 0u;
    uint64_t tm = hi + inc;
    const bool over = tm < inc;
    tm = over ? half : tm;
    te += over ? 1 : 0;
    bool slow = !t_zero && te < 1;

    const bool r_zero = r.m == 0;
    const bool r_big = r.e >= te;
    const uint64_t am = r_big ? r.m : tm, bm = r_big ? tm : r.m;
    int32_t e = r_big ? r.e : te;
    const uint32_t d = (uint32_t)(r_big ? r.e - te : te - r.e);
    const bool near = d < 64;
    const uint64_t bh = near ? bm >> (d & 63u) : 0u;
    lo = near ? (bm << 1) << ((63u - d) & 63u) : (d == 64 ? bm : 1u);
    hi = am + bh;
    const bool carry = hi < bh;
    const uint64_t lo2 = (lo >> 1) | (lo & 1u) | (hi << 63), hi2 = (hi >> 1) | half;
    lo = carry ? lo2 : lo;
    hi = carry ? hi2 : hi;
    e += carry ? 1 : 0;
    inc = (lo | (hi & 1u)) > half ? 1u : 0u;
    uint64_t m = hi + inc;
    const bool over2 = m < inc;
    m = over2 ? half : m;
    e += over2 ? 1 : 0;
    slow = slow || (!t_zero && !r_zero && e < 1);

    VgN80 out;
    out.m = t_zero ? r.m : (r_zero ? tm : m);
    out.e = t_zero ? r.e : (r_zero ? te : e);
    if (slow) out = n80_add(r, n80_mul(s, o));
    return out;
}

// a + b in the same style (the recursion's sum over a node's entries, which runs on the scalar unit: no jumps there either)
VG_X80_HD VgN80 n80_sum(VgN80 a, VgN80 b)
{
    const uint64_t half = 1ULL << 63;
    const bool a_zero = a.m == 0, b_zero = b.m == 0;
    const bool a_big = a.e >= b.e;
    const uint64_t am = a_big ? a.m : b.m, bm = a_big ? b.m : a.m;
    int32_t e = a_big ? a.e : b.e;
    const uint32_t d = (uint32_t)(a_big ? a.e - b.e : b.e - a.e);
    const bool near = d < 64;
    const uint64_t bh = near ? bm >> (d & 63u) : 0u;
    uint64_t lo = near ? (bm << 1) << ((63u - d) & 63u) : (d == 64 ? bm : 1u);
    uint64_t hi = am + bh;
    const bool carry = hi < bh;
    const uint64_t lo2 = (lo >> 1) | (lo & 1u) | (hi << 63), hi2 = (hi >> 1) | half;
    lo = carry ? lo2 : lo;
    hi = carry ? hi2 : hi;
    e += carry ? 1 : 0;
    const uint64_t inc = (lo | (hi & 1u)) > half ? 1u : 0u;
    uint64_t m = hi + inc;
    const bool over = m < inc;
    m = over ? half : m;
    e += over ? 1 : 0;
    VgN80 out;
    out.m = a_zero ? b.m : (b_zero ? a.m : m);
    out.e = a_zero ? b.e : (b_zero ? a.e : e);
    if (!a_zero && !b_zero && e < 1) out = n80_add(a, b);
    return out;
}

VG_X80_HD bool x80_is_zero(VgX80 v) { return v.m == 0; }

#endif
