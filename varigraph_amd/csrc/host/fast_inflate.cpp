#include "fast_inflate.hpp"

#include "inflate_core.hpp"

#include <immintrin.h>
#include <zlib.h>   // crc32: tail bytes, fallback and self-check of the carry-less-multiply path

#include <cstring>
#include <vector>

namespace vgh {

// ------------------------------------------------------------------------------------------------------------------
// CRC-32 by folding with PCLMULQDQ ("Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction",
// Gopal et al., Intel 2009; constants for the reflected gzip polynomial).  Checked against zlib's crc32 once per
// process; zlib's routine is used if the CPU lacks the instruction or the check fails.
// ------------------------------------------------------------------------------------------------------------------
namespace {

__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul_raw(const unsigned char* buf, size_t len, uint32_t crc)
{
    // len >= 64 and a multiple of 16; crc is the raw (inverted) register
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ULL, 0x01c6e41596ULL};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ULL, 0x00ccaa009eULL};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ULL, 0x0000000000ULL};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ULL, 0x01f7011641ULL};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00));
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20));
    x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k1k2));
    buf += 64;
    len -= 64;
    while (len >= 64) {   // four lanes of 16 bytes folded 64 bytes ahead
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30)));
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k3k4));   // four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);   // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(k5k0));
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(poly));   // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

bool clmul_usable()
{
    static const bool ok = [] {
        __builtin_cpu_init();
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        unsigned char t[64 + 48 + 16];
        for (size_t i = 0; i < sizeof t; ++i) t[i] = (unsigned char)(i * 37 + 11);
        for (size_t n : {(size_t)64, (size_t)80, sizeof t}) {
            const uint32_t want = (uint32_t)::crc32(0x1234abcdUL, t, (uInt)n);
            if (~crc32_clmul_raw(t, n, ~0x1234abcdu) != want) return false;
        }
        return true;
    }();
    return ok;
}

}  // namespace

uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t n)
{
    if (n >= 64 && clmul_usable()) {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32_clmul_raw(p, body, ~crc);
        p += body;
        n -= body;
    }
    while (n) {   // zlib takes a 32-bit length
        const size_t m = n < (1u << 30) ? n : (1u << 30);
        crc = (uint32_t)::crc32(crc, p, (uInt)m);
        p += m;
        n -= m;
    }
    return crc;
}

// ------------------------------------------------------------------------------------------------------------------
// DEFLATE
// ------------------------------------------------------------------------------------------------------------------
namespace {

using namespace inflate_core;

class Gunzip {
public:
    Gunzip(const GunzipIO& io, size_t cap) : io_(io), cap_(cap), ibuf_(kInBuf + kInPad), hist_(kHist)
    {
        in_ = in_end_ = ibuf_.data();
        std::memset(ibuf_.data(), 0, ibuf_.size());
        // fixed code of BTYPE 01
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        fixed_lit_.resize(kLitTable);
        build_table(l, 288, kSym.lit, kLitBits, fixed_lit_.data(), kLitTable);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        fixed_dist_.resize(kDistTable);
        build_table(d, 32, kSym.dist, kDistBits, fixed_dist_.data(), kDistTable);
        lit_.resize(kLitTable);
        dist_.resize(kDistTable);
    }

    GunzipEnd run()
    {
        GunzipEnd end = GunzipEnd::Clean;
        for (;;) {
            end = member();
            if (end != GunzipEnd::Clean) break;
            // another member?  (gz_look: fewer than two bytes, or no gzip magic = trailing garbage, ignored)
            if (!ensure(2) || in_[0] != 0x1f || in_[1] != 0x8b) break;
        }
        if (obase_ && out_ > obase_ && end != GunzipEnd::Stopped) io_.commit((size_t)(out_ - obase_));
        return end;
    }

private:
    enum class Step { Done, NeedIO, Bad, Truncated };

    // ---- input ----
    void fill_input()
    {
        if (file_eof_) return;
        const size_t rem = (size_t)(in_end_ - in_);
        std::memmove(ibuf_.data(), in_, rem);
        in_ = ibuf_.data();
        in_end_ = ibuf_.data() + rem;
        const size_t space = kInBuf - rem;
        const size_t n = io_.read(ibuf_.data() + rem, space);
        in_end_ += n;
        if (n < space) {
            file_eof_ = true;
            std::memset(ibuf_.data() + rem + n, 0, kInPad);
        }
    }
    bool ensure(size_t n)   // n <= a few KiB; byte-aligned state (nothing buffered in bitbuf_) or plain look-ahead
    {
        if ((size_t)(in_end_ - in_) < n) fill_input();
        return (size_t)(in_end_ - in_) >= n;
    }
    int get_byte() { return ensure(1) ? *in_++ : -1; }

    // ---- bits ----
    inline void refill()
    {
        uint64_t w;
        std::memcpy(&w, in_, 8);
        bitbuf_ |= w << bitcnt_;
        in_ += (63 - bitcnt_) >> 3;
        bitcnt_ |= 56;
    }
    inline bool overrun() const { return in_ > in_end_ && (uint64_t)(in_ - in_end_) * 8 > bitcnt_; }
    inline uint32_t take(uint32_t n)   // n <= 32, after a refill
    {
        const uint32_t v = (uint32_t)(bitbuf_ & ((1ULL << n) - 1));
        bitbuf_ >>= n;
        bitcnt_ -= n;
        return v;
    }
    void align_to_byte()
    {
        const uint32_t drop = bitcnt_ & 7;
        bitbuf_ >>= drop;
        bitcnt_ -= drop;
        in_ -= bitcnt_ >> 3;
        bitbuf_ = 0;
        bitcnt_ = 0;
    }
    // a block header (dynamic code lengths: < 400 bytes) must not run into the end of the buffer: compact + read more.
    // Bytes already inside bitbuf_ stay valid, in_ only moves together with its data.
    void top_up()
    {
        if ((size_t)(in_end_ - in_) < 2048 && !file_eof_) fill_input();
    }

    // ---- output ----
    void crc_up_to_out()
    {
        if (out_ > crc_from_) {
            crc_ = crc32_fast(crc_, crc_from_, (size_t)(out_ - crc_from_));
            isize_ += (uint64_t)(out_ - crc_from_);
            crc_from_ = out_;
        }
    }
    bool new_buffer()   // commits the current one
    {
        size_t h = 0, member_h = 0;
        if (obase_) {
            crc_up_to_out();
            h = hist_valid_ + (size_t)(out_ - obase_);
            if (h > kHist) h = kHist;
            member_h = (size_t)(out_ - win_start_);
            if (member_h > h) member_h = h;
            std::memcpy(hist_.data() + kHist - h, out_ - h, h);
            if (out_ > obase_) io_.commit((size_t)(out_ - obase_));
        }
        unsigned char* nb = io_.next_buffer(kHist, cap_);
        obase_ = nullptr;
        if (!nb) return false;
        std::memcpy(nb - h, hist_.data() + kHist - h, h);
        hist_valid_ = h;
        obase_ = out_ = nb;
        crc_from_ = nb;
        oend_ = nb + cap_;
        win_start_ = nb - member_h;
        return true;
    }

    // ---- gzip member ----
    GunzipEnd member()
    {
        // header (RFC 1952 2.3)
        unsigned char h[10];
        for (int i = 0; i < 10; ++i) {
            const int c = get_byte();
            if (c < 0) return GunzipEnd::Truncated;
            h[i] = (unsigned char)c;
        }
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) return GunzipEnd::Corrupt;
        const unsigned flg = h[3];
        if (flg & 4) {   // FEXTRA
            const int a = get_byte(), b = get_byte();
            if (a < 0 || b < 0) return GunzipEnd::Truncated;
            for (int n = a | (b << 8); n > 0; --n)
                if (get_byte() < 0) return GunzipEnd::Truncated;
        }
        for (unsigned bit : {8u, 16u})   // FNAME, FCOMMENT: zero-terminated
            if (flg & bit) {
                int c;
                while ((c = get_byte()) > 0) {}
                if (c < 0) return GunzipEnd::Truncated;
            }
        if (flg & 2) {   // FHCRC
            if (get_byte() < 0 || get_byte() < 0) return GunzipEnd::Truncated;
        }
        if (!obase_ && !new_buffer()) return GunzipEnd::Stopped;
        crc_up_to_out();   // nothing pending; keeps crc_from_ == out_
        crc_ = 0;
        isize_ = 0;
        win_start_ = out_;
        bitbuf_ = 0;
        bitcnt_ = 0;
        for (;;) {
            top_up();
            refill();
            const uint32_t final_block = take(1), type = take(2);
            Step s;
            if (type == 0) s = stored_block();
            else if (type == 1) s = huffman_block(fixed_lit_.data(), fixed_dist_.data());
            else if (type == 2) {
                s = read_dynamic_header();
                if (s == Step::Done) s = huffman_block(lit_.data(), dist_.data());
            } else s = Step::Bad;
            if (s == Step::NeedIO) return GunzipEnd::Stopped;
            if (s == Step::Truncated) return GunzipEnd::Truncated;
            if (s == Step::Bad) return GunzipEnd::Corrupt;
            if (overrun()) return GunzipEnd::Truncated;
            if (final_block) break;
        }
        align_to_byte();
        crc_up_to_out();
        if (!ensure(8)) return GunzipEnd::Truncated;
        const uint32_t crc = in_[0] | (in_[1] << 8) | (in_[2] << 16) | ((uint32_t)in_[3] << 24);
        const uint32_t isz = in_[4] | (in_[5] << 8) | (in_[6] << 16) | ((uint32_t)in_[7] << 24);
        in_ += 8;
        if (crc != crc_ || isz != (uint32_t)isize_) return GunzipEnd::Corrupt;
        return GunzipEnd::Clean;
    }

    Step stored_block()
    {
        align_to_byte();
        if (!ensure(4)) return Step::Truncated;
        const uint32_t len = in_[0] | (in_[1] << 8), nlen = in_[2] | (in_[3] << 8);
        in_ += 4;
        if ((len ^ 0xFFFFu) != nlen) return Step::Bad;
        uint32_t left = len;
        while (left) {
            if (in_ == in_end_) {
                fill_input();
                if (in_ == in_end_) return Step::Truncated;
            }
            if (out_ == oend_ && !new_buffer()) return Step::NeedIO;
            size_t n = left;
            if (n > (size_t)(in_end_ - in_)) n = (size_t)(in_end_ - in_);
            if (n > (size_t)(oend_ - out_)) n = (size_t)(oend_ - out_);
            std::memcpy(out_, in_, n);
            out_ += n;
            in_ += n;
            left -= (uint32_t)n;
        }
        return Step::Done;
    }

    Step read_dynamic_header()
    {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
        if (hlit > 286 || hdist > 30) return Step::Bad;
        uint8_t pre_lens[19] = {0};
        for (uint32_t i = 0; i < hclen; ++i) {
            if (bitcnt_ < 3) refill();
            pre_lens[order[i]] = (uint8_t)take(3);
        }
        uint32_t pre[1u << kPreBits];
        if (!build_table(pre_lens, 19, kSym.pre, kPreBits, pre, 1u << kPreBits)) return Step::Bad;
        uint8_t lens[286 + 30 + 140];
        uint32_t i = 0;
        const uint32_t total = hlit + hdist;
        while (i < total) {
            refill();
            const uint32_t e = pre[bitbuf_ & ((1u << kPreBits) - 1)];
            if (e_type(e) != T_LIT) return Step::Bad;
            bitbuf_ >>= e_nbits(e);
            bitcnt_ -= e_nbits(e);
            const uint32_t sym = e_pay(e);
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
            } else if (sym == 16) {
                if (i == 0) return Step::Bad;
                const uint32_t rep = 3 + take(2);
                std::memset(lens + i, lens[i - 1], rep);
                i += rep;
            } else {
                const uint32_t rep = sym == 17 ? 3 + take(3) : 11 + take(7);
                std::memset(lens + i, 0, rep);
                i += rep;
            }
            if (overrun()) return Step::Truncated;
        }
        if (i != total) return Step::Bad;
        if (lens[256] == 0) return Step::Bad;   // no end-of-block code
        if (!build_table(lens, hlit, kSym.lit, kLitBits, lit_.data(), kLitTable)) return Step::Bad;
        if (!build_table(lens + hlit, hdist, kSym.dist, kDistBits, dist_.data(), kDistTable)) return Step::Bad;
        return Step::Done;
    }

    // the symbols of one block; suspends for input / output at symbol boundaries
    Step huffman_block(const uint32_t* lit, const uint32_t* dist)
    {
        for (;;) {
            // careful mode near the end of the file: every symbol is checked against the real end of the data
            // before it produces output
            if ((size_t)(in_end_ - in_) < 64 + 8 && !file_eof_) fill_input();
            if (oend_ - out_ < (ptrdiff_t)kOutSlack && !new_buffer()) return Step::NeedIO;
            const bool near_end = (size_t)(in_end_ - in_) < 64 + 8 || in_ > in_end_;
            const Step s = near_end ? symbols<true>(lit, dist) : symbols<false>(lit, dist);
            if (s != Step::NeedIO) return s;
        }
    }

    template <bool CAREFUL>
    Step symbols(const uint32_t* lit, const uint32_t* dist)
    {
        const unsigned char* const in_safe = in_end_ - 64;       // fast mode: 8-byte loads stay inside the data
        unsigned char* const out_safe = oend_ - kOutSlack;
        uint64_t bitbuf = bitbuf_;
        uint32_t bitcnt = bitcnt_;
        const unsigned char* in = in_;
        unsigned char* out = out_;
        Step result = Step::NeedIO;
#define VG_REFILL()                                  \
    do {                                             \
        uint64_t w_;                                 \
        std::memcpy(&w_, in, 8);                     \
        bitbuf |= w_ << bitcnt;                      \
        in += (63 - bitcnt) >> 3;                    \
        bitcnt |= 56;                                \
    } while (0)
#define VG_OVERRUN() (in > in_end_ && (uint64_t)(in - in_end_) * 8 > bitcnt)
        for (;;) {
            if (CAREFUL) {
                if (in > in_end_ + kInPad - 16 || out > out_safe) break;
            } else if (in > in_safe || out > out_safe) {
                break;
            }
            VG_REFILL();
            uint32_t e = lit[bitbuf & ((1u << kLitBits) - 1)];
            if (e_type(e) == T_SUB) {
                bitbuf >>= kLitBits;
                bitcnt -= kLitBits;
                e = lit[e_pay(e) + (bitbuf & ((1u << e_extra(e)) - 1))];
            }
            bitbuf >>= e_nbits(e);
            bitcnt -= e_nbits(e);
            if (e_type(e) == T_LIT) {
                if (CAREFUL && VG_OVERRUN()) { result = Step::Truncated; break; }
                *out++ = (unsigned char)e_pay(e);
                if (CAREFUL) continue;
                // a second literal out of the same refill (at least 41 bits are left, a code takes at most 15)
                e = lit[bitbuf & ((1u << kLitBits) - 1)];
                if (e_type(e) == T_SUB) {
                    bitbuf >>= kLitBits;
                    bitcnt -= kLitBits;
                    e = lit[e_pay(e) + (bitbuf & ((1u << e_extra(e)) - 1))];
                }
                bitbuf >>= e_nbits(e);
                bitcnt -= e_nbits(e);
                if (e_type(e) == T_LIT) {
                    *out++ = (unsigned char)e_pay(e);
                    continue;
                }
                VG_REFILL();
            }
            if (e_type(e) == T_EOB) {
                result = (CAREFUL && VG_OVERRUN()) ? Step::Truncated : Step::Done;
                break;
            }
            if (e_type(e) != T_BASE) { result = Step::Bad; break; }
            // length, then distance: at most 5 + 15 + 13 bits, the refill above left at least 41
            uint32_t len = e_pay(e) + (uint32_t)(bitbuf & ((1u << e_extra(e)) - 1));
            bitbuf >>= e_extra(e);
            bitcnt -= e_extra(e);
            uint32_t d = dist[bitbuf & ((1u << kDistBits) - 1)];
            if (e_type(d) == T_SUB) {
                bitbuf >>= kDistBits;
                bitcnt -= kDistBits;
                d = dist[e_pay(d) + (bitbuf & ((1u << e_extra(d)) - 1))];
            }
            bitbuf >>= e_nbits(d);
            bitcnt -= e_nbits(d);
            if (e_type(d) != T_BASE) { result = Step::Bad; break; }
            const uint32_t distance = e_pay(d) + (uint32_t)(bitbuf & ((1u << e_extra(d)) - 1));
            bitbuf >>= e_extra(d);
            bitcnt -= e_extra(d);
            if (CAREFUL && VG_OVERRUN()) { result = Step::Truncated; break; }
            if (distance > (size_t)(out - win_start_)) { result = Step::Bad; break; }
            const unsigned char* src = out - distance;
            unsigned char* const end = out + len;
            if (distance >= 16) {
                do {
                    std::memcpy(out, src, 16);
                    out += 16;
                    src += 16;
                } while (out < end);
            } else if (distance == 1) {
                std::memset(out, *src, len);
            } else if (distance >= 8) {
                do {
                    std::memcpy(out, src, 8);
                    out += 8;
                    src += 8;
                } while (out < end);
            } else {
                do { *out++ = *src++; } while (out < end);
            }
            out = end;
        }
#undef VG_REFILL
#undef VG_OVERRUN
        bitbuf_ = bitbuf;
        bitcnt_ = bitcnt;
        in_ = in;
        out_ = out;
        if (result == Step::NeedIO && CAREFUL && in_ > in_end_ + kInPad - 16) return Step::Truncated;   // ran off the data
        return result;
    }

    const GunzipIO& io_;
    size_t cap_;
    std::vector<unsigned char> ibuf_, hist_;
    const unsigned char* in_ = nullptr;
    const unsigned char* in_end_ = nullptr;
    bool file_eof_ = false;
    uint64_t bitbuf_ = 0;
    uint32_t bitcnt_ = 0;
    unsigned char* obase_ = nullptr;
    unsigned char* out_ = nullptr;
    unsigned char* oend_ = nullptr;
    const unsigned char* crc_from_ = nullptr;
    const unsigned char* win_start_ = nullptr;
    size_t hist_valid_ = 0;
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
    std::vector<uint32_t> fixed_lit_, fixed_dist_, lit_, dist_;
};

}  // namespace

GunzipEnd fast_gunzip(const GunzipIO& io, size_t buffer_capacity)
{
    Gunzip g(io, buffer_capacity);
    return g.run();
}

}  // namespace vgh
