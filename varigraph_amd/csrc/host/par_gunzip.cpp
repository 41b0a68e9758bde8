// par_gunzip.cpp -- see par_gunzip.hpp
#include "par_gunzip.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>   // crc32_combine

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "fast_inflate.hpp"
#include "inflate_core.hpp"

namespace vgh {
namespace {
using namespace inflate_core;

constexpr size_t kWin = 32768;           // DEFLATE window; also the number of prefix symbols in front of a run's output
constexpr size_t kSlack = 320;           // a match (258) plus the overshoot of its wide copies, in symbols
constexpr size_t kRoom = 1u << 16;       // free symbols asked for before a stretch of decoding
constexpr uint64_t kNoBit = ~0ULL;
constexpr size_t kMaxRun = (size_t)1 << 30;   // symbols per run (spans are sized for about 6 MiB of text)
constexpr size_t kPad = 4096;            // readable zero bytes behind the end of the mapped file

// ---- the file, mapped, with kPad zero bytes behind its end (the bit reader loads eight bytes at a time) ----------------
class Mapped {
public:
    static std::unique_ptr<Mapped> open(const std::string& path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) {
            ::close(fd);
            return nullptr;
        }
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t size = (size_t)st.st_size;
        const size_t total = (size + page - 1) / page * page + (kPad + page - 1) / page * page;
        void* base = mmap(nullptr, total, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == MAP_FAILED) {
            ::close(fd);
            return nullptr;
        }
        // the file over the front of the zero mapping: the rest of its last page and the pages behind stay zero
        if (mmap(base, size, PROT_READ, MAP_PRIVATE | MAP_FIXED, fd, 0) == MAP_FAILED) {
            munmap(base, total);
            ::close(fd);
            return nullptr;
        }
        ::close(fd);
        auto m = std::unique_ptr<Mapped>(new Mapped());
        m->data = static_cast<const uint8_t*>(base);
        m->size = size;
        m->total_ = total;
        return m;
    }
    ~Mapped() { munmap(const_cast<uint8_t*>(data), total_); }
    const uint8_t* data = nullptr;
    size_t size = 0;

private:
    Mapped() = default;
    size_t total_ = 0;
};

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---- bits ------------------------------------------------------------------------------------------------------------------
struct Bits {
    const uint8_t* in = nullptr;
    uint64_t buf = 0;
    uint32_t cnt = 0;
    inline void refill()
    {
        uint64_t w;
        std::memcpy(&w, in, 8);
        buf |= w << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
    }
    void seek(const uint8_t* data, uint64_t bit)
    {
        in = data + (bit >> 3);
        buf = 0;
        cnt = 0;
        refill();
        const uint32_t k = (uint32_t)(bit & 7);
        buf >>= k;
        cnt -= k;
    }
    inline uint32_t take(uint32_t n)   // n <= 32 and n <= cnt
    {
        const uint32_t v = (uint32_t)(buf & ((1ULL << n) - 1));
        buf >>= n;
        cnt -= n;
        return v;
    }
    inline uint64_t pos(const uint8_t* data) const { return (uint64_t)(in - data) * 8 - cnt; }
};

// ---- one run of decoded symbols: T = uint8_t with the known window in front, uint16_t with markers in front -------------------
struct MemberEnd {
    uint64_t out_pos;       // symbols of this run in front of the member's end
    uint32_t crc, isize;    // its trailer
};

template <class T>
struct Run {
    T* buf = nullptr;
    size_t cap = 0;         // symbols allocated
    size_t n = 0;           // symbols in use, the kWin prefix included
    size_t floor = 0;       // a match may not reach below this index (start of the member / of what is known of the window)
    uint64_t bit = 0;       // where the next block starts
    uint32_t blocks = 0;    // blocks completed
    std::vector<MemberEnd> ends;
    Run() = default;
    Run(const Run&) = delete;
    Run& operator=(const Run&) = delete;
    ~Run() { std::free(buf); }
    bool reserve(size_t room)
    {
        if (n + room + kSlack <= cap) return true;
        if (n + room > kMaxRun) return false;   // a span that inflates to more than this is not decoded in one piece
        size_t nc = cap + cap / 2;
        if (nc < n + room + kSlack + (1u << 20)) nc = n + room + kSlack + (1u << 20);
        T* nb = static_cast<T*>(std::realloc(buf, nc * sizeof(T)));
        if (!nb) return false;
        buf = nb;
        cap = nc;
        return true;
    }
    size_t out() const { return n - kWin; }
};

struct Tables {
    std::vector<uint32_t> fixed_lit, fixed_dist, lit, dist;
    Tables() : fixed_lit(kLitTable), fixed_dist(kDistTable), lit(kLitTable), dist(kDistTable)
    {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build_table(l, 288, kSym.lit, kLitBits, fixed_lit.data(), kLitTable);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, kSym.dist, kDistBits, fixed_dist.data(), kDistTable);
    }
};

enum class Step { Done, Again, Bad, Truncated, NoMem };
enum class RunEnd { Boundary, DataEnd, Bad, Truncated, NoMem };

// complete prefix code?  (single: exactly one code, of length 1 -- what zlib tolerates for distances)
bool code_complete(const uint8_t* lens, uint32_t n, bool* single)
{
    uint32_t sum = 0, used = 0, one = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (lens[i]) {
            sum += 32768u >> lens[i];
            ++used;
            one += lens[i] == 1;
        }
    if (single) *single = used == 1 && one == 1;
    return sum == 32768u;
}

// strict: the three codes must be complete (what every encoder writes): the test a guessed block start has to pass
Step read_dynamic_header(const Mapped& m, Bits& b, Tables& tb, bool strict)
{
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const uint64_t end_bits = (uint64_t)m.size * 8;
    b.refill();
    const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return Step::Bad;
    uint8_t pre_lens[19] = {0};
    for (uint32_t i = 0; i < hclen; ++i) {
        if (b.cnt < 3) b.refill();
        pre_lens[order[i]] = (uint8_t)b.take(3);
    }
    if (strict && !code_complete(pre_lens, 19, nullptr)) return Step::Bad;
    uint32_t pre[1u << kPreBits];
    if (!build_table(pre_lens, 19, kSym.pre, kPreBits, pre, 1u << kPreBits)) return Step::Bad;
    uint8_t lens[286 + 30 + 140];
    uint32_t i = 0;
    const uint32_t total = hlit + hdist;
    while (i < total) {
        b.refill();
        const uint32_t e = pre[b.buf & ((1u << kPreBits) - 1)];
        if (e_type(e) != T_LIT) return Step::Bad;
        b.buf >>= e_nbits(e);
        b.cnt -= e_nbits(e);
        const uint32_t sym = e_pay(e);
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
        } else if (sym == 16) {
            if (i == 0) return Step::Bad;
            const uint32_t rep = 3 + b.take(2);
            std::memset(lens + i, lens[i - 1], rep);
            i += rep;
        } else {
            const uint32_t rep = sym == 17 ? 3 + b.take(3) : 11 + b.take(7);
            std::memset(lens + i, 0, rep);
            i += rep;
        }
        if (b.pos(m.data) > end_bits) return Step::Truncated;
    }
    if (i != total) return Step::Bad;
    if (lens[256] == 0) return Step::Bad;   // no end-of-block code
    if (strict) {
        bool single = false;
        if (!code_complete(lens, hlit, nullptr)) return Step::Bad;
        if (!code_complete(lens + hlit, hdist, &single) && !single) return Step::Bad;
    }
    if (!build_table(lens, hlit, kSym.lit, kLitBits, tb.lit.data(), kLitTable)) return Step::Bad;
    if (!build_table(lens + hlit, hdist, kSym.dist, kDistBits, tb.dist.data(), kDistTable)) return Step::Bad;
    return Step::Done;
}

template <class T>
Step stored_block(const Mapped& m, Bits& b, Run<T>& r)
{
    uint64_t byte = (b.pos(m.data) + 7) >> 3;
    if (byte + 4 > m.size) return Step::Truncated;
    const uint32_t len = m.data[byte] | (m.data[byte + 1] << 8), nlen = m.data[byte + 2] | (m.data[byte + 3] << 8);
    byte += 4;
    if ((len ^ 0xFFFFu) != nlen) return Step::Bad;
    const size_t have = (size_t)std::min<uint64_t>(len, m.size - byte);
    if (!r.reserve(have)) return Step::NoMem;
    for (size_t i = 0; i < have; ++i) r.buf[r.n + i] = m.data[byte + i];
    r.n += have;
    byte += have;
    b.seek(m.data, byte * 8);
    return have == len ? Step::Done : Step::Truncated;   // a cut-off stored block: what is there is delivered
}

// The symbols of one block (fast_inflate.cpp's loop over a mapped file and either symbol width).  CAREFUL near the end of
// the file: every symbol is checked against the real end of the data before it produces output.  Again: more room or the
// other mode is needed.
template <bool CAREFUL, class T>
Step symbols(const Mapped& m, Bits& bits, Run<T>& r, const uint32_t* lit, const uint32_t* dist)
{
    const uint8_t* const end = m.data + m.size;
    const uint8_t* const in_safe = m.size >= 64 ? end - 64 : m.data;
    T* const base = r.buf;
    T* const out_safe = r.buf + r.cap - kSlack;
    const T* const floor = r.buf + r.floor;
    uint64_t bitbuf = bits.buf;
    uint32_t bitcnt = bits.cnt;
    const uint8_t* in = bits.in;
    T* out = r.buf + r.n;
    Step result = Step::Again;
#define VG_REFILL()                                  \
    do {                                             \
        uint64_t w_;                                 \
        std::memcpy(&w_, in, 8);                     \
        bitbuf |= w_ << bitcnt;                      \
        in += (63 - bitcnt) >> 3;                    \
        bitcnt |= 56;                                \
    } while (0)
#define VG_OVERRUN() (in > end && (uint64_t)(in - end) * 8 > bitcnt)
    for (;;) {
        if (CAREFUL) {
            if (in > end + 64 || out > out_safe) break;
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
            *out++ = (T)e_pay(e);
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
                *out++ = (T)e_pay(e);
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
        const uint32_t len = e_pay(e) + (uint32_t)(bitbuf & ((1u << e_extra(e)) - 1));
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
        if (distance > (size_t)(out - floor)) { result = Step::Bad; break; }
        const T* src = out - distance;
        T* const stop = out + len;
        constexpr uint32_t kWide = 16 / sizeof(T);   // symbols per 16-byte copy
        if (distance >= kWide) {
            do {
                std::memcpy(out, src, 16);
                out += kWide;
                src += kWide;
            } while (out < stop);
        } else if (distance == 1) {
            const T v = *src;
            for (T* q = out; q < stop; ++q) *q = v;
        } else {
            do { *out++ = *src++; } while (out < stop);
        }
        out = stop;
    }
#undef VG_REFILL
#undef VG_OVERRUN
    bits.buf = bitbuf;
    bits.cnt = bitcnt;
    bits.in = in;
    r.n = (size_t)(out - base);
    if (result == Step::Again && CAREFUL && in > end + 64) return Step::Truncated;   // ran off the data
    return result;
}

template <class T>
Step huffman_block(const Mapped& m, Bits& b, Run<T>& r, const uint32_t* lit, const uint32_t* dist)
{
    for (;;) {
        if (!r.reserve(kRoom)) return Step::NoMem;
        const bool near_end = m.size < 64 || b.in > m.data + m.size - 64;
        const Step s = near_end ? symbols<true>(m, b, r, lit, dist) : symbols<false>(m, b, r, lit, dist);
        if (s != Step::Again) return s;
    }
}

enum class Hdr { Ok, Truncated, Bad };
// gzip member header (RFC 1952 2.3) at byte `at`; `after` = first byte of the DEFLATE data
Hdr member_header(const Mapped& m, size_t at, size_t& after)
{
    if (m.size - at < 10) return Hdr::Truncated;
    const uint8_t* h = m.data + at;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) return Hdr::Bad;
    const unsigned flg = h[3];
    size_t p = at + 10;
    if (flg & 4) {   // FEXTRA
        if (m.size - p < 2) return Hdr::Truncated;
        const size_t n = m.data[p] | (m.data[p + 1] << 8);
        p += 2;
        if (m.size - p < n) return Hdr::Truncated;
        p += n;
    }
    for (unsigned bit : {8u, 16u})   // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            while (p < m.size && m.data[p]) ++p;
            if (p >= m.size) return Hdr::Truncated;
            ++p;
        }
    if (flg & 2) {   // FHCRC
        if (m.size - p < 2) return Hdr::Truncated;
        p += 2;
    }
    after = p;
    return Hdr::Ok;
}

// Blocks from r.bit on until a block starts at or behind stop_bit (Boundary), max_blocks are done (Boundary), the data
// ends behind a member (DataEnd) or goes wrong.  Member ends are followed: trailer noted, next header read, window reset.
// GUESS (markers in front, the start was guessed): the codes must be complete, and bytes behind a member that are neither
// a gzip header nor the end of the file end the guess instead of the data.
template <bool GUESS, class T>
RunEnd decode_blocks(const Mapped& m, Run<T>& r, uint64_t stop_bit, uint32_t max_blocks, Tables& tb)
{
    const uint64_t end_bits = (uint64_t)m.size * 8;
    for (;;) {
        if (r.bit >= stop_bit || r.blocks >= max_blocks) return RunEnd::Boundary;
        if (r.bit + 3 > end_bits) return RunEnd::Truncated;
        Bits b;
        b.seek(m.data, r.bit);
        const uint32_t final_block = b.take(1), type = b.take(2);
        Step s;
        if (type == 0) s = stored_block(m, b, r);
        else if (type == 1) s = huffman_block(m, b, r, tb.fixed_lit.data(), tb.fixed_dist.data());
        else if (type == 2) {
            s = read_dynamic_header(m, b, tb, GUESS);
            if (s == Step::Done) s = huffman_block(m, b, r, tb.lit.data(), tb.dist.data());
        } else s = Step::Bad;
        if (s == Step::NoMem) return RunEnd::NoMem;
        if (s == Step::Truncated) return RunEnd::Truncated;
        if (s == Step::Bad) return RunEnd::Bad;
        const uint64_t p = b.pos(m.data);
        if (p > end_bits) return RunEnd::Truncated;
        r.bit = p;
        r.blocks++;
        if (!final_block) continue;
        size_t byte = (size_t)((p + 7) >> 3);
        if (m.size - byte < 8) return RunEnd::Truncated;   // no trailer: the member's bytes stand, unchecked (as gzread's do)
        r.ends.push_back(MemberEnd{r.out(), le32(m.data + byte), le32(m.data + byte + 4)});
        byte += 8;
        r.bit = (uint64_t)byte * 8;
        // another member?  (gz_look: fewer than two bytes, or no gzip magic = trailing garbage, ignored)
        if (m.size - byte < 2 || m.data[byte] != 0x1f || m.data[byte + 1] != 0x8b) {
            if (GUESS && byte != m.size) return RunEnd::Bad;
            return RunEnd::DataEnd;
        }
        size_t after = 0;
        const Hdr h = member_header(m, byte, after);
        if (h == Hdr::Truncated) return RunEnd::Truncated;
        if (h == Hdr::Bad) return RunEnd::Bad;
        r.bit = (uint64_t)after * 8;
        r.floor = r.n;
    }
}

// First bit position in [from, to) where a dynamic-code block (not the last of its member) starts whose header describes
// three complete codes and which, with the two blocks behind it, decodes.  The run then holds those blocks.
struct Kraft9 {            // sum of 128 >> l over three 3-bit code lengths (l = 0: unused)
    uint16_t v[512];
    Kraft9()
    {
        for (uint32_t i = 0; i < 512; ++i) {
            uint32_t s = 0;
            for (uint32_t k = 0; k < 3; ++k) {
                const uint32_t l = (i >> (3 * k)) & 7u;
                if (l) s += 128u >> l;
            }
            v[i] = (uint16_t)s;
        }
    }
};
const Kraft9 kraft9;

uint64_t find_start(const Mapped& m, uint64_t from, uint64_t to, Run<uint16_t>& r, Tables& tb, RunEnd& trial)
{
    trial = RunEnd::Bad;
    const uint64_t end_bits = (uint64_t)m.size * 8;
    if (to + 80 > end_bits) to = end_bits > 80 ? end_bits - 80 : 0;
    for (uint64_t p = from; p < to; ++p) {
        uint64_t w;
        std::memcpy(&w, m.data + (p >> 3), 8);
        w >>= (p & 7);
        // BFINAL 0, BTYPE 10, HLIT <= 29, HDIST <= 29
        if ((w & 7) != 4 || ((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;
        const uint32_t hclen = (uint32_t)((w >> 13) & 15) + 4;
        uint64_t w2;
        std::memcpy(&w2, m.data + ((p + 17) >> 3), 8);
        w2 >>= ((p + 17) & 7);   // 57 bits = 19 code lengths of 3 bits
        // Kraft sum of the code-length code, three lengths per table look-up
        w2 &= (1ULL << (3 * hclen)) - 1;
        uint32_t sum = 0;
        for (uint32_t c = 0; c < 7; ++c) sum += kraft9.v[(w2 >> (9 * c)) & 511u];
        if (sum != 128u) continue;
        r.n = kWin;
        r.floor = 0;
        r.bit = p;
        r.blocks = 0;
        r.ends.clear();
        const RunEnd e = decode_blocks<true>(m, r, kNoBit, 3, tb);
        trial = e;
        if (e == RunEnd::NoMem) return kNoBit;
        if (e == RunEnd::Boundary && r.blocks == 3) return p;
        if ((e == RunEnd::DataEnd || e == RunEnd::Truncated) && r.blocks >= 1) return p;   // the file ends here
    }
    return kNoBit;
}

// markers -> bytes, in place (byte i is written after symbol i was read).  window: the kWin bytes in front; markers below
// win_floor point at bytes that do not exist (in front of the member's start): the guess cannot be used.
bool resolve(uint16_t* sym, size_t n, const uint8_t* window, size_t win_floor)
{
    uint8_t* out = reinterpret_cast<uint8_t*>(sym);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint64_t q;
        std::memcpy(&q, sym + i, 8);
        if (!(q & 0x8000800080008000ULL)) {
            out[i] = (uint8_t)q;
            out[i + 1] = (uint8_t)(q >> 16);
            out[i + 2] = (uint8_t)(q >> 32);
            out[i + 3] = (uint8_t)(q >> 48);
            continue;
        }
        for (size_t k = 0; k < 4; ++k) {
            const uint16_t s = (uint16_t)(q >> (16 * k));
            if (s & 0x8000u) {
                const size_t w = s & 0x7FFFu;
                if (w < win_floor) return false;
                out[i + k] = window[w];
            } else {
                out[i + k] = (uint8_t)s;
            }
        }
    }
    for (; i < n; ++i) {
        uint16_t s;
        std::memcpy(&s, sym + i, 2);
        if (s & 0x8000u) {
            const size_t w = s & 0x7FFFu;
            if (w < win_floor) return false;
            out[i] = window[w];
        } else {
            out[i] = (uint8_t)s;
        }
    }
    return true;
}

// the same with every marker known to be good: one table look-up per symbol (lut[v] = v for bytes, lut[0x8000 + w] = window[w])
void resolve_all(uint16_t* sym, size_t n, const uint8_t* lut)
{
    uint8_t* out = reinterpret_cast<uint8_t*>(sym);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint64_t q;
        std::memcpy(&q, sym + i, 8);
        const uint8_t b0 = lut[(uint16_t)q], b1 = lut[(uint16_t)(q >> 16)], b2 = lut[(uint16_t)(q >> 32)], b3 = lut[(uint16_t)(q >> 48)];
        out[i] = b0;
        out[i + 1] = b1;
        out[i + 2] = b2;
        out[i + 3] = b3;
    }
    for (; i < n; ++i) {
        uint16_t v;
        std::memcpy(&v, sym + i, 2);
        out[i] = lut[v];
    }
}

enum class EndKind { None, Clean, Corrupt, Truncated, NoMem };

struct Seam {               // what a span hands to the one behind it
    uint64_t for_chunk = kNoBit;
    uint64_t end_bit = 0;   // where the next block starts
    bool data_end = false;
    std::vector<uint8_t> window = std::vector<uint8_t>(kWin);   // the last kWin bytes in front of end_bit ...
    size_t win_valid = 0;                                        // ... of which this many belong to the current member
};

struct Segment {            // bytes of a span between member ends
    size_t n;
    uint32_t crc;
    bool member_end;
    uint32_t want_crc, want_isize;
};

struct Chunk {
    uint64_t index = kNoBit;
    bool final_ = false;
    Run<uint16_t> spec;     // guessed part (its output is resolved in place)
    Run<uint8_t> exact;     // part decoded with the window known
    std::vector<uint8_t> joined;
    std::vector<uint8_t> lut;   // marker / byte -> byte for this span's window
    const uint8_t* payload = nullptr;
    size_t n = 0;
    std::vector<Segment> segs;
    EndKind end = EndKind::None;
    bool void_ = false;     // behind the end of the data
    Seam seam_in;
};

class ParGunzipSource final : public ByteSource {
public:
    ParGunzipSource(std::unique_ptr<Mapped> m, std::string path, unsigned threads, size_t span)
        : m_(std::move(m)), path_(std::move(path)), span_(span), n_spans_((m_->size + span - 1) / span), ring_(threads + 2), chunks_(threads + 2)
    {
        for (unsigned t = 0; t < threads; ++t) workers_.emplace_back([this] { work(); });
    }
    ~ParGunzipSource() override
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            cancel_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
        if (getenv("VGH_PAR_GUNZIP_STATS"))
            std::fprintf(stderr, "[par_gunzip] %s: %llu spans, %llu joined at the guessed start, %llu no start found, %llu decoded again in order\n",
                         path_.c_str(), (unsigned long long)n_spans_, (unsigned long long)st_joined_, (unsigned long long)st_nostart_,
                         (unsigned long long)st_again_),
            std::fprintf(stderr, "[par_gunzip] seconds summed over spans: search %.3f, guessed decode %.3f, waiting for the seam %.3f, resolve %.3f, crc %.3f\n",
                         t_search_, t_spec_, t_wait_, t_resolve_, t_crc_);
    }
    const char* kind() const override { return "gzip"; }

    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (holding_) {   // the span handed out last is done with
                holding_ = false;
                ++released_;
                cv_.notify_all();
            }
            if (ended_ || deliver_ >= n_spans_) return false;
            Chunk& c = chunks_[deliver_ % ring_];
            cv_.wait(lk, [&] { return c.index == deliver_ && c.final_; });
            ++deliver_;
            holding_ = true;
            if (c.void_) {
                ended_ = true;
                continue;
            }
            // member checks, in stream order: CRC-32 and length of every member that ends in this span
            size_t good = 0;
            bool damaged = false;
            for (const Segment& s : c.segs) {
                run_crc_ = (uint32_t)crc32_combine(run_crc_, s.crc, (z_off_t)s.n);
                run_len_ += s.n;
                good += s.n;
                if (s.member_end) {
                    if (run_crc_ != s.want_crc || (uint32_t)run_len_ != s.want_isize) {
                        damaged = true;
                        break;   // the member's bytes are delivered (a serial decoder has passed them on by now), nothing after
                    }
                    run_crc_ = 0;
                    run_len_ = 0;
                }
            }
            EndKind end = c.end;
            if (damaged) end = EndKind::Corrupt;
            if (end != EndKind::None) {
                ended_ = true;
                if (end == EndKind::NoMem) throw std::runtime_error("'" + path_ + "': out of memory while decompressing.");
                if (end == EndKind::Corrupt || end == EndKind::Truncated)
                    std::fprintf(stderr, "[varigraph-mi] warning: '%s': gzip stream %s; only what decoded cleanly is used\n", path_.c_str(),
                                 end == EndKind::Corrupt ? "is damaged (bad block, CRC-32 or length)" : "ends inside a member");
            }
            const size_t deliver = damaged ? good : c.n;
            if (deliver == 0) continue;
            p = c.payload;
            n = deliver;
            return true;
        }
    }

private:
    void work()
    {
        Tables tb;
        for (;;) {
            uint64_t i;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return cancel_ || data_ended_ || next_ >= n_spans_ || next_ < released_ + ring_; });
                if (cancel_ || data_ended_ || next_ >= n_spans_) return;
                i = next_++;
            }
            Chunk& c = chunks_[i % ring_];
            {
                std::lock_guard<std::mutex> lk(mu_);
                c.index = i;
                c.final_ = false;
            }
            c.payload = nullptr;
            c.n = 0;
            c.segs.clear();
            c.end = EndKind::None;
            c.void_ = false;
            process(i, c, tb);
            {
                std::lock_guard<std::mutex> lk(mu_);
                c.final_ = true;
                if (c.end != EndKind::None || c.void_) data_ended_ = true;
            }
            cv_.notify_all();
        }
    }

    // hands the seam to the span behind (its slot may still hold an older span: the seam fields are not that span's)
    void pass_seam(uint64_t i, const Seam& s)
    {
        if (i + 1 >= n_spans_) return;
        Chunk& nx = chunks_[(i + 1) % ring_];
        {
            std::lock_guard<std::mutex> lk(mu_);
            nx.seam_in.end_bit = s.end_bit;
            nx.seam_in.data_end = s.data_end;
            nx.seam_in.win_valid = s.win_valid;
            nx.seam_in.window = s.window;
            nx.seam_in.for_chunk = i + 1;
        }
        cv_.notify_all();
    }

    void process(uint64_t i, Chunk& c, Tables& tb)
    {
        const Mapped& m = *m_;
        const uint64_t span_to = (i + 1) * (uint64_t)span_ * 8;
        // ---- side by side: guess a start inside this span and decode from it, the window unknown
        uint64_t cand = kNoBit;
        RunEnd spec_end = RunEnd::Bad;
        if (i > 0) {
            if (!c.spec.buf) {
                if (!c.spec.reserve(kWin + kRoom)) {
                    finish_nomem(i, c);
                    return;
                }
                for (size_t j = 0; j < kWin; ++j) c.spec.buf[j] = (uint16_t)(0x8000u + j);
            }
            RunEnd trial;
            const auto t0 = now();
            cand = find_start(m, i * (uint64_t)span_ * 8, span_to, c.spec, tb, trial);
            const auto t1 = now();
            if (cand != kNoBit) spec_end = trial == RunEnd::Boundary ? decode_blocks<true>(m, c.spec, span_to, ~0u, tb) : trial;
            add_time(t_search_, t0, t1);
            add_time(t_spec_, t1, now());
        }
        const bool spec_ok = cand != kNoBit && (spec_end == RunEnd::Boundary || spec_end == RunEnd::DataEnd);

        // ---- in order: the seam of the span in front
        Seam seam;
        if (i == 0) {
            size_t after = 0;
            const Hdr h = member_header(m, 0, after);
            if (h != Hdr::Ok) {
                c.end = h == Hdr::Bad ? EndKind::Corrupt : EndKind::Truncated;
                seam.data_end = true;
                pass_seam(i, seam);
                return;
            }
            seam.end_bit = (uint64_t)after * 8;
        } else {
            const auto t0 = now();
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return cancel_ || c.seam_in.for_chunk == i; });
            t_wait_ += std::chrono::duration<double>(now() - t0).count();
            if (cancel_) {
                c.void_ = true;
                return;
            }
            seam.end_bit = c.seam_in.end_bit;
            seam.data_end = c.seam_in.data_end;
            seam.win_valid = c.seam_in.win_valid;
            seam.window.swap(c.seam_in.window);
            c.seam_in.window.resize(kWin);
        }
        if (seam.data_end) {
            c.void_ = true;
            pass_seam(i, seam);
            return;
        }

        // exact part: from the seam to the guessed start (nothing, as a rule), or the whole span when the guess is no use
        Run<uint8_t>& ex = c.exact;
        if (!ex.reserve(kWin + kRoom)) {
            finish_nomem(i, c);
            return;
        }
        std::memcpy(ex.buf, seam.window.data(), kWin);
        ex.n = kWin;
        ex.floor = kWin - seam.win_valid;
        ex.bit = seam.end_bit;
        ex.blocks = 0;
        ex.ends.clear();
        RunEnd end = RunEnd::Boundary;
        bool use_spec = false, resolve_later = false;
        if (spec_ok && cand >= seam.end_bit) {
            if (cand > seam.end_bit) end = decode_blocks<false>(m, ex, cand, ~0u, tb);
            if (end == RunEnd::Boundary && ex.bit == cand) {
                // the window of the guessed part: the last kWin bytes in front of it.  With all of it belonging to the
                // member every marker is good, and only the span's last kWin symbols have to be resolved before the next
                // span can go on; else (a member began less than 32 KiB ago) the markers are checked first.
                const uint8_t* win = ex.buf + ex.n - kWin;
                const size_t win_floor = ex.floor > ex.n - kWin ? ex.floor - (ex.n - kWin) : 0;
                if (win_floor == 0) {
                    if (c.lut.empty()) {
                        c.lut.resize(65536);
                        for (size_t v = 0; v < 256; ++v) c.lut[v] = (uint8_t)v;
                    }
                    std::memcpy(c.lut.data() + 0x8000, win, kWin);
                    use_spec = resolve_later = true;
                } else {
                    const auto t0 = now();
                    use_spec = resolve(c.spec.buf + kWin, c.spec.out(), win, win_floor);
                    add_time(t_resolve_, t0, now());
                }
            }
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (use_spec) ++st_joined_;
            else if (i > 0 && cand == kNoBit) ++st_nostart_;
            else if (i > 0) ++st_again_;
        }
        if (use_spec) {
            end = spec_end;
        } else if (end == RunEnd::Boundary) {
            end = decode_blocks<false>(m, ex, span_to, ~0u, tb);
        }
        if (end == RunEnd::NoMem) {
            finish_nomem(i, c);
            return;
        }

        // ---- the seam for the span behind: end position, member state, the last kWin bytes
        const size_t n_ex = ex.out(), n_sp = use_spec ? c.spec.out() : 0;
        c.n = n_ex + n_sp;
        std::vector<MemberEnd> ends = ex.ends;
        if (use_spec)
            for (const MemberEnd& e : c.spec.ends) ends.push_back(MemberEnd{e.out_pos + n_ex, e.crc, e.isize});
        Seam out;
        out.end_bit = use_spec ? c.spec.bit : ex.bit;
        if (end == RunEnd::DataEnd) c.end = EndKind::Clean;
        else if (end == RunEnd::Bad) c.end = EndKind::Corrupt;
        else if (end == RunEnd::Truncated) c.end = EndKind::Truncated;
        out.data_end = c.end != EndKind::None;
        if (!out.data_end) {
            const size_t since = ends.empty() ? seam.win_valid + c.n : c.n - (size_t)ends.back().out_pos;   // bytes of the current member so far
            out.win_valid = since < kWin ? since : kWin;
            size_t take = kWin;                       // filled from the back: guessed part, exact part, the old window
            const size_t t_sp = n_sp < take ? n_sp : take;
            uint8_t* dst = out.window.data();
            if (t_sp) {
                const uint16_t* sym = c.spec.buf + kWin + (n_sp - t_sp);
                if (resolve_later)
                    for (size_t j = 0; j < t_sp; ++j) dst[kWin - t_sp + j] = c.lut[sym[j]];
                else std::memcpy(dst + kWin - t_sp, reinterpret_cast<const uint8_t*>(c.spec.buf + kWin) + (n_sp - t_sp), t_sp);
            }
            take -= t_sp;
            const size_t t_ex = n_ex < take ? n_ex : take;
            if (t_ex) std::memcpy(dst + take - t_ex, ex.buf + kWin + (n_ex - t_ex), t_ex);
            take -= t_ex;
            if (take) std::memcpy(dst, seam.window.data() + (kWin - take), take);
        }
        pass_seam(i, out);

        // ---- side by side again: the span's bytes and the CRC-32 of its pieces
        if (resolve_later) {
            const auto t0 = now();
            resolve_all(c.spec.buf + kWin, n_sp, c.lut.data());
            add_time(t_resolve_, t0, now());
        }
        const uint8_t* sp_bytes = reinterpret_cast<const uint8_t*>(c.spec.buf + kWin);
        if (n_ex && n_sp) {
            c.joined.resize(n_ex + n_sp);
            std::memcpy(c.joined.data(), ex.buf + kWin, n_ex);
            std::memcpy(c.joined.data() + n_ex, sp_bytes, n_sp);
            c.payload = c.joined.data();
        } else {
            c.payload = n_sp ? sp_bytes : ex.buf + kWin;
        }
        size_t at = 0;
        const auto t_crc0 = now();
        for (const MemberEnd& e : ends) {
            c.segs.push_back(Segment{(size_t)e.out_pos - at, crc32_fast(0, c.payload + at, (size_t)e.out_pos - at), true, e.crc, e.isize});
            at = (size_t)e.out_pos;
        }
        if (at < c.n) c.segs.push_back(Segment{c.n - at, crc32_fast(0, c.payload + at, c.n - at), false, 0, 0});
        add_time(t_crc_, t_crc0, now());
    }

    static std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }
    void add_time(double& acc, std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b)
    {
        std::lock_guard<std::mutex> lk(mu_);
        acc += std::chrono::duration<double>(b - a).count();
    }

    void finish_nomem(uint64_t i, Chunk& c)
    {
        c.end = EndKind::NoMem;
        c.n = 0;
        Seam s;
        s.data_end = true;
        pass_seam(i, s);
    }

    std::unique_ptr<Mapped> m_;
    std::string path_;
    size_t span_;
    uint64_t n_spans_;
    uint64_t ring_;
    std::vector<Chunk> chunks_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    uint64_t next_ = 0;        // next span a worker takes
    uint64_t released_ = 0;    // spans the consumer is done with
    uint64_t deliver_ = 0;     // next span the consumer takes
    bool holding_ = false, ended_ = false, cancel_ = false, data_ended_ = false;
    uint32_t run_crc_ = 0;
    uint64_t run_len_ = 0;
    uint64_t st_joined_ = 0, st_nostart_ = 0, st_again_ = 0;
    double t_search_ = 0, t_spec_ = 0, t_wait_ = 0, t_resolve_ = 0, t_crc_ = 0;
};

}  // namespace

std::unique_ptr<ByteSource> open_parallel_gunzip(const std::string& path, unsigned threads, size_t span_bytes)
{
    if (threads < 2) return nullptr;
    auto m = Mapped::open(path);
    if (!m || m->size < 18 || m->data[0] != 0x1f || m->data[1] != 0x8b) return nullptr;
    size_t span = span_bytes;
    if (!span) {
        // compressed bytes that hold about 6 MiB of text (measured best: 3.5-7 MiB per span), from the ratio of the first
        // blocks: 1 MiB for FASTQ at gzip -6, far less for text that inflates a thousandfold (a span's symbols are held
        // in memory)
        span = (size_t)2 << 20;
        size_t after = 0;
        if (member_header(*m, 0, after) == Hdr::Ok) {
            Run<uint8_t> probe;
            Tables tb;
            if (probe.reserve(kWin + kRoom)) {
                probe.n = probe.floor = kWin;
                probe.bit = (uint64_t)after * 8;
                (void)decode_blocks<false>(*m, probe, probe.bit + ((uint64_t)256 << 13), 64, tb);
                const double in_bytes = (double)(probe.bit / 8 - after);
                if (in_bytes >= 1024 && probe.out() > 0) {
                    const double want = 6.0 * 1048576.0 * in_bytes / (double)probe.out();
                    span = want > 2097152.0 ? (size_t)2 << 20 : (want < 65536.0 ? (size_t)65536 : (size_t)want);
                }
            }
        }
    }
    if (const char* e = getenv("VGH_PARGZ_SPAN_KB"))
        if (atoi(e) >= 1) span = (size_t)atoi(e) << 10;
    if (span < 1024) span = 1024;
    if (threads > 64) threads = 64;
    return std::make_unique<ParGunzipSource>(std::move(m), path, threads, span);
}

}  // namespace vgh
