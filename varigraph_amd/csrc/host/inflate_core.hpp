// inflate_core.hpp -- what this repo's DEFLATE decoders share: table entry format, RFC 1951 base/extra tables and the
// canonical-code table builder (fast_inflate.cpp: one stream in order; par_gunzip.cpp: one ordinary gzip stream decoded by
// several threads).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace vgh {
namespace inflate_core {

constexpr size_t kHist = 32768;
constexpr size_t kInBuf = 1u << 20;
constexpr size_t kInPad = 64;        // zero bytes kept after the end of the data: the bit reader may run into them
constexpr size_t kOutSlack = 320;    // a match (258) plus the overshoot of its 16-byte copies
constexpr uint32_t kLitBits = 11, kDistBits = 8, kPreBits = 7;
constexpr size_t kLitTable = (1u << kLitBits) + 4608, kDistTable = (1u << kDistBits) + 4096;

// table entry: payload << 16 | extra bits << 8 | type << 5 | bits to consume
enum : uint32_t { T_LIT = 0, T_BASE = 1, T_EOB = 2, T_SUB = 3, T_BAD = 4 };
constexpr uint32_t mk(uint32_t payload, uint32_t type, uint32_t extra) { return payload << 16 | extra << 8 | type << 5; }
inline uint32_t e_nbits(uint32_t e) { return e & 31u; }
inline uint32_t e_type(uint32_t e) { return (e >> 5) & 7u; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31u; }
inline uint32_t e_pay(uint32_t e) { return e >> 16; }

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct SymEntries {
    uint32_t lit[288], dist[32], pre[19];
    SymEntries()
    {
        for (uint32_t s = 0; s < 256; ++s) lit[s] = mk(s, T_LIT, 0);
        lit[256] = mk(0, T_EOB, 0);
        for (uint32_t s = 257; s < 286; ++s) lit[s] = mk(kLenBase[s - 257], T_BASE, kLenExtra[s - 257]);
        lit[286] = lit[287] = mk(0, T_BAD, 0);
        for (uint32_t s = 0; s < 30; ++s) dist[s] = mk(kDistBase[s], T_BASE, kDistExtra[s]);
        dist[30] = dist[31] = mk(0, T_BAD, 0);
        for (uint32_t s = 0; s < 19; ++s) pre[s] = mk(s, T_LIT, 0);
    }
};
static const SymEntries kSym;

inline uint32_t bit_reverse(uint32_t code, uint32_t len)
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// canonical Huffman code (RFC 1951 3.2.2) -> lookup table indexed by the next main_bits stream bits, second-level
// tables for longer codes.  false: over-subscribed code or table space exhausted.
inline bool build_table(const uint8_t* lens, uint32_t n, const uint32_t* sym_entry, uint32_t main_bits, uint32_t* table, size_t table_cap)
{
    uint32_t count[16] = {0};
    for (uint32_t i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (uint32_t len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= (int)count[len];
        if (left < 0) return false;
    }
    uint32_t offs[17];
    offs[1] = 0;
    for (uint32_t len = 1; len <= 15; ++len) offs[len + 1] = offs[len] + count[len];
    uint16_t sorted[288];
    for (uint32_t i = 0; i < n; ++i)
        if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    const uint32_t main_size = 1u << main_bits;
    for (uint32_t i = 0; i < main_size; ++i) table[i] = mk(0, T_BAD, 0) | 1u;
    // longest code behind every first-level prefix
    uint8_t sub_bits[1u << kLitBits];
    std::memset(sub_bits, 0, main_size);
    uint32_t code = 0;
    for (uint32_t len = 1; len <= 15; ++len) {
        if (len > main_bits)
            for (uint32_t c = 0; c < count[len]; ++c) {
                const uint32_t prefix = bit_reverse(code + c, len) & (main_size - 1);
                if (len - main_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(len - main_bits);
            }
        code = (code + count[len]) << 1;
    }
    size_t next = main_size;
    for (uint32_t p = 0; p < main_size; ++p) {
        if (!sub_bits[p]) continue;
        const size_t sz = (size_t)1 << sub_bits[p];
        if (next + sz > table_cap || next > 0xFFFF) return false;
        table[p] = mk((uint32_t)next, T_SUB, sub_bits[p]) | main_bits;
        for (size_t i = 0; i < sz; ++i) table[next + i] = mk(0, T_BAD, 0) | 1u;
        next += sz;
    }
    code = 0;
    uint32_t idx = 0;
    for (uint32_t len = 1; len <= 15; ++len) {
        for (uint32_t c = 0; c < count[len]; ++c) {
            const uint32_t sym = sorted[idx++];
            const uint32_t rev = bit_reverse(code + c, len);
            if (len <= main_bits) {
                const uint32_t e = sym_entry[sym] | len;
                for (uint32_t r = rev; r < main_size; r += 1u << len) table[r] = e;
            } else {
                const uint32_t prefix = rev & (main_size - 1);
                const uint32_t sb = sub_bits[prefix], base = e_pay(table[prefix]);
                const uint32_t e = sym_entry[sym] | (len - main_bits);
                for (uint32_t r = rev >> main_bits; r < (1u << sb); r += 1u << (len - main_bits)) table[base + r] = e;
            }
        }
        code = (code + count[len]) << 1;
    }
    return true;
}

}  // namespace inflate_core
}  // namespace vgh
