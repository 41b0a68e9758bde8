// vgmi_inflate_dev.h -- what the two DEFLATE decoders of the device share: the per-wavefront tables in LDS, their construction from
// code lengths (RFC 1951 3.2.2), the packed entries a batch of 64 bit positions is decoded from, the bit-by-bit path of long codes.
// Used by vgmi_inflate.hip (block-gzip members: one wavefront per member, bytes out) and vgmi_gunzip.hip (ordinary gzip streams: one
// wavefront per stretch between guessed block starts, symbols with back-reference placeholders out).
#ifndef VGMI_INFLATE_DEV_H
#define VGMI_INFLATE_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vgk {

#define INF_LIT_BITS 10u
#define INF_DIST_BITS 8u
#define INF_MAXBITS 15
#define INF_RING 2048u          // bytes of output kept in LDS (a power of two)
#define INF_BATCH_OUT 384u      // a batch stops taking symbols once it has produced this much
#define INF_NEAR 1280u          // matches up to this far back read the ring (INF_NEAR + INF_BATCH_OUT < INF_RING: a batch's literals, written
                                // first, never land on a source); farther ones read global memory, where everything older than the
                                // 258 unflushed bytes already is (INF_NEAR - 258 > INF_BATCH_OUT + 258)
#define INF_WAVES 4u            // wavefronts per workgroup (8.3 KB of LDS each: four workgroups = 16 wavefronts per CU)

template <class RingT, uint32_t RING = INF_RING>
struct InfTablesT {                // per wavefront, in LDS
    uint32_t lit[1u << INF_LIT_BITS];     // while a table is built: symbol << 4 | code length (0: code longer than the index / unused);
                                          // then packed (inf_pack_lit): bits 0..3 code bits taken, 4..5 kind (0 literals, 1 length, 2 end of block),
                                          // literals: 6..7 how many (1..3), 8..31 the bytes; length: 6..8 extra bits, 9..17 base length
    uint32_t dist[1u << INF_DIST_BITS];   // built likewise (also serves the code-length alphabet); packed: 0..3 code bits, 4..7 extra bits, 8..22 base
    uint8_t len[320];              // code lengths: 0..287 literal/length, 288..319 distance
    uint16_t sorted[320];          // symbols ordered by code (canonical decoding of the long codes)
    uint16_t count[2][INF_MAXBITS + 1];
    uint16_t offs[2][INF_MAXBITS + 1];
    RingT ring[RING];              // output byte (or symbol: vgmi_gunzip.hip) p at ring[p % INF_RING]
};
typedef InfTablesT<uint8_t> InfTables;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The block headers and the one-symbol path read the input with SCALAR loads (constant address space, wave-uniform address)
typedef __attribute__((address_space(4))) const uint32_t inf_cu32;
__device__ __forceinline__ uint32_t ld32u(const uint32_t* p) { return *reinterpret_cast<inf_cu32*>((uintptr_t)p); }

__device__ __forceinline__ uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32 - len); }

__device__ __forceinline__ void inf_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Canonical Huffman tables of one alphabet from its code lengths (RFC 1951 3.2.2).  which: 0 literal/length, 1 distance.
// Returns false for an over-subscribed set of lengths (incomplete sets are legal only in the one-code cases zlib accepts;
// a code that is never assigned simply never matches and ends in the error path).
template <class T>
__device__ bool inf_build(T& t, uint32_t which, uint32_t first, uint32_t n, uint32_t lane)
{
    uint32_t* const tab = which ? t.dist : t.lit;
    const uint32_t bits = which ? INF_DIST_BITS : INF_LIT_BITS;
    for (uint32_t i = lane; i < (1u << bits); i += 64) tab[i] = 0;
    if (lane <= INF_MAXBITS) t.count[which][lane] = 0;
    inf_sync();
    // the counting and the canonical order are a few hundred steps: one lane
    uint32_t ok = 1;
    if (lane == 0) {
        for (uint32_t s = 0; s < n; ++s) t.count[which][t.len[first + s]]++;
        t.count[which][0] = 0;
        int32_t left = 1;
        uint32_t o = 0;
        for (uint32_t l = 1; l <= INF_MAXBITS; ++l) {
            left = (left << 1) - (int32_t)t.count[which][l];
            if (left < 0) ok = 0;
            t.offs[which][l] = (uint16_t)o;
            o += t.count[which][l];
        }
        if (ok) {
            uint16_t next[INF_MAXBITS + 1];
            for (uint32_t l = 1; l <= INF_MAXBITS; ++l) next[l] = t.offs[which][l];
            for (uint32_t s = 0; s < n; ++s) {
                const uint32_t l = t.len[first + s];
                if (l) t.sorted[which * 288 + next[l]++] = (uint16_t)s;
            }
        }
    }
    ok = uni(ok);
    inf_sync();
    if (!ok) return false;
    // fast table: every code of at most `bits` bits, replicated over the unused high index bits -- one symbol per lane
    // (its canonical code = first code of its length + its rank among the symbols of that length)
    uint32_t first_code[INF_MAXBITS + 2];
    {
        uint32_t code = 0;
        first_code[0] = 0;
        for (uint32_t l = 1; l <= INF_MAXBITS; ++l) {
            code = (code + t.count[which][l - 1]) << 1;
            first_code[l] = code;
        }
    }
    const uint32_t total = t.offs[which][INF_MAXBITS] + t.count[which][INF_MAXBITS];
    for (uint32_t r = lane; r < total; r += 64) {       // r = rank in canonical order
        const uint32_t s = t.sorted[which * 288 + r];
        const uint32_t l = t.len[first + s];
        if (l > bits) continue;
        const uint32_t code = first_code[l] + (r - t.offs[which][l]);
        const uint32_t e = s << 4 | l;
        for (uint32_t i = bitrev(code, l); i < (1u << bits); i += 1u << l) tab[i] = e;
    }
    inf_sync();
    return true;
}

// symbol << 4 | length  ->  what a lane of a batch needs in one word.  Sequence lines are runs of literals with 2-3 bit codes:
// an entry holds as many literals as the index has whole codes of (at most three).  Every lane reads its sixteen entries (and
// the entries their remaining index bits select) before any is rewritten.
template <class T>
__device__ void inf_pack_lit(T& t, uint32_t lane)
{
    uint32_t out[(1u << INF_LIT_BITS) / 64];
#pragma unroll
    for (uint32_t j = 0; j < (1u << INF_LIT_BITS) / 64; ++j) {
        const uint32_t i = lane + 64 * j;
        const uint32_t e = t.lit[i], l = e & 15u, sym = e >> 4;
        uint32_t r = 0;
        if (l && sym < 256) {
            uint32_t pos = l, n = 1, bytes = sym;
            while (n < 3) {
                const uint32_t e2 = t.lit[i >> pos];          // the bits above the index are unknown: only codes that fit count
                const uint32_t l2 = e2 & 15u, s2 = e2 >> 4;
                if (!l2 || l2 > INF_LIT_BITS - pos || s2 >= 256) break;
                bytes |= s2 << (8 * n);
                ++n;
                pos += l2;
            }
            r = pos | n << 6 | bytes << 8;
        } else if (l && sym == 256) {
            r = l | 2u << 4;
        } else if (l && sym - 257 < 29) {
            // length codes 257..285 (RFC 1951 3.2.5) in closed form
            const uint32_t c = sym - 257;
            const uint32_t eb = c < 8 || c == 28 ? 0u : (c >> 2) - 1u;
            const uint32_t base = c < 8 ? 3u + c : c == 28 ? 258u : ((4u + (c & 3u)) << eb) + 3u;
            r = l | 1u << 4 | eb << 6 | base << 9;
        }
        out[j] = r;       // (0: a code longer than the index, an unused code, a reserved symbol -- the one-symbol path decides)
    }
    inf_sync();
#pragma unroll
    for (uint32_t j = 0; j < (1u << INF_LIT_BITS) / 64; ++j) t.lit[lane + 64 * j] = out[j];
    inf_sync();
}

template <class T>
__device__ void inf_pack_dist(T& t, uint32_t lane)
{
    for (uint32_t i = lane; i < (1u << INF_DIST_BITS); i += 64) {
        const uint32_t e = t.dist[i], l = e & 15u, sym = e >> 4;
        uint32_t r = 0;
        if (l && sym < 30) {
            const uint32_t eb = sym < 4 ? 0u : (sym >> 1) - 1u;
            const uint32_t base = sym < 4 ? 1u + sym : ((2u + (sym & 1u)) << eb) + 1u;
            r = l | eb << 4 | base << 8;
        }
        t.dist[i] = r;
    }
    inf_sync();
}

// a code longer than the fast table: canonical decoding bit by bit (RFC 1951 3.2.2; rare by construction)
template <class T>
__device__ __forceinline__ int32_t inf_slow(const T& t, uint32_t which, uint64_t bitbuf, uint32_t& len_out)
{
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l <= INF_MAXBITS; ++l) {
        code |= (uint32_t)(bitbuf >> (l - 1)) & 1u;
        const uint32_t cnt = uni(t.count[which][l]);
        if (code < first + cnt) {
            len_out = l;
            return (int32_t)uni(t.sorted[which * 288 + index + (code - first)]);
        }
        index += cnt;
        first = (first + cnt) << 1;
        code <<= 1;
    }
    return -1;
}

static __device__ __constant__ uint8_t inf_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

}  // namespace vgk
#endif
