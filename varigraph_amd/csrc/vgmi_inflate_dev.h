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

// ---- a batch: the symbols that start inside a window of 64 bit positions --------------------------------------------------
// lo / hi: the 64 bits that start at bit (position + lane).  Every lane decodes the symbol that WOULD start at its bit -- two table
// gathers -- and then the real symbol starts are picked out of the 64 candidates without a serial walk: with J[i] = where the symbol
// behind candidate i starts, the n-th start is p_n = J^n(0), and the lanes learn p_0 .. p_31 by doubling (round k: lanes
// 2^k .. 2^(k+1) - 1 take J^(2^k) of what lanes 0 .. 2^k - 1 hold, J^(2^(k+1)) = J^(2^k) o J^(2^k): three ds_bpermute a round).
// Behind the call LANE n holds symbol n of the batch (its table entry, output offset, match length and distance), compacted.  The
// scalar walk this replaces (one v_readlane and ~17 scalar instructions per symbol) was what the decoders were short of: a CU has
// ONE scalar unit for its sixteen wavefronts.
struct InfBatch {
    uint32_t e;            // lane n: the literal/length table entry of symbol n (kind, literal bytes)
    uint32_t off;          // lane n: where symbol n's output starts, from the batch's first byte
    uint32_t mlen, mdist;  // lane n: a match's length and distance
    uint64_t lits, matches;   // symbols of the batch that are literals (or the end-of-block code) / matches, by lane
    uint32_t out, adv;     // output bytes and input bits of the batch
    bool eob, slow;        // the batch ends with the end-of-block code / in front of a symbol the tables do not hold whole
};

__device__ __forceinline__ uint32_t inf_bperm(uint32_t v, uint32_t from_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane << 2), (int)v);
}

template <class T>
__device__ __forceinline__ InfBatch inf_batch(const T& t, uint32_t lo, uint32_t hi, uint32_t lane)
{
    // what would start at this lane's bit
    const uint32_t e = t.lit[lo & ((1u << INF_LIT_BITS) - 1u)];
    const uint32_t cb = e & 15u, kind = (e >> 4) & 3u;
    const uint32_t eb = (e >> 6) & 7u;
    const uint32_t mlen = ((e >> 9) & 511u) + ((lo >> cb) & ((1u << eb) - 1u));
    const uint32_t pd = cb + eb;                                    // <= 15
    const uint32_t de = t.dist[(lo >> pd) & ((1u << INF_DIST_BITS) - 1u)];
    const uint32_t dl = de & 15u, deb = (de >> 4) & 15u;
    const uint32_t qd = pd + dl;                                    // <= 23
    const uint32_t mdist = (de >> 8) + (__builtin_amdgcn_alignbit(hi, lo, qd) & ((1u << deb) - 1u));
    uint32_t bits, ol, fl;     // input bits, output bytes, flags: 1 needs the one-symbol path, 2 end of block, 4 a match
    if (cb == 0) { bits = 0; ol = 0; fl = 1; }
    else if (kind == 0) { bits = cb; ol = (e >> 6) & 3u; fl = 0; }
    else if (kind == 2) { bits = cb; ol = 0; fl = 2; }
    else if (dl) { bits = qd + deb; ol = mlen; fl = 4; }
    else { bits = 0; ol = 0; fl = 1; }
    const uint32_t pinfo = bits | ol << 6 | fl << 15, pm = mlen | mdist << 9;
    // J: 64 = nothing behind this candidate inside the window (or it ends the batch)
    uint32_t Jk = (fl & 3u) || lane + bits > 63u ? 64u : lane + bits;
    uint32_t P = lane == 0 ? 0u : 64u;
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) {
        const uint32_t h = 1u << k;
        const uint32_t q = inf_bperm(P, (lane - h) & 63u);
        const uint32_t c = inf_bperm(Jk, q & 63u);
        if (lane >= h && lane < 2u * h) P = q < 64u ? c : 64u;
        if ((uint32_t)__builtin_amdgcn_readlane((int)P, (int)(2u * h - 1u)) >= 64u) break;      // the chain ended inside these lanes
        const uint32_t jj = inf_bperm(Jk, Jk & 63u);
        Jk = Jk < 64u ? jj : 64u;
    }
    // lane n <- symbol n
    const bool valid = P < 64u;
    const uint32_t si = inf_bperm(pinfo, P & 63u), sm = inf_bperm(pm, P & 63u);
    InfBatch B;
    B.e = inf_bperm(e, P & 63u);
    const uint32_t sbits = si & 63u, sfl = si >> 15;
    const uint64_t mvalid = __ballot(valid), mslow = __ballot(valid && (sfl & 1u)), meob = __ballot(valid && (sfl & 2u));
    uint32_t n_sym = mvalid == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~mvalid);          // p_n grows with n: the valid lanes are a prefix
    if (mslow) n_sym = min(n_sym, (uint32_t)__builtin_ctzll(mslow));
    if (meob) n_sym = min(n_sym, (uint32_t)__builtin_ctzll(meob) + 1u);
    // output offsets: prefix sums of the symbols' output bytes (within rows of 16 lanes, then across the rows)
    const uint32_t own = lane < n_sym ? (si >> 6) & 511u : 0u;
    uint32_t incl = own;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);      // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);      // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);      // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);      // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
    // no more symbols than leave the batch's output inside INF_BATCH_OUT bytes (the first always)
    const uint64_t mfit = __ballot(lane < n_sym && (incl <= INF_BATCH_OUT || lane == 0));
    n_sym = mfit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~mfit);
    B.off = incl - own;
    B.mlen = sm & 511u;
    B.mdist = sm >> 9;
    const uint64_t mtaken = n_sym >= 64u ? ~0ull : (1ull << n_sym) - 1ull;
    const uint64_t mmatch = __ballot(valid && (sfl & 4u));
    B.matches = mmatch & mtaken;
    B.lits = mtaken & ~mmatch;
    B.out = n_sym ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(n_sym - 1u)) : 0u;
    B.adv = n_sym ? (uint32_t)__builtin_amdgcn_readlane((int)(P + sbits), (int)(n_sym - 1u)) : 0u;
    B.eob = n_sym && ((meob >> (n_sym - 1u)) & 1ull);
    B.slow = n_sym < 64u && ((mslow >> n_sym) & 1ull);
    return B;
}

}  // namespace vgk
#endif
