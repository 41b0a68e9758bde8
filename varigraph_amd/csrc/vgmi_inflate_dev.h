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
// All lanes at once: lane i holds symbols i, i + 64, ...; per code length, a ballot counts the symbols and ranks each among
// those of its length in symbol order -- its canonical code is the first code of the length + the rank.  (One lane walking the
// symbols through LDS, as rounds 2-3 had it, took ~0.1 ms a table: a tenth of a block-gzip member's time, and most of what a
// candidate block header cost the start search of vgmi_gunzip.hip.)
template <class T>
__device__ bool inf_build(T& t, uint32_t which, uint32_t first, uint32_t n, uint32_t lane)
{
    uint32_t* const tab = which ? t.dist : t.lit;
    const uint32_t bits = which ? INF_DIST_BITS : INF_LIT_BITS;
    for (uint32_t i = lane; i < (1u << bits); i += 64) tab[i] = 0;
    uint32_t myl[5], code[5], pos[5];
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) {
        const uint32_t s = lane + 64u * k;
        myl[k] = s < n ? t.len[first + s] : 0u;
        code[k] = pos[k] = 0;
    }
    uint32_t fc = 0, o = 0;
    int32_t left = 1;
    bool ok = true;
#pragma unroll
    for (uint32_t l = 1; l <= INF_MAXBITS; ++l) {
        uint32_t running = 0;
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) {
            if (64u * k >= n) break;
            const uint64_t bm = __ballot(myl[k] == l);
            if (myl[k] == l) {
                const uint32_t r = running + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                code[k] = fc + r;
                pos[k] = o + r;
            }
            running += (uint32_t)__builtin_popcountll(bm);
        }
        left = (left << 1) - (int32_t)running;
        if (left < 0) ok = false;
        if (lane == 0) {
            t.count[which][l] = (uint16_t)running;
            t.offs[which][l] = (uint16_t)o;
        }
        o += running;
        fc = (fc + running) << 1;
    }
    if (lane == 0) t.count[which][0] = 0;
    inf_sync();
    if (!ok) return false;
    // symbols in canonical order; the fast table: every code of at most `bits` bits, replicated over the unused high index bits
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) {
        if (64u * k >= n) break;
        const uint32_t l = myl[k], sym = lane + 64u * k;
        if (!l) continue;
        t.sorted[which * 288 + pos[k]] = (uint16_t)sym;
        if (l > bits) continue;
        const uint32_t e = sym << 4 | l;
        for (uint32_t i = bitrev(code[k], l); i < (1u << bits); i += 1u << l) tab[i] = e;
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

// ---- a WIDE batch: 64 sub-blocks of 64 bits, one per lane, walked symbol by symbol (round 4, second form) ---------------------
// inf_batch spends ~350 instructions of a whole wavefront on 64 CANDIDATE positions of which ~10 are symbol starts -- 30 bytes of
// text.  Here every lane walks its OWN 64 bits of the window one symbol after the other (~45 instructions a step for up to 64
// symbols of the wavefront), starting where the walk of the lane in front of it ends -- which nobody knows at first.  Huffman
// streams re-synchronise: a walk started at a wrong bit falls in step with the true one after a few symbols, so every lane first
// walks from bit 0 of its sub-block (lane 0's start is true), remembers the starts it visited as a 64-bit mask and where it left
// the sub-block; then each lane is told where the lane in front of it really ends: a start its mask holds changes nothing behind
// it, any other is walked until it meets the mask (or leaves the sub-block, which the next lane then hears of).  Repeated until no
// lane changes -- lane k is final after k rounds at the latest, in practice after two or three -- every lane holds the true
// symbol starts of its 64 bits.  A second walk counts output bytes and matches per lane (prefix sums give every lane its place in
// the ring and in the match queue), a third writes the literals and queues the matches, which the caller copies in order.
// Codes longer than the tables' index are decoded in the lane (canonical decoding against the per-length limits: infw_limits).
// ring entries RING (4096 in both decoders), output of a batch at most kCap, matches up to kNear back read the ring: kCap + 513 <= kNear <= RING - kCap (a batch's literals, written first,
// never land on a near source; a far source is in global memory already)
#define INFW_MQ 192u            // matches of a batch at most
#define INFW_SHORT 16u          // matches up to this long are copied by one lane each, all at once; longer ones by the wavefront, one after the other
#define INFW_UNSET 0x1FFu
#define INFW_END_EOB 0x100u
#define INFW_END_BAD 0x101u
#define INFW_END_OFF 0x102u

template <class RingT, uint32_t RING>
struct InfWideT {
    static constexpr uint32_t kRing = RING, kCap = RING >= 4096u ? 1536u : 704u, kNear = RING - kCap;
    static_assert(kCap + 513u <= kNear && (RING & (RING - 1u)) == 0, "ring too small for its batches");
    uint32_t lit[1u << INF_LIT_BITS];
    uint32_t dist[1u << INF_DIST_BITS];
    uint8_t len[320];
    uint16_t sorted[320];
    uint16_t count[2][INF_MAXBITS + 1];
    uint16_t offs[2][INF_MAXBITS + 1];
    uint8_t lit2[1u << INF_LIT_BITS];     // beside lit[]: code bits of an entry's first literal | of its first two << 4 (where the symbols
                                          // inside a run of literals start: infw_pack_lit2)
    uint16_t lim[2][16];           // [l]: codes of l bits, left-aligned to 15 bits, are below this ([0] = 0)
    int32_t sbase[2][16];          // [l]: index into sorted[] of a code of l bits = sbase + code
    uint32_t mq[INFW_MQ];          // the batch's matches in order: (length - 3) << 15 | (distance - 1)
    uint16_t mqp[INFW_MQ];         // ... and where they go, from the batch's first byte
    RingT ring[RING];
};

// per-length limits of the canonical code `which` (after inf_build)
template <class T>
__device__ void infw_limits(T& t, uint32_t which, uint32_t lane)
{
    if (lane <= INF_MAXBITS) {
        uint32_t code = 0;
        for (uint32_t i = 1; i <= lane; ++i) code = (code + (i > 1 ? t.count[which][i - 1] : 0u)) << 1;
        t.lim[which][lane] = lane ? (uint16_t)((code + t.count[which][lane]) << (INF_MAXBITS - lane)) : (uint16_t)0;
        t.sbase[which][lane] = (int32_t)t.offs[which][lane] - (int32_t)code;
    }
    inf_sync();
}

// lit2[] from the unpacked table (symbol << 4 | length): call before inf_pack_lit, whose choice of literals it repeats
template <class T>
__device__ void infw_pack_lit2(T& t, uint32_t lane)
{
    for (uint32_t i = lane; i < (1u << INF_LIT_BITS); i += 64) {
        const uint32_t e = t.lit[i], l = e & 15u;
        uint32_t x = 0;
        if (l && (e >> 4) < 256) {
            x = l;
            const uint32_t e2 = t.lit[i >> l], l2 = e2 & 15u;
            if (l2 && l2 <= INF_LIT_BITS - l && (e2 >> 4) < 256) x |= (l + l2) << 4;
        }
        t.lit2[i] = (uint8_t)x;
    }
    inf_sync();
}

// canonical decoding of the code at the low end of v: symbol or -1, its length in l
template <class T>
__device__ __forceinline__ int32_t infw_canon(const T& t, uint32_t which, uint32_t v, uint32_t& l)
{
    const uint32_t c15 = __builtin_bitreverse32(v) >> 17;
    uint32_t n = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) n += c15 >= t.lim[which][i];
    l = n;
    if (n > INF_MAXBITS) return -1;      // (lim[15] = 32768 for a complete code: never reached then)
    return (int32_t)t.sorted[which * 288 + (uint32_t)(t.sbase[which][n] + (int32_t)(c15 >> (INF_MAXBITS - n)))];
}

struct InfEnt {
    uint32_t bits, ol, fl;      // input bits, output bytes, 0 literals / 2 end of block / 4 match / 8 no such code
    uint32_t e;                 // literals: count in bits 6..7, bytes in 8..31
    uint32_t mlen, mdist;
};

// the symbol (or run of up to three short literals) that starts at the low end of hi:lo
template <class T>
__device__ __forceinline__ InfEnt inf_entry(const T& t, uint32_t lo, uint32_t hi)
{
    InfEnt E;
    const uint32_t e = t.lit[lo & ((1u << INF_LIT_BITS) - 1u)];
    const uint32_t cb = e & 15u, kind = (e >> 4) & 3u, eb = (e >> 6) & 7u;
    const uint32_t pd = cb + eb;
    const uint32_t de = t.dist[(lo >> pd) & ((1u << INF_DIST_BITS) - 1u)];
    const uint32_t dl = de & 15u, deb = (de >> 4) & 15u, qd = pd + dl;
    E.e = e;
    E.mlen = ((e >> 9) & 511u) + ((lo >> cb) & ((1u << eb) - 1u));
    E.mdist = (de >> 8) + (__builtin_amdgcn_alignbit(hi, lo, qd) & ((1u << deb) - 1u));
    if (cb && kind == 0) { E.bits = cb; E.ol = (e >> 6) & 3u; E.fl = 0; }
    else if (cb && kind == 2) { E.bits = cb; E.ol = 0; E.fl = 2; }
    else if (cb && dl) { E.bits = qd + deb; E.ol = E.mlen; E.fl = 4; }
    else {
        // a code the tables do not hold whole
        const uint64_t bb = (uint64_t)hi << 32 | lo;
        uint32_t l;
        const int32_t sym = infw_canon(t, 0, lo, l);
        E.bits = 0; E.ol = 0; E.fl = 8;
        if (sym >= 0 && sym < 256) { E.bits = l; E.ol = 1; E.fl = 0; E.e = 1u << 6 | (uint32_t)sym << 8; }
        else if (sym == 256) { E.bits = l; E.fl = 2; }
        else if (sym > 256 && sym < 286) {
            const uint32_t c = (uint32_t)sym - 257u;
            const uint32_t x = c < 8 || c == 28 ? 0u : (c >> 2) - 1u;
            const uint32_t base = c < 8 ? 3u + c : c == 28 ? 258u : ((4u + (c & 3u)) << x) + 3u;
            const uint32_t len = base + ((uint32_t)(bb >> l) & ((1u << x) - 1u));
            const uint32_t p2 = l + x;                                 // <= 20
            const uint32_t v = (uint32_t)(bb >> p2);
            const uint32_t d2 = t.dist[v & ((1u << INF_DIST_BITS) - 1u)];
            uint32_t dbits = d2 & 15u, dx = (d2 >> 4) & 15u, dbase = d2 >> 8;
            bool ok = dbits != 0;
            if (!ok) {
                const int32_t ds = infw_canon(t, 1, v, dbits);
                if (ds >= 0 && ds < 30) {
                    ok = true;
                    dx = ds < 4 ? 0u : ((uint32_t)ds >> 1) - 1u;
                    dbase = ds < 4 ? 1u + (uint32_t)ds : ((2u + ((uint32_t)ds & 1u)) << dx) + 1u;
                }
            }
            if (ok) {
                E.mlen = len;
                E.mdist = dbase + ((uint32_t)(bb >> (p2 + dbits)) & ((1u << dx) - 1u));
                E.bits = p2 + dbits + dx;                              // <= 48
                E.ol = len;
                E.fl = 4;
            }
        }
    }
    return E;
}

// inclusive prefix sum over the wavefront
__device__ __forceinline__ uint32_t inf_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
    return v;
}

struct InfWideOut {
    uint32_t out, adv, n_match;     // output bytes, input bits, queued matches of the batch
    uint32_t eob, bad;              // it ends with the end-of-block code / in front of bits that are no code
    uint32_t last;                  // the last lane whose symbols were taken
};

// W0..W3: the 128 bits that start at bit (position + 64 * lane).  op: the ring position of the batch's first byte; room: output the
// batch may produce (<= T::kCap); nl: lanes that take part (what the last batches used: the rounds are not spent on more).  Literals are in the ring and the matches in t.mq / t.mqp when it returns (behind an inf_sync).
template <class T, class RingT>
__device__ __forceinline__ InfWideOut inf_wide(T& t, uint32_t W0, uint32_t W1, uint32_t W2, uint32_t W3, uint32_t op, uint32_t room, uint32_t nl, uint32_t lane)
{
    auto bits_at = [&](uint32_t p, uint32_t& lo, uint32_t& hi) {
        const bool up = p >= 32u;
        const uint32_t a = up ? W1 : W0, b = up ? W2 : W1, c = up ? W3 : W2;
        lo = __builtin_amdgcn_alignbit(b, a, p);
        hi = __builtin_amdgcn_alignbit(c, b, p);
    };
    // a sub-block ends with the last SYMBOL that starts inside it: a run of literals that reaches over bit 64 is taken up to there
    // only (where a lane leaves off must not depend on how its walk happened to group the literals)
    auto lit_take = [&](uint32_t p, uint32_t lo, const InfEnt& E, uint32_t& n, uint32_t& adv) {
        n = (E.e >> 6) & 3u;
        adv = E.bits;
        if (n > 1 && p + E.bits > 64u) {
            const uint32_t x = t.lit2[lo & ((1u << INF_LIT_BITS) - 1u)], l1 = x & 15u, l2 = x >> 4;
            if (p + l1 >= 64u) { n = 1; adv = l1; }
            else if (n > 2 && p + l2 >= 64u) { n = 2; adv = l2; }
        }
    };
    // ---- the true symbol starts of every sub-block
    uint64_t mask = 0;
    uint32_t start = 0xFFFFu, endv = INFW_UNSET;
    for (;;) {
        uint32_t inc = inf_bperm(endv, (lane - 1u) & 63u);
        if (lane == 0) inc = 0;
        else if (inc == INFW_UNSET) inc = 0;                    // first round: every lane tries its bit 0
        else if (inc >= INFW_END_EOB) inc = INFW_END_OFF;
        if (lane >= nl) inc = INFW_END_OFF;
        const bool changed = inc != start;
        if (!__ballot(changed)) break;
        if (changed) {
            start = inc;
            if (inc >= INFW_END_EOB) {
                mask = 0;
                endv = INFW_END_OFF;
            } else if ((mask >> inc) & 1ull) {
                mask &= ~0ull << inc;
            } else {
                uint64_t nm = 0;
                uint32_t p = inc, ne = 0;
                while (p < 64u && !((mask >> p) & 1ull)) {
                    // a step of the walk needs the symbol's BITS only: two gathers side by side (entry, its literals' inner
                    // boundaries), the distance code's behind them, selects instead of branches; the mask takes every SYMBOL
                    // start, also those inside a run of literals taken in one step (two walks over the same symbols that group
                    // them differently must still meet)
                    const bool up = p >= 32u;
                    const uint32_t lo = __builtin_amdgcn_alignbit(up ? W2 : W1, up ? W1 : W0, p);
                    const uint32_t ix = lo & ((1u << INF_LIT_BITS) - 1u);
                    const uint32_t e = t.lit[ix], x = t.lit2[ix];
                    const uint32_t cb = e & 15u, kind = (e >> 4) & 3u, pd = cb + ((e >> 6) & 7u);
                    const uint32_t de = t.dist[(lo >> pd) & ((1u << INF_DIST_BITS) - 1u)];
                    const uint32_t dl = de & 15u;
                    uint32_t adv, fl;             // fl: 0 goes on, 2 end of block, 8 no such code
                    uint64_t add = 1ull << p;
                    if (cb && (kind == 0 || kind == 2 || dl)) {
                        const uint32_t n = (e >> 6) & 3u, l1 = x & 15u, l2 = x >> 4;
                        const uint32_t s1 = p + l1, s2 = p + l2;
                        adv = kind == 1 ? pd + dl + ((de >> 4) & 15u) : cb;
                        fl = kind == 2 ? 2u : 0u;
                        if (kind == 0 && n > 1) {
                            if (s1 >= 64u) adv = l1;
                            else {
                                add |= 1ull << s1;
                                if (n > 2) {
                                    if (s2 >= 64u) adv = l2;
                                    else add |= 1ull << s2;
                                }
                            }
                        }
                    } else {
                        const uint32_t hi = __builtin_amdgcn_alignbit(up ? W3 : W2, up ? W2 : W1, p);
                        const InfEnt E = inf_entry(t, lo, hi);
                        adv = E.bits;
                        fl = E.fl & 10u;
                    }
                    nm |= add;
                    if (fl) {
                        ne = (fl & 2u) ? INFW_END_EOB : INFW_END_BAD;
                        p = 1000u;
                    } else p += adv;
                }
                if (p < 64u) mask = nm | (mask & (~0ull << p));      // met the old walk: what is behind stands
                else {
                    mask = nm;
                    endv = ne ? ne : p - 64u;
                }
            }
        }
    }
    const bool on = start < INFW_END_EOB;
    // ---- output bytes and matches per lane
    uint32_t n_out = 0, n_m = 0, c_bad = 0;
    if (on) {
        uint32_t p = start;
        while (p < 64u) {
            uint32_t lo, hi;
            bits_at(p, lo, hi);
            const InfEnt E = inf_entry(t, lo, hi);
            if (E.fl & 8u) { c_bad = 1; break; }
            if (E.fl & 2u) break;
            uint32_t n = E.ol, adv = E.bits;
            if (E.fl == 0) lit_take(p, lo, E, n, adv);
            n_out += n;
            n_m += E.fl >> 2;
            p += adv;
        }
    }
    const uint32_t incl = inf_scan(n_out | n_m << 20);
    const uint32_t incl_out = incl & 0xFFFFFu, incl_m = incl >> 20;
    const uint64_t m_on = __ballot(on);
    const uint64_t m_fit = __ballot(on && !c_bad && incl_out <= room && incl_m <= INFW_MQ);
    const uint32_t n_full = m_fit == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m_fit);
    const uint32_t last = n_full < 64u && ((m_on >> n_full) & 1ull) ? n_full : n_full - 1u;      // (lane 0 is always on)
    // ---- literals into the ring, matches into the queue
    const uint32_t ex_out = incl_out - n_out, ex_m = incl_m - n_m;
    uint32_t o = 0, m = 0, p_end = 0, f_eob = 0, f_bad = 0;
    if (on && lane <= last) {
        const uint32_t lim_o = room - ex_out, lim_m = INFW_MQ - ex_m;       // (lanes in front of `last` never meet them)
        uint32_t p = start;
        while (p < 64u) {
            uint32_t lo, hi;
            bits_at(p, lo, hi);
            const InfEnt E = inf_entry(t, lo, hi);
            if (E.fl & 8u) { f_bad = 1; break; }
            if (E.fl & 2u) { f_eob = 1; p += E.bits; break; }
            uint32_t n = E.ol, adv = E.bits;
            if (E.fl == 0) lit_take(p, lo, E, n, adv);
            if (o + n > lim_o) break;
            const uint32_t P = op + ex_out + o;
            if (E.fl & 4u) {
                if (m >= lim_m) break;
                t.mq[ex_m + m] = (E.mlen - 3u) << 15 | (E.mdist - 1u);
                t.mqp[ex_m + m] = (uint16_t)(ex_out + o);
                ++m;
            } else {
                const uint32_t e = E.e;
                t.ring[P & (T::kRing - 1u)] = (RingT)((e >> 8) & 255u);
                if (n > 1) t.ring[(P + 1u) & (T::kRing - 1u)] = (RingT)((e >> 16) & 255u);
                if (n > 2) t.ring[(P + 2u) & (T::kRing - 1u)] = (RingT)(e >> 24);
            }
            o += n;
            p += adv;
        }
        p_end = p;
    }
    inf_sync();
    InfWideOut B;
    B.out = (uint32_t)__builtin_amdgcn_readlane((int)(ex_out + o), (int)last);
    B.adv = 64u * last + (uint32_t)__builtin_amdgcn_readlane((int)p_end, (int)last);
    B.n_match = (uint32_t)__builtin_amdgcn_readlane((int)(ex_m + m), (int)last);
    B.eob = (uint32_t)__builtin_amdgcn_readlane((int)f_eob, (int)last);
    B.bad = (uint32_t)__builtin_amdgcn_readlane((int)f_bad, (int)last);
    B.last = last;
    return B;
}

// The matches a batch queued, copied.  A match's bytes come from the ring (near), from what the wavefront has flushed to global
// memory (far: `out`, read back behind a wavefront-scope fence -- see vgmi_inflate.hip), or -- PH, ordinary gzip -- from in front of
// the stretch, as placeholders.  Most matches of FASTQ text are short and reach far back (a 7-mer seen 20 KiB ago): copied one after
// the other, each waits a memory round trip for a handful of bytes -- THAT is what bounded both decoders.  Here 64 queued matches are
// taken at once, a lane each: a match may go as soon as everything it reads is final, i.e. lies in front of the first match still
// waiting (literals are in place already; matches write only at or behind that point).  The first one waiting always may (a source
// that overlaps its own output is copied in order by its lane).  Matches longer than INFW_SHORT go the old way, 64 bytes a step.
// back_ok: how far in front of the output's first byte a distance may reach (0; the window a gzip stretch may assume).
// Returns false for a distance beyond that.
template <class T, class RingT, bool PH>
__device__ __forceinline__ bool infw_matches(T& t, uint32_t n_match, uint32_t op, const RingT* __restrict__ out, uint32_t back_ok, uint32_t lane)
{
    constexpr uint32_t M = T::kRing - 1u;
    auto ph = [](int32_t q) -> RingT { return (RingT)(256 + 32768 + q); };
    for (uint32_t c0 = 0; c0 < n_match; c0 += 64u) {
        const uint32_t j = c0 + lane;
        const bool have = j < n_match;
        const uint32_t d = have ? t.mq[j] : 0u;
        const uint32_t P = op + (have ? (uint32_t)t.mqp[j] : 0u);
        const uint32_t len = (d >> 15) + 3u, dist = (d & 32767u) + 1u;
        if (__ballot(have && dist > P + back_ok)) return false;
        const int32_t q0 = (int32_t)(P - dist);
        const int32_t src_end = dist < len ? (int32_t)P : q0 + (int32_t)len;
        const bool far = dist > T::kNear, is_long = len > INFW_SHORT;
        uint64_t pend = __ballot(have);
        if (__ballot(have && far)) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        while (pend) {
            const uint32_t first = (uint32_t)__builtin_ctzll(pend);
            const int32_t R0 = (int32_t)__builtin_amdgcn_readlane((int)P, (int)first);
            const bool ready = ((pend >> lane) & 1ull) && src_end <= R0;
            if (ready && !is_long) {
                if (far) {
                    for (uint32_t i = 0; i < len; i += 8u) {
                        RingT v[8];
#pragma unroll
                        for (uint32_t k = 0; k < 8u; ++k) {
                            const int32_t q = q0 + (int32_t)(i + k);
                            v[k] = i + k < len ? (PH && q < 0 ? ph(q) : out[q]) : (RingT)0;
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 8u; ++k)
                            if (i + k < len) t.ring[(P + i + k) & M] = v[k];
                    }
                } else {
                    for (uint32_t i = 0; i < len; ++i) {
                        const int32_t q = q0 + (int32_t)i;
                        t.ring[(P + i) & M] = PH && q < 0 ? ph(q) : t.ring[(uint32_t)q & M];
                    }
                }
            }
            uint64_t lm = __ballot(ready && is_long);
            while (lm) {
                const uint32_t ml = (uint32_t)__builtin_ctzll(lm);
                lm &= lm - 1ull;
                const uint32_t Pl = (uint32_t)__builtin_amdgcn_readlane((int)P, (int)ml), ll = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)ml),
                               dl = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)ml);
                const bool fl = dl > T::kNear;
                for (uint32_t i = lane; i < ll; i += 64u) {
                    const int32_t q = (int32_t)(Pl - dl + (dl >= ll ? i : i % dl));
                    RingT v;
                    if (PH && q < 0) v = ph(q);
                    else v = fl ? out[q] : t.ring[(uint32_t)q & M];
                    t.ring[(Pl + i) & M] = v;
                }
                inf_sync();
            }
            inf_sync();
            pend &= ~__ballot(ready);
        }
    }
    return true;
}

}  // namespace vgk
#endif
