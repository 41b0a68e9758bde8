// vgmi_gunzip.hip -- ORDINARY gzip streams (one DEFLATE stream per file, what `gzip` and most sequencers write) inflated on the device.
//
// What it replaces: zlib's inflate behind gzread (include/kseq.h:59-72 over gzFile, src/fastq_kmer.cpp:74-78) -- one thread per
// file in the reference, several host threads per file in rounds 2-3 (csrc/host/par_gunzip.cpp: 4e7 reads/s with sixteen of them,
// a resource eight GPUs of a node share).  A DEFLATE stream has no member boundaries to split it at, so the scheme of
// par_gunzip.cpp moves to the device:
//   1. gz_find_kernel      a wavefront per 32 KiB of compressed bytes looks for the first position at or behind its offset where a
//                          dynamic-code block can start: 64 bit positions per step against the cheap tests (BFINAL = 0, BTYPE = 2,
//                          HLIT / HDIST in range, the code-length code's Kraft sum exact), survivors against the whole header
//                          (literal/length and distance codes complete).
//   2. gz_decode_kernel    a wavefront per stretch between two such starts decodes it the way vgmi_inflate.hip decodes a block-gzip
//                          member (batches of 64 bit positions), but into 16-bit SYMBOLS: a back-reference that reaches in front
//                          of the stretch -- into the 32 KiB window it cannot know -- yields placeholders 256 + (offset into that
//                          window), which later copies propagate like bytes.  A stretch must END exactly where the next one starts:
//                          that is the check of the guessed start (by induction from the stream's true first bit every start in an
//                          unbroken chain is a true block start); a stretch that does not is where the device path ends.
//   3. gz_window1/2_kernel the last 32 KiB of text behind every stretch: a chain, walked in two levels (groups side by side, then their tails)
//   4. gz_resolve_kernel   every stretch's symbols -> bytes with its predecessor's window, written where the text chunk wants them
// The host walks the gzip header, ships bytes, and reads back one small record per stretch.  Whatever the device cannot vouch for
// -- a broken chain, a stretch that outgrows its room, the end of a member (trailer, a next member's header) -- ends the device path
// at a known compressed offset and the host decoder (csrc/host/fast_inflate.cpp) goes on from there with the window it is handed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "vgmi_inflate_dev.h"
#include "vgmi_kernels.h"

namespace vgk {

#define GZ_WAVES 5u             // decode, batches of 64 bit positions: wavefronts per workgroup (10.3 KB of LDS each: three workgroups = 15 wavefronts per CU)
#define GZW_WAVES 4u            // decode, wide batches: 16.4 KB of LDS each, two workgroups = 8 wavefronts per CU
#define GZ_FIND_WAVES 4u        // find: 0.6 KB of LDS each
#define GZ_WIN 32768u
#define GZ_NONE 0xFFFFFFFFu

struct GzFindTables {            // per wavefront: the code-length code's symbols in canonical order
    uint16_t sorted[32];
    uint16_t queue[1024];         // positions of a step that passed the bit-parallel test (a third of 2048 at most)
};
// where symbol s of the code-length alphabet stands in the header's order (RFC 1951 3.2.7: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15)
static __device__ __constant__ uint8_t gz_clen_inv[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};

// ---- 1. block starts ---------------------------------------------------------------------------------------------------
// starts[i] = the first bit of sub-range i -- [8 * i * sub_bytes, 8 * (i + 1) * sub_bytes) -- where a dynamic block's header stands, GZ_NONE
// if there is none (or nobody looked).  Every sub-range has its own wavefront (round 4: a wavefront per 48 KiB stretch that searched
// up to two stretches far made the kernel as slow as its unluckiest wavefront); the host picks the stretches' starts from the list:
// the first one in the `per` sub-ranges behind j * per sub-ranges, else in the `per` behind those.  So the kernel is launched in
// phases: phase p looks at sub-ranges p * pw .. p * pw + pw - 1 of every group of `per`, and only where the phases before found
// nothing.  What a wavefront spends its time on is the ~220 code lengths of every candidate header that has a complete
// code-length code (one in ~3000 positions), walked in scalar registers at ~250 ns a length: small sub-ranges keep that chain short,
// and half of the positions never need looking at.
__global__ __launch_bounds__(64 * GZ_FIND_WAVES, 8) void gz_find_kernel(const uint8_t* __restrict__ comp, uint32_t n_bytes, uint32_t sub_bytes, uint32_t n_sub,
                                                                   uint32_t per, uint32_t pw, uint32_t phase, uint32_t* __restrict__ starts)
{
    __shared__ GzFindTables tabs[GZ_FIND_WAVES];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = uni(threadIdx.x >> 6);
    const uint32_t wv = blockIdx.x * GZ_FIND_WAVES + wave_in_block;
    const uint32_t grp = wv / pw, sub = pw * phase + wv % pw;
    const uint32_t j = grp * per + sub;
    if (sub >= per || j >= n_sub) return;
    if (__ballot(lane < pw * phase && starts[grp * per + lane] != GZ_NONE)) return;      // this group has its start (pw * phase <= 64)
    GzFindTables& t = tabs[wave_in_block];
    const uint32_t* const in4 = reinterpret_cast<const uint32_t*>(comp);      // the batch buffer is 256-byte aligned
    const uint32_t end_bits = n_bytes * 8u;
    const uint32_t from = j * sub_bytes * 8u;
    uint32_t to = from + sub_bytes * 8u;
    if (to > end_bits - (end_bits < 2048u ? end_bits : 2048u)) to = end_bits - (end_bits < 2048u ? end_bits : 2048u);    // a header needs room
    uint32_t found = GZ_NONE;

    // scalar bit reader (vgmi_inflate.hip's), restarted per candidate
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0, ip = 0;
    const uint32_t* wq;
    uint32_t wa, wb, wc;
    auto reload = [&]() {
        const uint32_t lead = ip & 3u;
        wq = in4 + (ip >> 2);
        wa = ld32u(wq);
        wb = ld32u(wq + 1);
        wc = ld32u(wq + 2);
        bitbuf = (uint64_t)(wa >> (8u * lead));
        bitcnt = 32u - 8u * lead;
        ip += 4u - lead;
        wa = wb;
        wb = wc;
        wc = ld32u(wq + 3);
        ++wq;
    };
    auto refill = [&]() {
        if (bitcnt <= 32u) {
            bitbuf |= (uint64_t)wa << bitcnt;
            bitcnt += 32u;
            ip += 4u;
            wa = wb;
            wb = wc;
            wc = ld32u(wq + 3);
            ++wq;
        }
    };
    auto take = [&](uint32_t n) -> uint32_t {
        const uint32_t v = (uint32_t)bitbuf & ((1u << n) - 1u);
        bitbuf >>= n;
        bitcnt -= n;
        return v;
    };
    // the whole header at bit p: three complete codes (what every encoder writes)?
    auto header_ok = [&](uint32_t p) -> bool {
        ip = p >> 3;
        reload();
        take(p & 7u);
        refill();
        take(3);
        const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        // The code-length code (19 symbols, lengths of 3 bits in the order of RFC 1951 3.2.7), built in registers: lane s holds symbol
        // s; counts per length by ballot, a symbol's canonical code = first code of its length + its rank among the symbols of that
        // length; every lane then fills two of the 128 entries of the 7-bit decoding table.  (The general table builder -- serial
        // loops of one lane over LDS -- cost 60 000 clock ticks a candidate, two thirds of this kernel: gpurun_out/r4w3.)
        refill();
        const uint32_t f_lo = take(30);
        refill();
        const uint64_t F = ((uint64_t)take(27) << 30 | f_lo) & ((1ull << (3u * hclen)) - 1ull);
        const uint32_t my_i = lane < 19 ? (uint32_t)gz_clen_inv[lane] : 63u;
        const uint32_t my_l = my_i < 19 ? (uint32_t)(F >> (3u * my_i)) & 7u : 0u;
        uint32_t first[8], cnt[8], off[8], my_off = 0;
        {
            uint32_t fc = 0, o = 0;
#pragma unroll
            for (uint32_t l = 1; l < 8; ++l) {
                const uint64_t bm = __ballot(my_l == l);
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                if (my_l == l) my_off = o + below;
                first[l] = fc;
                cnt[l] = (uint32_t)__builtin_popcountll(bm);
                off[l] = o;
                o += cnt[l];
                fc = (fc + cnt[l]) << 1;
            }
        }
        if (my_l) t.sorted[my_off] = (uint16_t)lane;
        inf_sync();
        uint32_t ent[2];          // the table stays in registers: entry i in lane i % 64 of ent[i / 64], looked up with v_readlane
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
            const uint32_t e = lane + 64u * h;
            uint32_t at = 0, len = 0;
#pragma unroll
            for (uint32_t l = 1; l < 8; ++l) {
                const uint32_t c = __builtin_bitreverse32(e) >> (32u - l);       // the first l bits of the index as a code
                if (c >= first[l] && c - first[l] < cnt[l]) { at = off[l] + c - first[l]; len = l; }
            }
            ent[h] = len ? (uint32_t)t.sorted[at] << 4 | len : 0u;
        }
        inf_sync();
        // the reader goes on behind the 3-bit lengths
        ip = (p + 17u + 3u * hclen) >> 3;
        reload();
        take((p + 17u + 3u * hclen) & 7u);
        // the hlit + hdist code lengths, with the Kraft sums of the two codes kept as they come (32768 = complete): a candidate that
        // is no header over-subscribes one of them within a few dozen lengths and ends there (decoding all ~300 lengths of every
        // false candidate was two thirds of this kernel)
        uint32_t idx = 0, prev = 0, sl = 0, sd = 0, nd = 0, one = 0, eob = 0;
        bool bad = false;
        auto put = [&](uint32_t i, uint32_t v) {
            if (!v) return;
            if (i < hlit) {
                sl += 32768u >> v;
                if (i == 256) eob = v;
            } else {
                sd += 32768u >> v;
                ++nd;
                one += v == 1;
            }
        };
        while (idx < hlit + hdist) {
            refill();
            const uint32_t i7 = (uint32_t)bitbuf & 127u;
            const uint32_t e = (i7 & 64u) ? (uint32_t)__builtin_amdgcn_readlane((int)ent[1], (int)(i7 & 63u)) : (uint32_t)__builtin_amdgcn_readlane((int)ent[0], (int)i7);
            const uint32_t l = e & 15u, sym = e >> 4;
            if (!l) { bad = true; break; }
            take(l);
            if (sym < 16) {
                put(idx++, sym);
                prev = sym;
            } else {
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (idx == 0) { bad = true; break; }
                    val = prev;
                    rep = 3 + take(2);
                } else if (sym == 17) rep = 3 + take(3);
                else rep = 11 + take(7);
                if (idx + rep > hlit + hdist) { bad = true; break; }
                if (val) {      // a run of equal lengths, taken whole: what falls in front of hlit is literal/length code, the rest distance code
                    const uint32_t n_lit = idx < hlit ? (rep < hlit - idx ? rep : hlit - idx) : 0u, n_dist = rep - n_lit, c = 32768u >> val;
                    sl += n_lit * c;
                    sd += n_dist * c;
                    nd += n_dist;
                    if (val == 1) one += n_dist;
                    if (idx <= 256u && 256u < idx + n_lit) eob = val;
                }
                idx += rep;
                prev = val;
            }
            if (sl > 32768u || sd > 32768u) { bad = true; break; }
        }
        if (bad) return false;
        // three complete codes (a single distance code of one bit is what zlib tolerates), the end-of-block code among them
        return sl == 32768u && eob != 0 && (sd == 32768u || (nd == 1 && one == 1));
    };

    // 2048 bit positions a step.  Lane i holds word i of the step and tests its 32 positions AT ONCE, bit-parallel: BFINAL 0 and
    // BTYPE 10 are the stream bits (0, 0, 1) at the position; HLIT <= 29 <=> not all of bits 4..7, HDIST <= 29 <=> not all of bits
    // 9..12 (the five-bit fields start at bits 3 and 8, least significant bit first) -- one in nine positions passes.  Those go, in
    // order, into a queue; the Kraft sum of the code-length code (19 three-bit lengths from bit 17 on: one in ~250 is complete)
    // is then taken by 64 queued positions at a time, and what passes that gets the whole header.  (A lane per position, Kraft sum
    // and all, was ~150 instructions per 64 positions; the search was issue-bound at 4.3 ms a piece.)
    for (uint32_t wbase = 0; from + 32u * wbase < to && found == GZ_NONE; wbase += 64u) {
        const uint32_t rel = 32u * (wbase + lane);                  // this lane's first position, from `from`
        const uint32_t* const w = in4 + ((from + rel) >> 5);
        const uint32_t w0 = w[0], w1 = w[1];
        auto sh = [&](uint32_t k) { return __builtin_amdgcn_alignbit(w1, w0, k); };
        uint32_t m = ~w0 & ~sh(1) & sh(2) & ~(sh(4) & sh(5) & sh(6) & sh(7)) & ~(sh(9) & sh(10) & sh(11) & sh(12));
        if (from + rel >= to) m = 0;
        const uint32_t cnt = (uint32_t)__builtin_popcount(m), incl = inf_scan(cnt);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t q = incl - cnt; m; ++q) {
            t.queue[q] = (uint16_t)(rel + (uint32_t)__builtin_ctz(m));
            m &= m - 1u;
        }
        inf_sync();
        for (uint32_t g = 0; g < total && found == GZ_NONE; g += 64u) {
            const bool have = g + lane < total;
            const uint32_t pos = have ? (uint32_t)t.queue[g + lane] : 0u, b = from + pos;
            const uint32_t* const v = in4 + (b >> 5);
            const uint32_t v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
            const uint32_t lo = __builtin_amdgcn_alignbit(v1, v0, b & 31u), mid = __builtin_amdgcn_alignbit(v2, v1, b & 31u),
                           hi = __builtin_amdgcn_alignbit(v3, v2, b & 31u);
            const uint32_t hclen = ((lo >> 13) & 15u) + 4u;
            // the 3-bit lengths start at bit 17: 57 bits at most
            const uint64_t f = ((uint64_t)__builtin_amdgcn_alignbit(hi, mid, 17) << 32 | __builtin_amdgcn_alignbit(mid, lo, 17)) & ((1ULL << (3u * hclen)) - 1ULL);
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t c = 0; c < 19; ++c) {
                const uint32_t l = (uint32_t)(f >> (3u * c)) & 7u;
                sum += l ? 128u >> l : 0u;
            }
            uint64_t mm = __ballot(have && b < to && sum == 128u);
            while (mm && found == GZ_NONE) {
                const uint32_t p = from + (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)__builtin_ctzll(mm));
                mm &= mm - 1ull;
                if (header_ok(p)) found = p;
            }
        }
        inf_sync();
    }
    if (lane == 0) starts[j] = found;
}

// ---- 2. a stretch between two block starts -> symbols ------------------------------------------------------------------
struct GzSeg {
    uint32_t start_bit, stop_bit;     // decode blocks from start_bit until one ends at or behind stop_bit (GZ_NONE: to the data's end)
    uint32_t sym_off, sym_cap;        // room in the symbol pool (symbols)
    uint32_t win_avail;               // bytes of text that exist in front of the stretch, GZ_WIN at most (0 at a member's start: a
                                      // back-reference beyond them is "invalid distance too far back", as zlib has it)
    uint32_t pad;
};
struct GzSegOut {
    uint32_t n_sym;        // symbols written
    uint32_t end_bit;      // where the last decoded block ended
    uint32_t status;       // 0 good: end_bit == stop_bit.  1 code lengths, 2 bad symbol / distance, 3 room / input overrun, 6 stored header,
                           // 7 reserved type, 8 ended behind stop_bit (a guessed start was no block start), 9 the data ended inside a block
    uint32_t final_block;  // the last decoded block carried BFINAL: the member ends at end_bit
};

template <bool WIDE>
__global__ __launch_bounds__(64 * (WIDE ? GZW_WAVES : GZ_WAVES), WIDE ? 2 : 4) void gz_decode_kernel(const uint8_t* __restrict__ comp, uint32_t n_bytes, const GzSeg* __restrict__ segs,
                                                                            uint32_t n_seg, uint16_t* __restrict__ pool, GzSegOut* __restrict__ outs)
{
    typedef InfWideT<uint16_t, 4096> WideTables;      // (2 048 entries and batches of 704 symbols leave LDS for twelve wavefronts a CU
                                                       // instead of eight, and cost 17.5 against 10.5 ms a piece: gpurun_out/r4w12)
    typedef typename std::conditional<WIDE, WideTables, InfTablesT<uint16_t>>::type GzTables;
    constexpr uint32_t RING = WIDE ? WideTables::kRing : INF_RING, NEAR = WIDE ? WideTables::kNear : INF_NEAR;
    constexpr uint32_t WAVES = WIDE ? GZW_WAVES : GZ_WAVES;
    __shared__ GzTables tabs[WAVES];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = uni(threadIdx.x >> 6);
    const uint32_t sg = blockIdx.x * WAVES + wave_in_block;
    if (sg >= n_seg) return;
    GzTables& t = tabs[wave_in_block];
    const uint8_t* const in = comp;
    const uint32_t* const in4 = reinterpret_cast<const uint32_t*>(comp);
    const uint32_t end_bits = n_bytes * 8u;
    const uint32_t stop_bit = uni(segs[sg].stop_bit);
    uint16_t* const out = pool + uni(segs[sg].sym_off);
    const uint32_t out_cap = uni(segs[sg].sym_cap);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(out_cap * 2u), 0x00020000);
    uint32_t* const ring32 = reinterpret_cast<uint32_t*>(t.ring);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(comp), 0, (int)((n_bytes + 3u) & ~3u), 0x00020000);

    uint32_t bp = uni(segs[sg].start_bit);
    const uint32_t win_avail = uni(segs[sg].win_avail);
    uint32_t op = 0, flushed = 0, err = 0;

    // ring -> global memory: whole blocks of 128 symbols (the pool slice is word aligned); all of it at the end
    auto flush = [&](bool all) {
        while (op - flushed >= 128u) {
            const uint32_t r = (flushed + 2u * lane) & (RING - 1u);
            __builtin_amdgcn_raw_buffer_store_b32(ring32[r >> 1], orsrc, (flushed + 2u * lane) * 2u, 0, 0);
            flushed += 128u;
        }
        if (all)
            while (flushed < op) {
                const uint32_t p = flushed + lane;
                if (p < op) __builtin_amdgcn_raw_buffer_store_b16(t.ring[p & (RING - 1u)], orsrc, p * 2u, 0, 0);
                flushed = flushed + 64u < op ? flushed + 64u : op;
            }
    };
    // one LZ77 match at output position P: sources in front of the stretch are placeholders for the window it does not know
    auto copy_match = [&](uint32_t P, uint32_t len, uint32_t dist) {
        const bool far = dist > NEAR;
        if (far) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // flushed symbols are read back (see vgmi_inflate.hip)
        for (uint32_t i = lane; i < len; i += 64) {
            const int32_t q = (int32_t)(P - dist + (dist >= len ? i : i % dist));
            uint16_t v;
            if (q < 0) v = (uint16_t)(256 + (int32_t)GZ_WIN + q);
            else v = far ? out[q] : t.ring[(uint32_t)q & (RING - 1u)];
            t.ring[(P + i) & (RING - 1u)] = v;
        }
        inf_sync();
    };

    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0, ip = 0;
    const uint32_t* wq;
    uint32_t wa, wb, wc;
    auto reload = [&]() {
        const uint32_t lead = ip & 3u;
        wq = in4 + (ip >> 2);
        wa = ld32u(wq);
        wb = ld32u(wq + 1);
        wc = ld32u(wq + 2);
        bitbuf = (uint64_t)(wa >> (8u * lead));
        bitcnt = 32u - 8u * lead;
        ip += 4u - lead;
        wa = wb;
        wb = wc;
        wc = ld32u(wq + 3);
        ++wq;
    };
    auto refill = [&]() {
        if (bitcnt <= 32u) {
            bitbuf |= (uint64_t)wa << bitcnt;
            bitcnt += 32u;
            ip += 4u;
            wa = wb;
            wb = wc;
            wc = ld32u(wq + 3);
            ++wq;
        }
    };
    auto need = [&](uint32_t n) { if (bitcnt < n) refill(); };
    auto take = [&](uint32_t n) -> uint32_t {
        const uint32_t v = (uint32_t)bitbuf & ((1u << n) - 1u);
        bitbuf >>= n;
        bitcnt -= n;
        return v;
    };
    auto scalar_at_bp = [&]() {
        ip = bp >> 3;
        reload();
        take(bp & 7u);
    };
    auto scalar_done = [&]() { bp = 8u * ip - bitcnt; };

    bool last = false;
    uint32_t block_end = bp;        // end of the last block decoded whole
    while (!last && !err && bp < stop_bit) {
        if (bp + 3u > end_bits) { err = 9; break; }
        scalar_at_bp();
        refill();
        last = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) {            // stored
            take(bitcnt & 7u);
            refill();
            const uint32_t len = take(16), nlen = take(16);
            if ((len ^ 0xFFFFu) != nlen) { err = 6; break; }
            const uint32_t src = ip - (bitcnt >> 3);
            if (src + len > n_bytes) { err = 9; break; }
            if (op + len > out_cap) { err = 3; break; }
            for (uint32_t done = 0; done < len;) {
                const uint32_t n = len - done < 128u ? len - done : 128u;
                for (uint32_t i = lane; i < n; i += 64) t.ring[(op + i) & (RING - 1u)] = in[src + done + i];
                inf_sync();
                op += n;
                done += n;
                flush(false);
            }
            bp = 8u * (src + len);
            block_end = bp;
            continue;
        }
        if (type == 3) { err = 7; break; }
        if (type == 1) {
            for (uint32_t s = lane; s < 288; s += 64) t.len[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) t.len[288 + lane] = 5;
            inf_sync();
            if (!inf_build(t, 0, 0, 288, lane) || !inf_build(t, 1, 288, 30, lane)) { err = 1; break; }
        } else {
            const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = 1; break; }
            refill();
            if (lane < 19) t.len[288 + lane] = 0;
            inf_sync();
            for (uint32_t i = 0; i < hclen; ++i) {
                if (bitcnt < 3) refill();
                const uint32_t v = take(3);
                if (lane == 0) t.len[288 + uni(inf_clen_order[i])] = (uint8_t)v;
            }
            inf_sync();
            if (!inf_build(t, 1, 288, 19, lane)) { err = 1; break; }
            uint32_t idx = 0, prev = 0;
            uint8_t* const stage = reinterpret_cast<uint8_t*>(t.lit);
            auto put = [&](uint32_t i, uint32_t v) { if (lane == 0) stage[i] = (uint8_t)v; };
            while (idx < hlit + hdist && !err) {
                refill();
                const uint32_t e = uni(t.dist[(uint32_t)bitbuf & ((1u << INF_DIST_BITS) - 1u)]);
                const uint32_t l = e & 15u, sym = e >> 4;
                if (!l) { err = 1; break; }
                take(l);
                if (sym < 16) {
                    put(idx++, sym);
                    prev = sym;
                } else {
                    uint32_t rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) { err = 1; break; }
                        val = prev;
                        rep = 3 + take(2);
                    } else if (sym == 17) rep = 3 + take(3);
                    else rep = 11 + take(7);
                    if (idx + rep > hlit + hdist) { err = 1; break; }
                    for (uint32_t r = 0; r < rep; ++r) put(idx++, val);
                    prev = val;
                }
            }
            if (err) break;
            inf_sync();
            uint8_t mine[5];
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) {
                const uint32_t s = lane + 64 * q;
                uint32_t v = 0;
                if (s < 288) {
                    if (s < hlit) v = stage[s];
                } else if (s - 288 < hdist) v = stage[hlit + (s - 288)];
                mine[q] = (uint8_t)v;
            }
            inf_sync();
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) t.len[lane + 64 * q] = mine[q];
            inf_sync();
            if (uni(t.len[256]) == 0) { err = 1; break; }
            if (!inf_build(t, 0, 0, 288, lane) || !inf_build(t, 1, 288, 30, lane)) { err = 1; break; }
        }
        if constexpr (WIDE) {
            infw_limits(t, 0, lane);
            infw_limits(t, 1, lane);
            infw_pack_lit2(t, lane);
        }
        inf_pack_lit(t, lane);
        inf_pack_dist(t, lane);
        scalar_done();

        bool eob = false;
        if constexpr (WIDE) {
            uint32_t nl = 40;      // sub-blocks a batch looks at: what the batches before it got through, and a few
            while (!eob && !err) {
                const uint32_t g = bp + 64u * lane;
                const uint32_t wo = (g >> 5) * 4u, sh = g & 31u;
                const uint32_t x0 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo, 0, 0), x1 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 4u, 0, 0),
                               x2 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 8u, 0, 0), x3 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 12u, 0, 0),
                               x4 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 16u, 0, 0);
                const uint32_t room = out_cap - op < WideTables::kCap ? out_cap - op : WideTables::kCap;
                const InfWideOut B = inf_wide<GzTables, uint16_t>(t, __builtin_amdgcn_alignbit(x1, x0, sh), __builtin_amdgcn_alignbit(x2, x1, sh),
                                                                  __builtin_amdgcn_alignbit(x3, x2, sh), __builtin_amdgcn_alignbit(x4, x3, sh), op, room, nl, lane);
                if (B.bad) { err = 2; break; }
                if (!B.adv) { err = 3; break; }                        // the stretch outgrows its room
                if (bp + B.adv > end_bits) { err = 9; break; }         // the symbols ran into the padding behind the data
                if (!infw_matches<GzTables, uint16_t, true>(t, B.n_match, op, out, win_avail, lane)) { err = 2; break; }
                op += B.out;
                bp += B.adv;
                eob = B.eob != 0;
                if (!eob) nl = B.last + 2u >= nl ? (nl + 8u < 64u ? nl + 8u : 64u) : B.last + 4u;
                flush(false);
            }
        } else
        while (!eob && !err) {
            const uint32_t b = bp + lane;
            const uint32_t* const w = in4 + (b >> 5);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, b & 31u), hi = __builtin_amdgcn_alignbit(w2, w1, b & 31u);
            const InfBatch B = inf_batch(t, lo, hi, lane);
            eob = B.eob;
            const bool slow = B.slow;
            const uint32_t off = B.out, pos = B.adv;
            uint64_t matches = B.matches;
            if (op + off > out_cap) { err = 3; break; }
            if (bp + pos > end_bits) { err = 9; break; }       // the symbols ran into the padding behind the data
            if ((B.lits >> lane) & 1ull) {
                const uint32_t e = B.e, n = (e >> 6) & 3u, P = op + B.off;
                if (((e >> 4) & 3u) == 0) {
                    t.ring[P & (RING - 1u)] = (uint16_t)((e >> 8) & 255u);
                    if (n > 1) t.ring[(P + 1u) & (RING - 1u)] = (uint16_t)((e >> 16) & 255u);
                    if (n > 2) t.ring[(P + 2u) & (RING - 1u)] = (uint16_t)(e >> 24);
                }
            }
            inf_sync();
            while (matches) {
                const uint32_t ml = (uint32_t)__builtin_ctzll(matches);
                matches &= matches - 1ull;
                const uint32_t P = op + (uint32_t)__builtin_amdgcn_readlane((int)B.off, (int)ml);
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)B.mlen, (int)ml);
                const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)B.mdist, (int)ml);
                if (dist > P + win_avail) { err = 2; break; }
                copy_match(P, len, dist);
            }
            if (err) break;
            op += off;
            bp += pos;
            flush(false);
            if (slow) {
                scalar_at_bp();
                need(32);
                uint32_t l;
                int32_t sym = inf_slow(t, 0, bitbuf, l);
                sym = (int32_t)uni((uint32_t)sym);
                l = uni(l);
                if (sym < 0) { err = 2; break; }
                take(l);
                if (sym < 256) {
                    if (op >= out_cap) { err = 3; break; }
                    if (lane == 0) t.ring[op & (RING - 1u)] = (uint16_t)sym;
                    inf_sync();
                    ++op;
                } else if (sym == 256) {
                    eob = true;
                } else {
                    sym -= 257;
                    if (sym >= 29) { err = 2; break; }
                    uint32_t len;
                    if (sym < 8) len = 3 + (uint32_t)sym;
                    else if (sym == 28) len = 258;
                    else {
                        const uint32_t x = ((uint32_t)sym >> 2) - 1;
                        len = ((4u + ((uint32_t)sym & 3u)) << x) + 3u + take(x);
                    }
                    need(32);
                    uint32_t dl2;
                    int32_t dsym = inf_slow(t, 1, bitbuf, dl2);
                    dsym = (int32_t)uni((uint32_t)dsym);
                    dl2 = uni(dl2);
                    if (dsym < 0 || dsym >= 30) { err = 2; break; }
                    take(dl2);
                    uint32_t dist;
                    if (dsym < 4) dist = 1 + (uint32_t)dsym;
                    else {
                        const uint32_t x = ((uint32_t)dsym >> 1) - 1;
                        dist = ((2u + ((uint32_t)dsym & 1u)) << x) + 1u + take(x);
                    }
                    if (dist > op + win_avail) { err = 2; break; }
                    if (op + len > out_cap) { err = 3; break; }
                    copy_match(op, len, dist);
                    op += len;
                }
                scalar_done();
                if (bp > end_bits) { err = 9; break; }
                flush(false);
            }
        }
        if (!err) block_end = bp;
    }
    if (!err && !last && stop_bit != GZ_NONE && block_end != stop_bit) err = 8;
    if (!err && !last && stop_bit == GZ_NONE) err = 9;        // the data ended between two blocks: the stream goes on in the next piece
    // (a member that ends in front of stop_bit is no error of this stretch: the chain ends with it, final_block says so)
    flush(true);
    if (lane == 0) outs[sg] = GzSegOut{op, block_end, err, last ? 1u : 0u};
}

// ---- 3. the window behind every stretch ------------------------------------------------------------------------------------
// The last GZ_WIN bytes of text behind stretch j need the window behind stretch j - 1: a chain over all stretches, 20 us a link
// when walked by one workgroup (75 of 114 ms for 3 350 stretches).  Two levels instead.  Stretches are taken in groups of
// GZ_GROUP; (a) every group walks its own chain, all groups side by side, leaving placeholders for the window in front of the
// GROUP in what it cannot know (16-bit windows w1); (b) one workgroup walks the chain of group tails, GZ_GROUP times shorter,
// into byte windows t (t[0] = the window in front of the piece); (c) the resolve kernel looks a placeholder up in the 16-bit
// window of the stretch before and, if that is a placeholder still, in the byte window of the group before.
#define GZ_GROUP 64u

__device__ __forceinline__ uint16_t gz_win_entry(const uint16_t* sym, uint32_t n, const uint16_t* prev, uint32_t x)
{
    // entry x of the window behind a stretch of n symbols whose predecessor's window is prev (nullptr: the group's first stretch,
    // whose predecessor is the unknown window itself: placeholder x stands for its entry x)
    if (n < GZ_WIN && x < GZ_WIN - n) return prev ? prev[x + n] : (uint16_t)(256u + x + n);
    const uint16_t s = sym[n >= GZ_WIN ? n - GZ_WIN + x : x - (GZ_WIN - n)];
    return s < 256 || !prev ? s : prev[s - 256];
}

__global__ __launch_bounds__(1024) void gz_window1_kernel(const uint16_t* __restrict__ pool, const GzSeg* __restrict__ segs, const GzSegOut* __restrict__ outs,
                                                          uint32_t n_seg, uint16_t* __restrict__ w1)
{
    const uint32_t j0 = blockIdx.x * GZ_GROUP, j1 = j0 + GZ_GROUP < n_seg ? j0 + GZ_GROUP : n_seg;
    for (uint32_t j = j0; j < j1; ++j) {
        const uint16_t* const prev = j == j0 ? nullptr : w1 + (size_t)(j - 1) * GZ_WIN;
        uint16_t* const cur = w1 + (size_t)j * GZ_WIN;
        const uint16_t* const sym = pool + segs[j].sym_off;
        const uint32_t n = outs[j].n_sym;
#pragma unroll 4
        for (uint32_t x = threadIdx.x; x < GZ_WIN; x += 1024u) cur[x] = gz_win_entry(sym, n, prev, x);
        __threadfence_block();
        __syncthreads();
    }
}

// t[(g + 1) * GZ_WIN ..] = the text window behind group g's last stretch (t[0 .. GZ_WIN): in front of the piece)
__global__ __launch_bounds__(1024) void gz_window2_kernel(const uint16_t* __restrict__ w1, uint32_t n_seg, uint8_t* __restrict__ t)
{
    const uint32_t n_groups = (n_seg + GZ_GROUP - 1) / GZ_GROUP;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t last = ((g + 1) * GZ_GROUP < n_seg ? (g + 1) * GZ_GROUP : n_seg) - 1u;
        const uint16_t* const w = w1 + (size_t)last * GZ_WIN;
        const uint8_t* const prev = t + (size_t)g * GZ_WIN;
        uint8_t* const cur = t + (size_t)(g + 1) * GZ_WIN;
#pragma unroll 4
        for (uint32_t x = threadIdx.x; x < GZ_WIN; x += 1024u) {
            const uint16_t s = w[x];
            cur[x] = s < 256 ? (uint8_t)s : prev[s - 256];
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- 4. symbols -> bytes --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gz_resolve_kernel(const uint16_t* __restrict__ pool, const GzSeg* __restrict__ segs, const GzSegOut* __restrict__ outs,
                                                         const uint64_t* __restrict__ text_off, uint32_t n_seg, const uint16_t* __restrict__ w1,
                                                         const uint8_t* __restrict__ t, uint8_t* __restrict__ text)
{
    const uint32_t j = blockIdx.x / 16u, part = blockIdx.x % 16u;      // sixteen workgroups per stretch
    if (j >= n_seg) return;
    const uint32_t g = j / GZ_GROUP;
    const uint8_t* const tg = t + (size_t)g * GZ_WIN;                  // the window in front of the stretch's group
    const uint16_t* const prev = j % GZ_GROUP ? w1 + (size_t)(j - 1) * GZ_WIN : nullptr;
    const uint16_t* const sym = pool + segs[j].sym_off;
    uint8_t* const dst = text + text_off[j];
    const uint32_t n = outs[j].n_sym;
    for (uint32_t x = part * 256u + threadIdx.x; x < n; x += 16u * 256u) {
        uint16_t s = sym[x];
        if (s >= 256 && prev) s = prev[s - 256];
        dst[x] = s < 256 ? (uint8_t)s : tg[s - 256];
    }
}

// ---- 5. CRC-32 of the resolved text -------------------------------------------------------------------------------------------
// What zlib checks at a member's trailer (gzread behind include/kseq.h:59-72): the chain of block starts proves the stretches'
// boundaries, not the windows and placeholders behind them -- the CRC does.  Computed in the LINEAR form R(M) = M(x) x^32 mod p
// (register starts at 0, no final inversion): R ignores zero bytes in front, so the text is cut into pieces counted from its END
// and every piece but a virtual zero-filled first one is whole -- one constant multiplier per level of every tree:
//     R(A || B) = R(A) x^(8 |B|) + R(B),        crc32(M) = R(M) ^ 0xFFFFFFFF x^(8 |M|) ^ 0xFFFFFFFF.
// gz_crc_kernel: a workgroup per GZ_CRC_CHUNK bytes, a lane per 64 of them out of LDS; gz_crc_fold_kernel: one workgroup folds the
// chunks' remainders and appends the piece to the member's running remainder and length (GzCrcState, device-resident across pieces).
#define GZ_POLY 0xEDB88320u
#define GZ_CRC_CHUNK 16384u
__host__ __device__ constexpr uint32_t gz_gf2_mul(uint32_t a, uint32_t b)      // a(x) b(x) mod p(x); bit 31 = x^0
{
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ GZ_POLY : b >> 1;
    }
    return p;
}
__host__ __device__ constexpr uint32_t gz_gf2_x8n(uint64_t n)                  // x^(8 n) mod p(x)
{
    uint32_t p = 1u << 31, sq = 0x00800000u;
    for (; n; n >>= 1) {
        if (n & 1u) p = gz_gf2_mul(sq, p);
        sq = gz_gf2_mul(sq, sq);
    }
    return p;
}
template <uint32_t B>
__device__ __forceinline__ uint32_t gz_mul_const(uint32_t a)                   // a(x) B(x): B's 32 shifts are immediates
{
    uint32_t p = 0, b = B;
#pragma unroll
    for (uint32_t j = 0; j < 32; ++j) {
        p ^= (a >> (31u - j) & 1u) ? b : 0u;
        b = (b & 1u) ? (b >> 1) ^ GZ_POLY : b >> 1;
    }
    return p;
}
template <uint32_t LVL>
__device__ __forceinline__ void gz_crc_tree(uint32_t* s, uint32_t lane)         // s[0] = R of 256 slices of 64 bytes
{
    if constexpr (LVL < 8) {
        constexpr uint32_t step = 1u << LVL;
        if ((lane & (2u * step - 1u)) == 0) s[lane] = gz_mul_const<gz_gf2_x8n(64ull * step)>(s[lane]) ^ s[lane + step];
        __syncthreads();
        gz_crc_tree<LVL + 1>(s, lane);
    }
}

__global__ __launch_bounds__(256) void gz_crc_kernel(const uint8_t* __restrict__ text, uint64_t n, uint32_t n_chunks, uint32_t* __restrict__ chunk_r)
{
    __shared__ uint32_t s_tab[256];
    __shared__ uint32_t s_data[256 * 17];      // a lane's 16 words + one of padding (bank = lane + word)
    __shared__ uint32_t s_r[256];
    const uint32_t lane = threadIdx.x;
    {
        uint32_t c = lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ GZ_POLY : c >> 1;
        s_tab[lane] = c;
    }
    // chunk b holds the bytes [n - (n_chunks - b) CHUNK, + CHUNK); what lies in front of byte 0 reads as zero
    const int64_t base = (int64_t)n - (int64_t)(n_chunks - blockIdx.x) * (int64_t)GZ_CRC_CHUNK;
    for (uint32_t w = lane; w < GZ_CRC_CHUNK / 4u; w += 256u) {
        const int64_t at = base + 4 * (int64_t)w;
        uint32_t v = 0;
        if (at >= 0 && !(reinterpret_cast<uintptr_t>(text + at) & 3u)) v = *reinterpret_cast<const uint32_t*>(text + at);
        else
            for (int k = 0; k < 4; ++k)
                if (at + k >= 0) v |= (uint32_t)text[at + k] << (8 * k);
        s_data[(w >> 4) * 17u + (w & 15u)] = v;
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll 4
    for (uint32_t w = 0; w < 16; ++w) {
        const uint32_t v = s_data[lane * 17u + w];
        c = s_tab[(c ^ v) & 0xFFu] ^ (c >> 8);
        c = s_tab[(c ^ (v >> 8)) & 0xFFu] ^ (c >> 8);
        c = s_tab[(c ^ (v >> 16)) & 0xFFu] ^ (c >> 8);
        c = s_tab[(c ^ (v >> 24)) & 0xFFu] ^ (c >> 8);
    }
    s_r[lane] = c;
    __syncthreads();
    gz_crc_tree<0>(s_r, lane);
    if (lane == 0) chunk_r[blockIdx.x] = s_r[0];
}

__global__ __launch_bounds__(1024) void gz_crc_fold_kernel(const uint32_t* __restrict__ chunk_r, uint32_t n_chunks, uint64_t n, GzCrcState* __restrict__ state)
{
    __shared__ uint32_t s_r[1024];
    const uint32_t lane = threadIdx.x;
    const uint32_t per = (n_chunks + 1023u) / 1024u;
    // lane l folds the chunks [n_chunks - (1024 - l) per, + per): counted from the end again, chunks in front of the first are zero
    uint32_t r = 0;
    for (uint32_t i = 0; i < per; ++i) {
        const int64_t b = (int64_t)n_chunks - (int64_t)(1024u - lane) * per + i;
        r = gz_mul_const<gz_gf2_x8n(GZ_CRC_CHUNK)>(r) ^ (b >= 0 ? chunk_r[b] : 0u);
    }
    s_r[lane] = r;
    __syncthreads();
    uint32_t x = gz_gf2_x8n((uint64_t)GZ_CRC_CHUNK * per);
    for (uint32_t step = 1; step < 1024u; step <<= 1) {
        if ((lane & (2u * step - 1u)) == 0) s_r[lane] = gz_gf2_mul(s_r[lane], x) ^ s_r[lane + step];
        x = gz_gf2_mul(x, x);
        __syncthreads();
    }
    if (lane == 0) {
        state->r = gz_gf2_mul(state->r, gz_gf2_x8n(n)) ^ s_r[0];
        state->len += n;
    }
}

hipError_t launch_gz_crc(const uint8_t* text, uint64_t n, uint32_t* chunk_r, GzCrcState* state, hipStream_t st)
{
    if (!n) return hipSuccess;
    const uint32_t n_chunks = (uint32_t)((n + GZ_CRC_CHUNK - 1) / GZ_CRC_CHUNK);
    hipLaunchKernelGGL(gz_crc_kernel, dim3(n_chunks), dim3(256), 0, st, text, n, n_chunks, chunk_r);
    hipLaunchKernelGGL(gz_crc_fold_kernel, dim3(1), dim3(1024), 0, st, chunk_r, n_chunks, n, state);
    return hipGetLastError();
}
size_t gz_crc_chunks(size_t n_text) { return (n_text + GZ_CRC_CHUNK - 1) / GZ_CRC_CHUNK; }
// the member's CRC-32 from its running remainder and length
uint32_t gz_crc_finish(uint32_t r, uint64_t len) { return r ^ gz_gf2_mul(0xFFFFFFFFu, gz_gf2_x8n(len)) ^ 0xFFFFFFFFu; }

hipError_t launch_gz_find(const uint8_t* comp, uint32_t n_bytes, uint32_t sub_bytes, uint32_t n_sub, uint32_t per, uint32_t* starts, hipStream_t st)
{
    if (!n_sub || !per) return hipSuccess;
    const uint32_t pw = per > 96 ? 32u : (per + 2u) / 3u;      // three phases (more for very long stretches: a phase checks <= 64 earlier sub-ranges)
    const uint32_t n_waves = pw * ((n_sub + per - 1) / per);
    for (uint32_t phase = 0; pw * phase < per && pw * phase <= 64u; ++phase)
        hipLaunchKernelGGL(gz_find_kernel, dim3((n_waves + GZ_FIND_WAVES - 1) / GZ_FIND_WAVES), dim3(64 * GZ_FIND_WAVES), 0, st, comp, n_bytes, sub_bytes, n_sub, per,
                           pw, phase, starts);
    return hipGetLastError();
}

hipError_t launch_gz_decode(const uint8_t* comp, uint32_t n_bytes, const void* segs, uint32_t n_seg, uint16_t* pool, void* outs, hipStream_t st)
{
    if (n_seg) {
        static const bool wide = !(getenv("VGMI_INFLATE_WIDE") && getenv("VGMI_INFLATE_WIDE")[0] == '0');
        const uint32_t waves = wide ? GZW_WAVES : GZ_WAVES;
        const dim3 grid((n_seg + waves - 1) / waves), block(64 * waves);
        if (wide) hipLaunchKernelGGL(gz_decode_kernel<true>, grid, block, 0, st, comp, n_bytes, static_cast<const GzSeg*>(segs), n_seg, pool, static_cast<GzSegOut*>(outs));
        else hipLaunchKernelGGL(gz_decode_kernel<false>, grid, block, 0, st, comp, n_bytes, static_cast<const GzSeg*>(segs), n_seg, pool, static_cast<GzSegOut*>(outs));
    }
    return hipGetLastError();
}

// w1: n_seg windows of GZ_WIN 16-bit entries; t: (n_groups + 1) windows of GZ_WIN bytes, t[0 .. GZ_WIN) = the window in front of the piece.
// Behind the call t + n_groups * GZ_WIN is the window behind the last stretch.
hipError_t launch_gz_resolve(const uint16_t* pool, const void* segs, const void* outs, const uint64_t* text_off, uint32_t n_seg, uint16_t* w1, uint8_t* t,
                             uint8_t* text, hipStream_t st)
{
    if (n_seg) {
        const uint32_t n_groups = (n_seg + GZ_GROUP - 1) / GZ_GROUP;
        hipLaunchKernelGGL(gz_window1_kernel, dim3(n_groups), dim3(1024), 0, st, pool, static_cast<const GzSeg*>(segs), static_cast<const GzSegOut*>(outs), n_seg, w1);
        hipLaunchKernelGGL(gz_window2_kernel, dim3(1), dim3(1024), 0, st, w1, n_seg, t);
        hipLaunchKernelGGL(gz_resolve_kernel, dim3(n_seg * 16u), dim3(256), 0, st, pool, static_cast<const GzSeg*>(segs), static_cast<const GzSegOut*>(outs), text_off,
                           n_seg, w1, t, text);
    }
    return hipGetLastError();
}

uint32_t gz_groups(uint32_t n_seg) { return (n_seg + GZ_GROUP - 1) / GZ_GROUP; }

}  // namespace vgk
