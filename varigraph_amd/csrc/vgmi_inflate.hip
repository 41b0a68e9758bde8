// vgmi_inflate.hip -- block-gzip (BGZF) members inflated on the device (gfx950), one wavefront per member.
//
// What it replaces: zlib's inflate behind gzread (include/kseq.h:59-72 over gzFile, src/fastq_kmer.cpp:74-78) for files
// written by bgzip / htslib: a BGZF file is a series of complete gzip members of at most 64 KiB (the 'BC' extra field
// carries each member's size), so the members are independent DEFLATE streams (RFC 1951) -- the host only walks the
// headers and ships compressed bytes; every member becomes text in its place of the chunk the FASTQ kernels
// (vgmi_fastq.hip) parse.  A member the kernel cannot vouch for -- bad Huffman code, output that does not match ISIZE
// or CRC-32 (RFC 1952) -- is reported, the chunk is cut in front of it and the host decoder (csrc/host/fast_inflate.cpp,
// pinned against zlib) takes the stream over there; nothing the device emits is unchecked.
//
// Shape of the kernel (round 4).  DEFLATE decoding is serial inside a member: where a symbol starts is known only once the
// symbol before it is decoded.  Rounds 2-3 ran that chain one look-up at a time in scalar registers (one LDS round trip and
// ~14 scalar instructions per one to three output bytes: latency-bound at 58 cycles per byte and SIMD).  Now a BATCH of 64 bit
// positions is decoded at once: lane i decodes the symbol that would start at bit (position + i) -- one gather from the
// literal/length table (entries hold up to three literals), one from the distance table -- and only then a short scalar walk
// hops from symbol start to symbol start over the lanes' results (one v_readlane per symbol), handing every real symbol its
// output offset.  Literals of the whole batch are written in one step; matches are copied one after the other, each copy
// 64 bytes wide (a match of length n from distance d is the periodic extension of the d bytes before it).  Output goes to a
// 2 KiB LDS ring (the LZ77 window for near matches) and leaves for global memory in aligned 256-byte blocks.  Codes longer
// than the tables' index, stored blocks and block headers take a scalar path, one symbol at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "vgmi_kernels.h"

#include "vgmi_inflate_dev.h"

namespace vgk {

// ---- CRC-32 (gzip polynomial, reflected) ---------------------------------------------------------------------------
#define INF_POLY 0xEDB88320u
// a(x) * b(x) mod p(x); bit 31 = x^0 (the arithmetic of zlib's crc32_combine)
__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ INF_POLY : b >> 1;
    }
    return p;
}
// x^(8 n) mod p(x)
__device__ __forceinline__ uint32_t gf2_x8n(uint32_t n)
{
    uint32_t p = 1u << 31, sq = 0x00800000u;   // x^8
    for (; n; n >>= 1) {
        if (n & 1u) p = gf2_mul(sq, p);
        sq = gf2_mul(sq, sq);
    }
    return p;
}

// One member per wavefront.  status[m]: 0 = good, else the reason (1 code lengths, 2 bad symbol / distance, 3 output or
// input overrun, 4 length != ISIZE, 5 CRC-32, 6 stored-block header, 7 reserved block type).
template <bool WIDE>
__global__ __launch_bounds__(64 * INF_WAVES, WIDE ? 3 : 4) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const BgzfMember* __restrict__ members,
                                                                                uint32_t n_members, uint8_t* __restrict__ out_base, uint32_t* __restrict__ status,
                                                                                const uint32_t* __restrict__ crc_table)
{
    typedef InfWideT<uint8_t, 4096> WideTables;
    typedef typename std::conditional<WIDE, WideTables, InfTables>::type Tables;
    constexpr uint32_t RING = WIDE ? WideTables::kRing : INF_RING, NEAR = WIDE ? WideTables::kNear : INF_NEAR;
    __shared__ Tables tabs[INF_WAVES];
    __shared__ uint32_t s_crc[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_crc[i] = crc_table[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = uni(threadIdx.x >> 6);
    const uint32_t m = blockIdx.x * INF_WAVES + wave_in_block;
    if (m >= n_members) return;
    Tables& t = tabs[wave_in_block];
    const uint8_t* in = comp + uni(members[m].c_off);
    const uint32_t in_len = uni(members[m].c_len);       // deflate bytes (header and trailer stripped by the host walk)
    uint8_t* out = out_base + uni(members[m].u_off);
    const uint32_t out_len = uni(members[m].u_len);      // ISIZE
    const uint32_t want_crc = uni(members[m].crc);
    // stores at offsets >= ISIZE are dropped by the hardware
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_len, 0x00020000);
    // the batches read the input as aligned 32-bit words: in4 + lead_bits is the member's first bit
    const uint32_t* const in4 = reinterpret_cast<const uint32_t*>((uintptr_t)in & ~(uintptr_t)3);
    const uint32_t lead_bits = 8u * (uint32_t)((uintptr_t)in & 3u);
    uint32_t* const ring32 = reinterpret_cast<uint32_t*>(t.ring);
    // the wide batches read the input through a descriptor: words behind the member's last read as zero
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(in4), 0, (int)((lead_bits / 8u + in_len + 3u) & ~3u), 0x00020000);

    // ---- wave-uniform decoder state ----
    uint32_t bp = 0;            // bits of the member's deflate data consumed
    uint32_t op = 0;            // output bytes produced (in the ring)
    uint32_t flushed = 0;       // ... of which the first `flushed` are in global memory
    uint32_t err = 0;
    const uint32_t head = (4u - (uint32_t)((uintptr_t)out & 3u)) & 3u;      // bytes in front of the first aligned word of the output

    // ring -> global memory: whole 256-byte blocks of aligned words (all of it at the end)
    auto flush = [&](bool all) {
        if (flushed < head && (op >= head || all)) {
            const uint32_t n = op < head ? op : head;
            if (lane < n) __builtin_amdgcn_raw_buffer_store_b8(t.ring[lane], orsrc, lane, 0, 0);
            flushed = n;
        }
        while (flushed >= head && op - flushed >= 256u) {
            const uint32_t r = (flushed + 4u * lane) & (RING - 1u);
            const uint32_t w0 = ring32[r >> 2], w1 = ring32[((r >> 2) + 1u) & (RING / 4u - 1u)];
            const uint32_t v = __builtin_amdgcn_alignbyte(w1, w0, r & 3u);
            __builtin_amdgcn_raw_buffer_store_b32(v, orsrc, flushed + 4u * lane, 0, 0);
            flushed += 256u;
        }
        if (all)
            while (flushed < op) {
                const uint32_t p = flushed + lane;
                if (p < op) __builtin_amdgcn_raw_buffer_store_b8(t.ring[p & (RING - 1u)], orsrc, p, 0, 0);
                flushed = flushed + 64u < op ? flushed + 64u : op;
            }
    };
    // one LZ77 match at output position P (every earlier byte is in the ring, or in global memory when far): 64 bytes per step
    auto copy_match = [&](uint32_t P, uint32_t len, uint32_t dist) {
        if (dist <= NEAR) {
            for (uint32_t i = lane; i < len; i += 64) {
                const uint8_t b = t.ring[(P - dist + (dist >= len ? i : i % dist)) & (RING - 1u)];
                t.ring[(P + i) & (RING - 1u)] = b;
            }
        } else {
            // far: the source is in front of everything still unflushed (NEAR > the ring's unflushed part + a batch); read it
            // back once the wavefront's stores have landed (its own stores and loads go through the same vector cache; an
            // agent-scope fence here writes the whole L2 back and made the kernel 15 x slower -- whatever this read could get
            // wrong, the CRC below catches and the host decoder redoes)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint8_t* const src = out + P - dist;
            for (uint32_t i = lane; i < len; i += 64) t.ring[(P + i) & (RING - 1u)] = src[dist >= len ? i : i % dist];
        }
        inf_sync();
    };

    // ---- the scalar bit reader of block headers, stored blocks and the one-symbol path: (re)started at bit bp ----
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0, ip = 0;
    const uint32_t* wq;
    uint32_t wa, wb, wc;
    auto reload = [&]() {          // start at input byte ip: the bit buffer is empty
        const uint32_t lead = (uint32_t)((uint64_t)(in + ip) & 3u);
        wq = reinterpret_cast<const uint32_t*>((uint64_t)(in + ip) & ~3ULL);
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
    auto refill = [&]() {          // leaves at least 33 bits
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
    while (!last && !err) {
        scalar_at_bp();
        refill();
        last = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) {            // stored
            take(bitcnt & 7u);      // to the byte boundary
            refill();
            const uint32_t len = take(16), nlen = take(16);
            if ((len ^ 0xFFFFu) != nlen) { err = 6; break; }
            const uint32_t src = ip - (bitcnt >> 3);      // the bytes still in the bit buffer come first
            if (src + len > in_len || op + len > out_len) { err = 3; break; }
            for (uint32_t done = 0; done < len;) {
                const uint32_t n = len - done < 256u ? len - done : 256u;
                for (uint32_t i = lane; i < n; i += 64) t.ring[(op + i) & (RING - 1u)] = in[src + done + i];
                inf_sync();
                op += n;
                done += n;
                flush(false);
            }
            bp = 8u * (src + len);
            continue;
        }
        if (type == 3) { err = 7; break; }
        if (type == 1) {            // fixed Huffman codes (RFC 1951 3.2.6)
            for (uint32_t s = lane; s < 288; s += 64) t.len[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) t.len[288 + lane] = 5;
            inf_sync();
            if (!inf_build(t, 0, 0, 288, lane) || !inf_build(t, 1, 288, 30, lane)) { err = 1; break; }
        } else {                    // dynamic codes: the code lengths are themselves Huffman coded
            const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = 1; break; }
            refill();
            // code length alphabet: built in the distance slots (19 symbols), decoded through the distance table
            if (lane < 19) t.len[288 + lane] = 0;
            inf_sync();
            for (uint32_t i = 0; i < hclen; ++i) {
                if (bitcnt < 3) refill();
                const uint32_t v = take(3);
                if (lane == 0) t.len[288 + uni(inf_clen_order[i])] = (uint8_t)v;
            }
            inf_sync();
            if (!inf_build(t, 1, 288, 19, lane)) { err = 1; break; }
            // the hlit + hdist lengths, written to a staging area first (the code-length code occupies len[288..306])
            uint32_t idx = 0, prev = 0;
            uint8_t* const stage = reinterpret_cast<uint8_t*>(t.lit);        // rebuilt below
            auto put = [&](uint32_t i, uint32_t v) { if (lane == 0) stage[i] = (uint8_t)v; };
            while (idx < hlit + hdist && !err) {
                refill();
                const uint32_t e = uni(t.dist[(uint32_t)bitbuf & ((1u << INF_DIST_BITS) - 1u)]);
                const uint32_t l = e & 15u, sym = e >> 4;
                if (!l) { err = 1; break; }     // code-length codes are at most 7 bits: always in the fast table
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
            // staging -> len[]: literal/length 0..hlit-1 (rest 0), distance 288..288+hdist-1 (rest 0)
            uint8_t mine[5];
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) {
                const uint32_t s = lane + 64 * q;      // 0..319
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
            if (uni(t.len[256]) == 0) { err = 1; break; }    // no end-of-block code
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
            // ---- symbols of this block: batches of 64 sub-blocks of 64 bits (vgmi_inflate_dev.h: inf_wide) ----
            while (!eob && !err) {
                const uint32_t g = lead_bits + bp + 64u * lane;
                const uint32_t wo = (g >> 5) * 4u, sh = g & 31u;
                const uint32_t x0 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo, 0, 0), x1 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 4u, 0, 0),
                               x2 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 8u, 0, 0), x3 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 12u, 0, 0),
                               x4 = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(irsrc, wo + 16u, 0, 0);
                const uint32_t room = out_len - op < WideTables::kCap ? out_len - op : WideTables::kCap;
                const InfWideOut B = inf_wide<Tables, uint8_t>(t, __builtin_amdgcn_alignbit(x1, x0, sh), __builtin_amdgcn_alignbit(x2, x1, sh),
                                                               __builtin_amdgcn_alignbit(x3, x2, sh), __builtin_amdgcn_alignbit(x4, x3, sh), op, room, nl, lane);
                if (B.bad) { err = 2; break; }
                if (!B.adv || (bp >> 3) > in_len + 8u) { err = 3; break; }      // nothing fits: more text than ISIZE says; or a damaged stream running away
                if (!infw_matches<Tables, uint8_t, false>(t, B.n_match, op, out, 0u, lane)) { err = 2; break; }
                op += B.out;
                bp += B.adv;
                eob = B.eob != 0;
                if (!eob) nl = B.last + 2u >= nl ? (nl + 8u < 64u ? nl + 8u : 64u) : B.last + 4u;
                flush(false);
            }
        } else
        // ---- symbols of this block: batches of 64 bit positions ----
        while (!eob && !err) {
            // the 64 bits that start at bit bp + lane
            const uint32_t b = lead_bits + bp + lane;
            const uint32_t* const w = in4 + (b >> 5);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, b & 31u), hi = __builtin_amdgcn_alignbit(w2, w1, b & 31u);
            // what would start here
            const InfBatch B = inf_batch(t, lo, hi, lane);
            eob = B.eob;
            const bool slow = B.slow;
            const uint32_t off = B.out, pos = B.adv;
            uint64_t matches = B.matches;
            if (op + off > out_len || (bp >> 3) > in_len + 8u) { err = 3; break; }     // (a damaged stream must not run away over the input)
            // literals: up to three bytes per lane
            if ((B.lits >> lane) & 1ull) {
                const uint32_t e = B.e, n = (e >> 6) & 3u, P = op + B.off;
                if (((e >> 4) & 3u) == 0) {
                    t.ring[P & (RING - 1u)] = (uint8_t)(e >> 8);
                    if (n > 1) t.ring[(P + 1u) & (RING - 1u)] = (uint8_t)(e >> 16);
                    if (n > 2) t.ring[(P + 2u) & (RING - 1u)] = (uint8_t)(e >> 24);
                }
            }
            inf_sync();
            // matches, in order
            while (matches) {
                const uint32_t ml = (uint32_t)__builtin_ctzll(matches);
                matches &= matches - 1ull;
                const uint32_t P = op + (uint32_t)__builtin_amdgcn_readlane((int)B.off, (int)ml);
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)B.mlen, (int)ml);
                const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)B.mdist, (int)ml);
                if (dist > P) { err = 2; break; }
                copy_match(P, len, dist);
            }
            if (err) break;
            op += off;
            bp += pos;
            flush(false);
            if (slow) {
                // one symbol the tables do not hold whole (a code longer than the index): the scalar way
                scalar_at_bp();
                need(32);
                uint32_t l;
                int32_t sym = inf_slow(t, 0, bitbuf, l);
                sym = (int32_t)uni((uint32_t)sym);
                l = uni(l);
                {   // (a short code the packed table rejected -- a reserved length symbol -- decodes here too)
                    if (sym < 0) { err = 2; break; }
                }
                take(l);
                if (sym < 256) {
                    if (op >= out_len) { err = 3; break; }
                    if (lane == 0) t.ring[op & (RING - 1u)] = (uint8_t)sym;
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
                    need(32);     // a distance code (<= 15 bits) and its extra bits (<= 13)
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
                    if (dist > op) { err = 2; break; }
                    if (op + len > out_len) { err = 3; break; }
                    copy_match(op, len, dist);
                    op += len;
                }
                scalar_done();
                flush(false);
            }
        }
        if ((bp + 7u) / 8u > in_len) err = 3;   // the block ran past the member's deflate bytes
    }
    if (!err && op != out_len) err = 4;
    flush(true);
    if (!err && out_len) {
        // CRC-32 of the output: one slice per lane, then crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / p(x).  The
        // bytes are read back from global memory once the wavefront's stores have landed.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t slice = (out_len + 63u) / 64u;
        const uint32_t b = lane * slice < out_len ? lane * slice : out_len;
        const uint32_t e = b + slice < out_len ? b + slice : out_len;
        uint32_t c = 0xFFFFFFFFu;
        {
            // sixteen bytes a step, as four aligned words fetched together (one byte load a step, each waited for, was 0.4 of a
            // member's 3 ms)
            uint32_t i = b;
            for (; i < e && ((uintptr_t)(out + i) & 3u); ++i) c = s_crc[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
            for (; i + 16u <= e; i += 16u) {
                const uint32_t* const w = reinterpret_cast<const uint32_t*>(out + i);
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
#pragma unroll
                for (uint32_t k = 0; k < 16u; ++k) {
                    const uint32_t word = k < 4 ? w0 : k < 8 ? w1 : k < 12 ? w2 : w3;
                    c = s_crc[(c ^ (word >> (8u * (k & 3u)))) & 0xFFu] ^ (c >> 8);
                }
            }
            for (; i < e; ++i) c = s_crc[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
        }
        t.lit[lane] = ~c;      // (the decode tables are no longer needed)
        inf_sync();
        if (lane == 0) {
            const uint32_t x_full = gf2_x8n(slice);
            uint32_t crc = t.lit[0];
            for (uint32_t i = 1; i < 64; ++i) {
                const uint32_t bi = i * slice;
                if (bi >= out_len) break;
                const uint32_t li = bi + slice <= out_len ? slice : out_len - bi;
                crc = gf2_mul(li == slice ? x_full : gf2_x8n(li), crc) ^ t.lit[i];
            }
            if (crc != want_crc) err = 5;
        }
        err = uni(err);
    }
    if (lane == 0) status[m] = err;
}

// first bad member of the batch (n_members if all are good), and the chunk length the FASTQ kernels may parse
__global__ void bgzf_verdict_kernel(const BgzfMember* members, const uint32_t* status, uint32_t n_members, BgzfVerdict* v)
{
    __shared__ uint32_t first;
    if (threadIdx.x == 0) first = n_members;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_members; i += blockDim.x)
        if (status[i]) atomicMin(&first, i);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (v->first_bad_batch == 0xFFFFFFFFu && first < n_members) {   // sticky: only the first failure of the stream counts
            v->first_bad_batch = v->batches;
            v->first_bad_member = first;
            v->reason = status[first];
        }
        // text that may be parsed: everything in front of the first bad member (nothing once an earlier batch failed)
        uint32_t good_bytes = 0;
        if (v->first_bad_batch == 0xFFFFFFFFu) good_bytes = n_members ? members[n_members - 1].u_off + members[n_members - 1].u_len : 0;
        else if (v->first_bad_batch == v->batches) good_bytes = members[first].u_off;
        v->good_bytes = good_bytes;
        v->batches++;
    }
}

// wavefronts of the inflate kernel the device holds at once (a member each): a commit of a whole multiple leaves no round of
// wavefronts half empty
uint32_t bgzf_wave_slots(int n_cu)
{
    static const bool wide = !(getenv("VGMI_INFLATE_WIDE") && getenv("VGMI_INFLATE_WIDE")[0] == '0');
    int nb = 0;
    const hipError_t e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bgzf_inflate_kernel<true>, 64 * INF_WAVES, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bgzf_inflate_kernel<false>, 64 * INF_WAVES, 0);
    if (e != hipSuccess || nb <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    return (uint32_t)nb * INF_WAVES * (uint32_t)n_cu;
}

hipError_t launch_bgzf_inflate(const uint8_t* comp, const BgzfMember* members, uint32_t n_members, uint8_t* out_base, uint32_t* status,
                               const uint32_t* crc_table, BgzfVerdict* verdict, hipStream_t s)
{
    static const bool wide = !(getenv("VGMI_INFLATE_WIDE") && getenv("VGMI_INFLATE_WIDE")[0] == '0');
    if (n_members) {
        if (wide)
            hipLaunchKernelGGL(bgzf_inflate_kernel<true>, dim3((n_members + INF_WAVES - 1) / INF_WAVES), dim3(64 * INF_WAVES), 0, s, comp, members, n_members, out_base,
                               status, crc_table);
        else
            hipLaunchKernelGGL(bgzf_inflate_kernel<false>, dim3((n_members + INF_WAVES - 1) / INF_WAVES), dim3(64 * INF_WAVES), 0, s, comp, members, n_members, out_base,
                               status, crc_table);
    }
    hipLaunchKernelGGL(bgzf_verdict_kernel, dim3(1), dim3(256), 0, s, members, status, n_members, verdict);
    return hipGetLastError();
}

}  // namespace vgk
