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
// Shape of the kernel: DEFLATE decoding is serial inside a member, so the wavefront runs it as ONE scalar-like thread --
// every quantity of the decoder (bit buffer, positions, the decoded symbol) is wave-uniform and lives in SGPRs; the 64
// lanes do what can be wide: Huffman table construction, LZ77 copies (a match of length n from distance d is the
// periodic extension of the d bytes before it, so lane i writes byte i from out[pos - d + i % d] -- no lane depends
// on another), and the CRC-32 of the result (one slice per lane, combined with GF(2) arithmetic).  Thousands of members
// are in flight per chunk; the kernel's rate is set by how many, not by one member's latency.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_kernels.h"

namespace vgk {

#define INF_LIT_BITS 10u
#define INF_DIST_BITS 8u
#define INF_MAXBITS 15
#define INF_RING 2048u

struct InfTables {                 // per wavefront, in LDS
    uint32_t multi[1u << INF_LIT_BITS];   // up to three LITERALS decoded from the same index: bytes 0..2 | count << 24 | bits << 26
    uint16_t lit[1u << INF_LIT_BITS];     // entry: symbol << 4 | code length (0 = code longer than INF_LIT_BITS or unused)
    uint16_t dist[1u << INF_DIST_BITS];
    uint8_t len[320];              // code lengths: 0..287 literal/length, 288..319 distance
    uint16_t sorted[320];          // symbols ordered by code (canonical decoding of the long codes)
    uint16_t count[2][INF_MAXBITS + 1];
    uint16_t offs[2][INF_MAXBITS + 1];
    uint8_t ring[INF_RING];        // the last INF_RING output bytes: LZ77 sources come from here, not from global memory --
                                   // a global read-back would wait (vmcnt) for every store still in flight
    uint8_t pad[64];               // where the lanes that have no byte of a literal run write theirs
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)uni((uint32_t)(v >> 32)) << 32 | uni((uint32_t)v); }
// The compressed input is read with SCALAR loads (constant address space, wave-uniform address): they count on lgkmcnt,
// so a refill never waits behind the byte stores of the output (a vector load's vmcnt(0) drained every store in flight:
// one memory round trip per 8 input bytes, most of the kernel's time in its first version)
typedef __attribute__((address_space(4))) const uint64_t inf_cu64;
__device__ __forceinline__ uint64_t ld64u(const uint64_t* p) { return *reinterpret_cast<inf_cu64*>((uintptr_t)p); }
typedef __attribute__((address_space(4))) const uint32_t inf_cu32;
__device__ __forceinline__ uint32_t ld32u(const uint32_t* p) { return *reinterpret_cast<inf_cu32*>((uintptr_t)p); }

__device__ __forceinline__ uint32_t bitrev(uint32_t code, uint32_t len) { return __builtin_bitreverse32(code) >> (32 - len); }

// Canonical Huffman tables of one alphabet from its code lengths (RFC 1951 3.2.2).  which: 0 literal/length, 1 distance.
// Returns false for an over-subscribed set of lengths (incomplete sets are legal only in the one-code cases zlib accepts;
// a code that is never assigned simply never matches and ends in the error path).
__device__ bool inf_build(InfTables& t, uint32_t which, uint32_t first, uint32_t n, uint32_t lane)
{
    uint16_t* const tab = which ? t.dist : t.lit;
    const uint32_t bits = which ? INF_DIST_BITS : INF_LIT_BITS;
    for (uint32_t i = lane; i < (1u << bits); i += 64) tab[i] = 0;
    if (lane <= INF_MAXBITS) t.count[which][lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
        const uint16_t e = (uint16_t)(s << 4 | l);
        for (uint32_t i = bitrev(code, l); i < (1u << bits); i += 1u << l) tab[i] = e;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// Sequence lines are runs of literals with 2-3 bit codes: one lookup yields as many literals as the index holds whole
// codes of (at most three).  Built from the single-symbol table: entry i = the literals coded by the low bits of i.
__device__ void inf_build_multi(InfTables& t, uint32_t lane)
{
    for (uint32_t i = lane; i < (1u << INF_LIT_BITS); i += 64) {
        uint32_t pos = 0, n = 0, bytes = 0;
        while (n < 3) {
            const uint32_t e = t.lit[i >> pos];          // the bits above the index are unknown: only codes that fit count
            const uint32_t l = e & 15u, sym = e >> 4;
            if (!l || l > INF_LIT_BITS - pos || sym >= 256) break;
            bytes |= sym << (8 * n);
            ++n;
            pos += l;
        }
        t.multi[i] = bytes | n << 24 | pos << 26;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// a code longer than the fast table: canonical decoding bit by bit (RFC 1951 3.2.2; rare by construction)
__device__ __forceinline__ int32_t inf_slow(const InfTables& t, uint32_t which, uint64_t bitbuf, uint32_t& len_out)
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

__device__ __constant__ uint8_t inf_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

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
__global__ __launch_bounds__(256) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const BgzfMember* __restrict__ members, uint32_t n_members,
                                                           uint8_t* __restrict__ out_base, uint32_t* __restrict__ status,
                                                           const uint32_t* __restrict__ crc_table)
{
    __shared__ InfTables tabs[4];
    __shared__ uint32_t s_crc[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_crc[i] = crc_table[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_block = uni(threadIdx.x >> 6);
    const uint32_t m = blockIdx.x * 4u + wave_in_block;
    if (m >= n_members) return;
    InfTables& t = tabs[wave_in_block];
    const uint8_t* in = comp + uni(members[m].c_off);
    const uint32_t in_len = uni(members[m].c_len);       // deflate bytes (header and trailer stripped by the host walk)
    uint8_t* out = out_base + uni(members[m].u_off);
    const uint32_t out_len = uni(members[m].u_len);      // ISIZE
    const uint32_t want_crc = uni(members[m].crc);
    // table indices are formed on the vector side (a mask the compiler cannot see through): the scalar unit is the busy one
    uint32_t vmask_lit, vmask_dist;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vmask_lit) : "s"((1u << INF_LIT_BITS) - 1u));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vmask_dist) : "s"((1u << INF_DIST_BITS) - 1u));
    // stores at offsets >= ISIZE are dropped by the hardware
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_len, 0x00020000);

    // ---- wave-uniform decoder state ----
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0, ip = 0, op = 0, err = 0;
    // Input: aligned 32-bit words appended whole to the 64-bit bit buffer whenever it holds 32 bits or fewer, fetched with
    // scalar loads three words ahead of use, so a refill never waits for memory (and costs eight scalar instructions: the
    // scalar unit is what this kernel is short of).  ip = input bytes appended so far.  The host pads every batch (>= 32
    // readable bytes behind the last member); bytes past in_len are only consumed by a damaged stream, which the
    // position check reports.
    const uint32_t* wq;
    uint32_t wa, wb, wc;
    auto reload = [&]() {          // (re)start at input byte ip: the bit buffer is empty
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
    reload();
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

    bool last = false;
    while (!last && !err) {
        refill();
        last = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) {            // stored
            take(bitcnt & 7u);      // to the byte boundary
            refill();
            const uint32_t len = take(16), nlen = take(16);
            if ((len ^ 0xFFFFu) != nlen) { err = 6; break; }
            // the bytes still in the bit buffer come first
            const uint32_t src = ip - (bitcnt >> 3);
            if (src + len > in_len || op + len > out_len) { err = 3; break; }
            for (uint32_t i = lane; i < len; i += 64) {
                const uint8_t b = in[src + i];
                out[op + i] = b;
                t.ring[(op + i) & (INF_RING - 1u)] = b;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            op += len;
            ip = src + len;
            bitbuf = 0;
            bitcnt = 0;
            reload();
            continue;
        }
        if (type == 3) { err = 7; break; }
        if (type == 1) {            // fixed Huffman codes (RFC 1951 3.2.6)
            for (uint32_t s = lane; s < 288; s += 64) t.len[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) t.len[288 + lane] = 5;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!inf_build(t, 0, 0, 288, lane) || !inf_build(t, 1, 288, 30, lane)) { err = 1; break; }
            inf_build_multi(t, lane);
        } else {                    // dynamic codes: the code lengths are themselves Huffman coded
            const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = 1; break; }
            refill();
            // code length alphabet: built in the distance slots (19 symbols), decoded through the distance table
            if (lane < 19) t.len[288 + lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = 0; i < hclen; ++i) {
                if (bitcnt < 3) refill();
                const uint32_t v = take(3);
                if (lane == 0) t.len[288 + uni(inf_clen_order[i])] = (uint8_t)v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!inf_build(t, 1, 288, 19, lane)) { err = 1; break; }
            // the hlit + hdist lengths, written to a staging area first (the code-length code occupies len[288..306])
            uint32_t idx = 0, prev = 0;
            uint8_t* const stage = reinterpret_cast<uint8_t*>(t.multi);      // rebuilt below; 256 bytes, continued in sorted[] (unused until the next build)
            uint8_t* const stage2 = reinterpret_cast<uint8_t*>(t.sorted);
            auto put = [&](uint32_t i, uint32_t v) {
                if (lane == 0) {
                    if (i < 256) stage[i] = (uint8_t)v;
                    else stage2[i - 256] = (uint8_t)v;
                }
            };
            while (idx < hlit + hdist && !err) {
                refill();
                uint32_t e = uni(t.dist[(uint32_t)bitbuf & vmask_dist]);
                uint32_t l = e & 15u, sym = e >> 4;
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
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // staging -> len[]: literal/length 0..hlit-1 (rest 0), distance 288..288+hdist-1 (rest 0)
            uint8_t mine[5];
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) {
                const uint32_t s = lane + 64 * q;      // 0..319
                uint32_t v = 0;
                if (s < 288) {
                    if (s < hlit) v = s < 256 ? stage[s] : stage2[s - 256];
                } else if (s - 288 < hdist) {
                    const uint32_t i = hlit + (s - 288);
                    v = i < 256 ? stage[i] : stage2[i - 256];
                }
                mine[q] = (uint8_t)v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) t.len[lane + 64 * q] = mine[q];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (uni(t.len[256]) == 0) { err = 1; break; }    // no end-of-block code
            if (!inf_build(t, 0, 0, 288, lane) || !inf_build(t, 1, 288, 30, lane)) { err = 1; break; }
            inf_build_multi(t, lane);
        }
        // ---- symbols of this block ----
        for (;;) {
            // Runs of literals first: up to three per look-up, one byte per lane.  The kernel is bound by SCALAR issue
            // (11.5 SALU + 3 branches per output byte, one scalar instruction per SIMD and four cycles: four members per
            // SIMD take 58 cycles per byte), so this loop keeps the scalar unit out of what the vector side can do: no exec
            // mask games (lanes without a byte store to an offset the buffer descriptor drops, and write their LDS byte
            // to a pad), no bounds branch (the descriptor ends at ISIZE; the position is checked once per run).
            // (the loop is written with its exit test at the bottom: one compare and one branch per look-up, where the
            // `for (;;) { ...; if (!n) break; ... }` form cost a compare, a select, two mask operations and two branches)
            need(32);         // a literal/length code (<= 15 bits) and its extra bits (<= 5)
            uint32_t mm = uni(t.multi[(uint32_t)bitbuf & vmask_lit]);
            while ((mm >> 24) & 3u) {
                const uint32_t n = (mm >> 24) & 3u;
                const bool mine = lane < n;
                const uint8_t b = (uint8_t)(mm >> (8u * (lane & 3u)));
                __builtin_amdgcn_raw_buffer_store_b8(b, orsrc, mine ? op + lane : 0xFFFFFFFFu, 0, 0);
                uint8_t* const cell = mine ? &t.ring[(op + lane) & (INF_RING - 1u)] : &t.pad[lane];
                *cell = b;
                op += n;
                take(mm >> 26);
                need(32);
                mm = uni(t.multi[(uint32_t)bitbuf & vmask_lit]);
            }
            if (op > out_len) { err = 3; break; }
            uint32_t e = uni(t.lit[(uint32_t)bitbuf & vmask_lit]);
            uint32_t l = e & 15u;
            int32_t sym = (int32_t)(e >> 4);
            if (!l) {
                sym = inf_slow(t, 0, bitbuf, l);
                sym = (int32_t)uni((uint32_t)sym);
                l = uni(l);
                if (sym < 0) { err = 2; break; }
            }
            take(l);
            if (sym < 256) {
                if (op >= out_len) { err = 3; break; }
                if (lane == 0) {
                    out[op] = (uint8_t)sym;
                    t.ring[op & (INF_RING - 1u)] = (uint8_t)sym;
                }
                ++op;
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = 2; break; }
            // length codes 257..285 (RFC 1951 3.2.5) in closed form: no table fetch on the serial path
            uint32_t len;
            if (sym < 8) len = 3 + (uint32_t)sym;
            else if (sym == 28) len = 258;
            else {
                const uint32_t e = ((uint32_t)sym >> 2) - 1;
                len = ((4u + ((uint32_t)sym & 3u)) << e) + 3u + take(e);
            }
            need(32);     // a distance code (<= 15 bits) and its extra bits (<= 13)
            uint32_t de = uni(t.dist[(uint32_t)bitbuf & vmask_dist]);
            uint32_t dl = de & 15u;
            int32_t dsym = (int32_t)(de >> 4);
            if (!dl) {
                dsym = inf_slow(t, 1, bitbuf, dl);
                dsym = (int32_t)uni((uint32_t)dsym);
                dl = uni(dl);
                if (dsym < 0) { err = 2; break; }
            }
            take(dl);
            if (dsym >= 30) { err = 2; break; }
            uint32_t dist;
            if (dsym < 4) dist = 1 + (uint32_t)dsym;
            else {
                const uint32_t e = ((uint32_t)dsym >> 1) - 1;
                dist = ((2u + ((uint32_t)dsym & 1u)) << e) + 1u + take(e);
            }
            if (dist > op) { err = 2; break; }
            if (op + len > out_len) { err = 3; break; }
            if (dist <= INF_RING - 258u) {
                // sources from the LDS ring (the LDS pipe is in order per wavefront: earlier ring writes of any lane are seen)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = lane; i < len; i += 64) {
                    const uint8_t b = t.ring[(op - dist + (dist >= len ? i : i % dist)) & (INF_RING - 1u)];
                    out[op + i] = b;
                    t.ring[(op + i) & (INF_RING - 1u)] = b;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
            } else {
                // far match: read the output back; every earlier store of the wave must have landed
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint8_t* const src = out + op - dist;
                for (uint32_t i = lane; i < len; i += 64) {
                    const uint8_t b = src[dist >= len ? i : i % dist];
                    out[op + i] = b;
                    t.ring[(op + i) & (INF_RING - 1u)] = b;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            op += len;
        }
        if (ip - (bitcnt >> 3) > in_len) err = 3;   // the block ran past the member's deflate bytes
    }
    if (!err && op != out_len) err = 4;
    if (!err && out_len) {
        // CRC-32 of the output: one slice per lane, then crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / p(x)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t slice = (out_len + 63u) / 64u;
        const uint32_t b = lane * slice < out_len ? lane * slice : out_len;
        const uint32_t e = b + slice < out_len ? b + slice : out_len;
        uint32_t c = 0xFFFFFFFFu;
        for (uint32_t i = b; i < e; ++i) c = s_crc[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
        t.multi[lane] = ~c;      // (the decode tables are no longer needed)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const uint32_t x_full = gf2_x8n(slice);
            uint32_t crc = t.multi[0];
            for (uint32_t i = 1; i < 64; ++i) {
                const uint32_t bi = i * slice;
                if (bi >= out_len) break;
                const uint32_t li = bi + slice <= out_len ? slice : out_len - bi;
                crc = gf2_mul(li == slice ? x_full : gf2_x8n(li), crc) ^ t.multi[i];
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

hipError_t launch_bgzf_inflate(const uint8_t* comp, const BgzfMember* members, uint32_t n_members, uint8_t* out_base, uint32_t* status,
                               const uint32_t* crc_table, BgzfVerdict* verdict, hipStream_t s)
{
    if (n_members) hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((n_members + 3) / 4), dim3(256), 0, s, comp, members, n_members, out_base, status, crc_table);
    hipLaunchKernelGGL(bgzf_verdict_kernel, dim3(1), dim3(256), 0, s, members, status, n_members, verdict);
    return hipGetLastError();
}

}  // namespace vgk
