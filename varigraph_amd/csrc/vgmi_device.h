// vgmi_device.h -- arithmetic shared by host and gfx950 device code.
//
// Reference semantics restated here (file:line under the reference tree):
//   hash64                include/hash64.hpp:5-14
//   seq_nt4_table         include/seq_nt4_table.hpp:5-22
//   MurmurHash3_x64_128   src/MurmurHash3.cpp:255-332 (len == 8 path), summed as in
//                         src/counting_bloom_filter.cpp:90-98
// plus the inverse of hash64 (not in the reference): hash64 is a bijection of [0, 2^(2k)), so
// the device table stores canonical k-mers instead of hashed keys and the read kernel never
// evaluates hash64 -- same membership, same counters, ~40 fewer integer ops per k-mer.
#ifndef VGMI_DEVICE_H
#define VGMI_DEVICE_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VG_HD __host__ __device__ static inline
#else
#define VG_HD static inline
#endif

#define VG_EMPTY 0xFFFFFFFFFFFFFFFFULL
// The k-mer word of a slot: bits 0..55 canonical k-mer (k <= 28), VG_EMPTY when free, and two flags:
//   VG_SLOT_CHAIN  some key probed PAST this slot when it was inserted.  A lookup that finds another key here goes
//                  on only if the flag is set: an absent k-mer stops at the first probe almost always.
//   VG_SLOT_SAT    (compact format) the slot's counter has reached the 255 clamp: later hits skip their atomic,
//                  as the reference skips its increment.  Cleared by vgmi_counts_reset.
// The all-ones pattern cannot be a stored k-mer: its low 2k bits are T^k, whose canonical form is A^k = 0.
#define VG_SLOT_KMER_MASK ((1ULL << 56) - 1)
#define VG_SLOT_CHAIN (1ULL << 62)
#define VG_SLOT_SAT (1ULL << 63)

// Two table formats (chosen at upload, ImageHeader::slot_bytes):
//   16 B  VgSlot {k-mer word, count, key_index}: one dwordx4 per probe.  In-slot counters for small graphs, dense
//         per-key counters (TableView::counts[key_index]) for large ones.
//    8 B  compact: the k-mer word only, counters in a parallel array indexed by slot (TableView::counts[slot]).
//         k = 27 graphs of <= 65 536 k-mers (the LDS-filter kernel): half the table bytes at twice the slots.
struct __attribute__((aligned(16))) VgSlot {
    unsigned long long canon;  // k-mer word
    unsigned int count;        // occurrences seen this sample (clamped to 255 on read-out)
    unsigned int key_index;    // index of the key in the uploaded keys[]
};

// include/hash64.hpp:5-14
VG_HD uint64_t vg_hash64(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// multiplicative inverse of an odd a modulo 2^64 (Newton)
VG_HD uint64_t vg_inv_odd(uint64_t a)
{
    uint64_t x = a;  // correct to 3 bits
    for (int i = 0; i < 6; ++i) x *= 2 - a * x;
    return x;
}

// inverse of vg_hash64 on [0, mask]; mask = 2^(2k)-1
VG_HD uint64_t vg_hash64_inv(uint64_t h, uint64_t mask)
{
    uint64_t x;
    // key + (key << 31) = key * (2^31 + 1)
    h = (h * vg_inv_odd((1ULL << 31) + 1)) & mask;
    // key ^ key >> 28
    x = h; x = h ^ (x >> 28); x = h ^ (x >> 28); h = x;
    // key * 21
    h = (h * vg_inv_odd(21)) & mask;
    // key ^ key >> 14
    x = h; x = h ^ (x >> 14); x = h ^ (x >> 14); x = h ^ (x >> 14); x = h ^ (x >> 14); h = x;
    // key * 265
    h = (h * vg_inv_odd(265)) & mask;
    // key ^ key >> 24
    x = h; x = h ^ (x >> 24); x = h ^ (x >> 24); x = h ^ (x >> 24); h = x;
    // ~key + (key << 21) = key * (2^21 - 1) - 1
    h = ((h + 1) * vg_inv_odd((1ULL << 21) - 1)) & mask;
    return h;
}

// include/seq_nt4_table.hpp:5-22 as a function (the kernels stage the table in LDS)
VG_HD uint32_t vg_nt4(uint32_t c)
{
    if (c < 4) return c;
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

VG_HD uint64_t vg_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
VG_HD uint64_t vg_fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
// h1+h2 of MurmurHash3_x64_128(&key, 8, (uint32_t)seed)
VG_HD uint64_t vg_murmur_sum(uint64_t key, uint32_t seed32)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed32, h2 = seed32;
    uint64_t k1 = key;
    k1 *= c1; k1 = vg_rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = vg_fmix64(h1); h2 = vg_fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 + h2;
}

// ---- hashes of the device table (not in the reference; any function works, both sides of the
// table -- build and probe -- use these) ------------------------------------------------------
// Filter hash: three 24-bit multiplies (full-rate v_mul_u32_u24 / v_mad_u32_u24 on CDNA).
VG_HD uint32_t vg_mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
// word selector: use the TOP bits
VG_HD uint32_t vg_fhash_word(uint64_t canon)
{
    const uint32_t lo = (uint32_t)canon, mid = (uint32_t)(canon >> 24), hi = (uint32_t)(canon >> 48);
    return vg_mul24(lo, 0x9E3779u) + vg_mul24(mid, 0x85EBCBu) + vg_mul24(hi, 0xC2B2AFu);
}
// bit selector inside the 32-bit word (two bits per key).  Small filters (<= 2^15 words, the
// LDS-resident case) index with the top <= 15 bits of the word hash, so bits [7,17) of the same
// hash are free to pick the two bit positions; larger filters take them from a second product.
VG_HD uint32_t vg_fhash_bits_small(uint32_t word_hash)
{
    return (1u << ((word_hash >> 12) & 31u)) | (1u << ((word_hash >> 7) & 31u));
}
VG_HD uint32_t vg_fhash_bits_large(uint64_t canon)
{
    const uint32_t lo = (uint32_t)canon, mid = (uint32_t)(canon >> 24);
    const uint32_t g = vg_mul24(lo, 0x5BD1E9u) + vg_mul24(mid, 0x27D4EBu);
    return (1u << (g >> 27)) | (1u << ((g >> 22) & 31u));
}
VG_HD uint32_t vg_fhash_bits(uint64_t canon, uint32_t filter_words_log2)
{
    return filter_words_log2 <= 15 ? vg_fhash_bits_small(vg_fhash_word(canon)) : vg_fhash_bits_large(canon);
}
// reverse complement of a 2-bit packed k-mer (first base most significant)
VG_HD uint64_t vg_revcomp(uint64_t x, uint32_t k)
{
    uint64_t r = ~x;
#if defined(__HIP_DEVICE_COMPILE__)
    r = __builtin_bitreverse64(r);  // two v_bfrev_b32; bits of each 2-bit field are now swapped
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
#else
    // reverse the order of the 32 two-bit fields of the 64-bit word
    r = ((r >> 2) & 0x3333333333333333ULL) | ((r & 0x3333333333333333ULL) << 2);
    r = ((r >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((r & 0x0F0F0F0F0F0F0F0FULL) << 4);
    r = ((r >> 8) & 0x00FF00FF00FF00FFULL) | ((r & 0x00FF00FF00FF00FFULL) << 8);
    r = ((r >> 16) & 0x0000FFFF0000FFFFULL) | ((r & 0x0000FFFF0000FFFFULL) << 16);
    r = (r >> 32) | (r << 32);
#endif
    return r >> (64 - 2 * k);
}

// Grid filter of the fast read kernels (k = 27).  Every 27-mer contains exactly one 16-mer that
// ENDS at a stream position divisible by 12 (27 - 16 + 1 = 12).  The filter is a blocked Bloom
// filter (3 bits in one 32-bit word) over every canonical 16-mer found at any of the 12 offsets of any
// graph k-mer.  A read therefore probes it only at every 12th position -- in the
// orientation it is read in, no reverse complement, no canonical min -- and a hit makes the 12
// k-mers containing that 16-mer candidates for the exact table.  Small graphs keep 2^15 words
// (128 KiB) in LDS; larger graphs use a global bitmap of >= 32 bits per key.
#define VG_GRID_STEP 12u
#define VG_GRID_MER 16u
#define VG_GRID_LDS_WORDS_LOG2 15u
#define VG_GRID_LDS_WORDS (1u << VG_GRID_LDS_WORDS_LOG2)
#define VG_GRID_LDS_MAX_KEYS 65536u
// reverse complement of a 16-mer (32 bits, first base most significant)
VG_HD uint32_t vg_revcomp16(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r = __builtin_bitreverse32(x);
#else
    uint32_t r = x;
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    r = ((r >> 2) & 0x33333333u) | ((r & 0x33333333u) << 2);
    r = ((r >> 4) & 0x0F0F0F0Fu) | ((r & 0x0F0F0F0Fu) << 4);
    r = ((r >> 8) & 0x00FF00FFu) | ((r & 0x00FF00FFu) << 8);
    r = (r >> 16) | (r << 16);
#endif
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);  // bits of each 2-bit field back in order
    return ~r;
}
// The filter is keyed on the CANONICAL 16-mer (min with its reverse complement): a graph k-mer and its
// reverse complement then share their 12 entries, which halves the fill and cuts false candidates ~6x.
//
// Global-memory variant (large graphs): entries are 64 bits.  The low word is the blocked Bloom word; the high
// word holds OFFSET bits: bit (rot + b) mod 32 says "some graph k-mer has this 16-mer at offset b", where the
// offset is counted from the k-mer's end when the 16-mer is canonical as it stands and from its start otherwise
// (so a k-mer and its reverse complement set the same bit), and rot is 5 more hash bits.  A candidate run then
// probes only the windows whose offset bit is set -- about half of the 12 table probes of a true run, and two
// thirds of a false one, never leave the CU.
VG_HD void vg_grid_probe(uint32_t mer16, uint32_t words_log2, uint64_t& word, uint32_t& mask, uint32_t& rot, bool& as_is)
{
    const uint32_t rc = vg_revcomp16(mer16);
    as_is = mer16 <= rc;
    const uint32_t cm = as_is ? mer16 : rc;
    // one 32 x 32 -> 64 multiply: word index = top bits of the LOW product word (multiplicative hashing; the top of
    // the full product would be monotonic in cm), bit choices from the low bits of the high word (the product's
    // well-mixed middle)
    const uint64_t x = (uint64_t)cm * 0x9E3779B1u;
    word = (uint32_t)x >> (32 - words_log2);
    const uint32_t y = (uint32_t)(x >> 32);
    mask = (1u << (y & 31u)) | (1u << ((y >> 5) & 31u)) | (1u << ((y >> 10) & 31u));
    rot = (y >> 15) & 31u;
}
VG_HD void vg_grid_probe(uint32_t mer16, uint32_t words_log2, uint64_t& word, uint32_t& mask)
{
    uint32_t rot;
    bool as_is;
    vg_grid_probe(mer16, words_log2, word, mask, rot, as_is);
}

// Small graphs (<= VG_GRID_LDS_MAX_KEYS k-mers: the filter lives in LDS, the table in L2) take a coarser grid: every 27-mer
// contains exactly one 12-mer that ends at a stream position = 15 (mod 16) (27 - 12 + 1 = 16), so a lane of count27s_kernel
// owns 16 bytes (one dwordx4) and one grid position per row, and the per-position work -- neighbour exchange, validity,
// canonical form, hash, filter word, enqueue -- is paid once per 16 bases instead of once per 12.  A 12-mer is specific
// enough only while the key set is small: 6.5e4 k-mers put <= ~7e4 of the 1.7e7 possible 12-mers into the filter (a random
// position matches by chance 0.4 % of the time, next to ~1 % Bloom false positives); a chr20-class graph would make every
// position a candidate, which is why large graphs keep the 16-mer grid.  Same blocked Bloom form: 3 bits in one of 2^15
// 32-bit words, keyed on the canonical 12-mer; both products are 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate).
#define VG_GRID12_STEP 16u
#define VG_GRID12_MER 12u
VG_HD uint32_t vg_revcomp12(uint32_t x)
{
    return vg_revcomp16(x) >> 8;     // the 12-mer sits in the low 24 bits: its complement-reverse lands in the top 24
}
VG_HD void vg_grid12_probe(uint32_t mer12, uint32_t& word, uint32_t& mask)
{
    const uint32_t rc = vg_revcomp12(mer12);
    const uint32_t cm = mer12 < rc ? mer12 : rc;
    const uint32_t h1 = vg_mul24(cm, 0x9E3779u), h2 = vg_mul24(cm, 0x85EBCBu);
    word = h1 >> (32 - VG_GRID_LDS_WORDS_LOG2);
    mask = (1u << (h2 >> 27)) | (1u << ((h2 >> 22) & 31u)) | (1u << ((h2 >> 17) & 31u));
}

// bucket hash of the path table's 12-mer index (build_ptable, count27s_kernel: once per run, not per position -- two 32-bit multiplies
// are affordable here, and the plain multiplicative hash left three times the expected number of overfull buckets)
VG_HD uint32_t vg_idx_hash(uint32_t cx)
{
    uint32_t h = cx * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}

// slot hash of the exact table (evaluated only for filter passes): three 24-bit multiplies (full rate on CDNA; the 32-bit
// v_mul_lo_u32 is quarter rate) and a fold; home slots are 32-bit (tables beyond 2^32 slots stay correct, linear probing
// just starts in the low part)
VG_HD uint64_t vg_thash(uint64_t kmer)
{
    uint32_t h = vg_mul24((uint32_t)kmer, 0x9E3779u) + vg_mul24((uint32_t)(kmer >> 24), 0x85EBCBu) + vg_mul24((uint32_t)(kmer >> 48), 0xC2B2AFu);
    h ^= h >> 15;
    return h;
}

// locality variant of the home slot (k = 27, tables that live in HBM): the bucket is drawn from the k-mer's MINIMISER
// -- the smallest hash among its twelve canonical 16-mers -- so the k-mers of consecutive read positions, which share
// their minimiser for ~6 positions on average, have their home slots inside the same 1 << bucket_log2 slots (one or
// two 128-byte lines) instead of twelve random sectors.  The position inside the bucket comes from the whole k-mer.
// `rc` is the reverse complement of `canon` (window off of one is the reverse complement of window 11 - off of the
// other, so the twelve canonical 16-mers need no per-window reversal).
// With `by_offset` the place inside the bucket is not hashed either: it is 2 * o + one hash bit, o = how far the minimiser
// sits from the k-mer's end, counted in the minimiser's OWN canonical orientation (so it does not depend on which strand
// of the k-mer is the canonical one).  Along a read the k-mers that share a minimiser occurrence have o = e, e + 1, ...
// (or 11 - e, 10 - e, ...): their slots -- and their per-slot counters -- are neighbours, a candidate run's probes read
// one or two lines and its counter updates leave as one or two atomic requests (tools/ubench_mem2: the memory system
// charges per request, the lanes of a request are free).
VG_HD uint64_t vg_thash_local(uint64_t canon, uint64_t rc, uint32_t bucket_log2, bool by_offset = false)
{
    uint32_t best = 0xFFFFFFFFu, best_o = 0;
    for (uint32_t off = 0; off < 12; ++off) {
        const uint32_t m = (uint32_t)(canon >> (2 * off)), r = (uint32_t)(rc >> (2 * (11 - off)));
        uint32_t h = (m < r ? m : r) * 0x9E3779B1u;
        h ^= h >> 15;
        const uint32_t o = m <= r ? off : 11u - off;
        best_o = h < best ? o : best_o;
        best = h < best ? h : best;
    }
    uint32_t b = best * 0x85EBCA77u;
    b ^= b >> 13;
    const uint32_t sub = (uint32_t)canon * 0x9E3779B1u + (uint32_t)(canon >> 32) * 0x85EBCA77u;
    // 16 offset places per bucket (12 used), each 1 << (bucket_log2 - 4) slots wide: the k-mers that share minimiser AND
    // offset (the alleles of one site, combinations with a neighbouring site) spread over them by hash
    if (by_offset && bucket_log2 >= 5) {
        const uint32_t w = bucket_log2 - 4;
        return ((uint64_t)b << bucket_log2) | (best_o << w) | (sub >> (32 - w));
    }
    return ((uint64_t)b << bucket_log2) | (sub >> (32 - bucket_log2));
}

#endif
