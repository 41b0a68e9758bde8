// vgmi_xtable.h -- device functions of the table keyed by the read's grid 16-mer (see vgmi_xtable.hip), shared with the
// generic kernels of vgmi_kernels.hip (ragged tails of a block count through the same table and counters).
#ifndef VGMI_XTABLE_H
#define VGMI_XTABLE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_ctable.h"
#include "vgmi_device.h"
#include "vgmi_kernels.h"

namespace vgk {

#define XT_EMPTY 0xFFFFFFFFFFFFFFFFULL

__device__ __forceinline__ uint32_t xt_hash(uint32_t cx)      // bijection of 32 bits
{
    uint32_t h = cx * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}

// (line, match word) of the window that has grid 16-mer `x` (as read) w bases before its end; `kmer` = the window
__device__ __forceinline__ void xt_key(const XTableView& t, uint64_t kmer, uint32_t w, uint64_t& line, uint64_t& want)
{
    const uint32_t x = (uint32_t)(kmer >> (2 * w));
    const uint32_t rc = vg_revcomp16(x);
    const bool as_is = x <= rc;
    const uint32_t low = (uint32_t)kmer & ((1u << (2 * w)) - 1u);          // w bases behind X
    const uint32_t high = (uint32_t)(kmer >> (2 * w + 32));                 // 11 - w bases in front of X
    uint32_t f = (high << (2 * w)) | low;                                   // 22 bits
    uint32_t j = w;
    if (!as_is) {
        f = (uint32_t)vg_revcomp(f, 11);
        j = 11u - w;
    }
    const uint32_t h = xt_hash(as_is ? x : rc);
    line = ((uint64_t)h * t.n_lines) >> 32;          // the h-values of XT_HOPS + 1 neighbouring lines are consecutive: their low tag_bits differ
    // bits 26 .. id_shift - 1 are the tag; the rest of h rides along above (every compare masks it off) for xt_kmer_of
    want = (uint64_t)j | (uint64_t)f << 4 | (uint64_t)h << 26;
}

// The k-mers that found no room in their home line or the XT_HOPS lines behind it (16-mers of repeats: hundreds of graph
// k-mers under one X) live in a small exact table keyed by the canonical k-mer.  A lookup gets there only after it has
// seen XT_HOPS + 1 full lines, exactly as the insert did.
__device__ __forceinline__ uint32_t xt_over_hash(uint64_t canon) { return xt_hash((uint32_t)canon ^ xt_hash((uint32_t)(canon >> 32))); }

__device__ __forceinline__ uint32_t xt_over_find(const XTableView& t, uint64_t kmer)
{
    if (!t.over) return 0xFFFFFFFFu;
    const uint64_t rc = vg_revcomp(kmer, t.k);
    const uint64_t canon = kmer < rc ? kmer : rc;
    uint32_t s = xt_over_hash(canon) & t.over_mask;
    for (;;) {
        const ulonglong2 e = t.over[s];
        if (e.x == XT_EMPTY) return 0xFFFFFFFFu;
        if (e.x == canon) return (uint32_t)e.y;
        s = (s + 1) & t.over_mask;
    }
}

// The rest of a lookup whose home slot j' (entry e0, not empty) holds another k-mer: spill slots of the home line, then
// slot j' and the spill slots of the next XT_HOPS lines (a lookup moves on only past a line whose four spill slots are
// taken), then the overflow table.  tag_bits covers the h-values of XT_HOPS + 1 consecutive lines, so a (j', f, tag) match
// in any of them is the same 16-mer, hence the same k-mer.  Returns the matching entry or XT_EMPTY; over_id gets the id
// when the overflow table answers.
__device__ __forceinline__ uint32_t xt_hash_inv(uint32_t h)
{
    h ^= (h >> 13) ^ (h >> 26);
    h *= (uint32_t)vg_inv_odd(0x85EBCA77u);
    h ^= (h >> 15) ^ (h >> 30);
    return h * (uint32_t)vg_inv_odd(0x9E3779B1u);
}

// the k-mer (one of its two strands) that `want` stands for: it carries h(X) whole, h is a bijection, and j', f are the
// window around X -- used on the overflow path only, so the count kernels need not keep the window around
__device__ __forceinline__ uint64_t xt_kmer_of(uint64_t want)
{
    const uint64_t x = xt_hash_inv((uint32_t)(want >> 26));
    const uint32_t j = (uint32_t)want & 15u, f = (uint32_t)(want >> 4) & 0x3FFFFFu;
    const uint64_t low = f & ((1u << (2 * j)) - 1u), high = f >> (2 * j);
    return high << (2 * j + 32) | x << (2 * j) | low;
}

__device__ __forceinline__ unsigned long long xt_lookup_rest(const XTableView& t, uint64_t line, uint64_t want, uint32_t& over_id)
{
    const uint64_t key_mask = (1ULL << t.id_shift) - 1;
    const uint32_t j = (uint32_t)want & 15u;
    over_id = 0xFFFFFFFFu;
    for (uint32_t hop = 0;; ++hop) {
        const unsigned long long* L = t.lines + ((line + hop) << 4);
        if (hop) {
            const unsigned long long e = L[j];
            if (e == XT_EMPTY) return XT_EMPTY;
            if (((e ^ want) & key_mask) == 0) return e;
        }
        const ulonglong2 s01 = *reinterpret_cast<const ulonglong2*>(L + 12);
        const ulonglong2 s23 = *reinterpret_cast<const ulonglong2*>(L + 14);
        const unsigned long long sp[4] = {s01.x, s01.y, s23.x, s23.y};
        bool full = true;
        unsigned long long hit = XT_EMPTY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (sp[q] == XT_EMPTY) full = false;
            else if (((sp[q] ^ want) & key_mask) == 0) hit = sp[q];
        }
        if (hit != XT_EMPTY || !full) return hit;
        if (hop == XT_HOPS) break;
    }
    over_id = xt_over_find(t, xt_kmer_of(want));
    return XT_EMPTY;
}

// counter id of the window `kmer` (as read; grid 16-mer w bases before its end), or 0xFFFFFFFF
__device__ __forceinline__ uint32_t xt_lookup(const XTableView& t, uint64_t kmer, uint32_t w)
{
    uint64_t line, want;
    xt_key(t, kmer, w, line, want);
    const uint64_t key_mask = (1ULL << t.id_shift) - 1;
    const unsigned long long e = t.lines[(line << 4) + ((uint32_t)want & 15u)];
    if (e == XT_EMPTY) return 0xFFFFFFFFu;
    if (((e ^ want) & key_mask) == 0) return (uint32_t)(e >> t.id_shift);
    uint32_t over_id;
    const unsigned long long r = xt_lookup_rest(t, line, want, over_id);
    return r != XT_EMPTY ? (uint32_t)(r >> t.id_shift) : over_id;
}

__device__ __forceinline__ void xt_count(const XTableView& t, uint64_t kmer, uint32_t w)
{
    const uint32_t id = xt_lookup(t, kmer, w);
    // no return value: nothing waits for the atomic.  The clamp to 255 happens at read-out; the host pulls counters
    // far above it back (xclamp_kernel, every 2^31 submitted bytes) so that none can ever wrap
    if (id != 0xFFFFFFFFu) __hip_atomic_fetch_add(t.counts + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// id of a k-mer (either strand), or 0xFFFFFFFF: the lookup of xt_count without the count
__device__ __forceinline__ uint32_t xt_find(const XTableView& t, uint64_t kmer) { return xt_lookup(t, kmer, 0); }

// ---- context table (vgmi_ctable.h, vgmi_ctable.hip): one k-mer looked up as a one-window context -----------------------
// counter id of a k-mer (either strand), or 0xFFFFFFFF: the generic kernels' ragged tails count through the same table
__device__ __forceinline__ uint32_t ct_find(const XTableView& t, uint64_t kmer)
{
    uint32_t cx, cl, cr, vs;
    ct_orient_kmer(kmer, cx, cl, cr, vs, t.k);
    const uint32_t f = ct_flank(t.k), ex = ct_excess(t.k);
    const uint64_t b0 = ((uint64_t)ct_hash(cx) * t.n_buckets) >> 32;
    for (uint32_t hop = 0; hop <= CT_HOPS; ++hop) {
        const uint32_t* B = reinterpret_cast<const uint32_t*>(t.cb + ((b0 + hop) << 2));
        const uint4 xs = *reinterpret_cast<const uint4*>(B);
        const uint32_t x[4] = {xs.x, xs.y, xs.z, xs.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (x[q] != cx) continue;
            const CtEntry e = {cx, B[4 + 3 * q], B[5 + 3 * q], B[6 + 3 * q]};
            const uint32_t h = ct_match(e, cx, cl, cr, f, ex) & vs;
            if (h) return ct_id(e, (uint32_t)__builtin_ctz(h));
        }
        if (xs.w == 0xFFFFFFFFu || !(B[5] & ct_mark(cx))) return 0xFFFFFFFFu;      // not full, or full and nothing with this mark went on
    }
    return xt_over_find(t, kmer);
}

__device__ __forceinline__ void ct_count(const XTableView& t, uint64_t kmer)
{
    const uint32_t id = ct_find(t, kmer);
    if (id != 0xFFFFFFFFu) __hip_atomic_fetch_add(t.counts + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace vgk
#endif
