// vgmi_xtable.h -- device functions of the table keyed by the read's grid 16-mer (see vgmi_xtable.hip), shared with the
// generic kernels of vgmi_kernels.hip (ragged tails of a block count through the same table and counters).
#ifndef VGMI_XTABLE_H
#define VGMI_XTABLE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    line = ((uint64_t)h * t.n_lines) >> 32;          // the h-values of one line are consecutive: their low tag_bits differ
    const uint32_t tag = h & ((1u << t.tag_bits) - 1u);
    want = (uint64_t)j | (uint64_t)f << 4 | (uint64_t)tag << 26;
}

__device__ __forceinline__ void xt_count(const XTableView& t, uint64_t kmer, uint32_t w)
{
    uint64_t line, want;
    xt_key(t, kmer, w, line, want);
    const uint64_t key_mask = (1ULL << t.id_shift) - 1;
    const uint32_t j = (uint32_t)want & 15u;
    for (;;) {
        const unsigned long long* L = t.lines + (line << 4);
        unsigned long long e = L[j];
        bool hit = e != XT_EMPTY && ((e ^ want) & key_mask) == 0;
        bool full = false;
        if (!hit && e != XT_EMPTY) {          // slot j' holds another k-mer: the spill slots
            const ulonglong2 s01 = *reinterpret_cast<const ulonglong2*>(L + 12);
            const ulonglong2 s23 = *reinterpret_cast<const ulonglong2*>(L + 14);
            const unsigned long long sp[4] = {s01.x, s01.y, s23.x, s23.y};
            full = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (sp[q] == XT_EMPTY) full = false;
                else if (((sp[q] ^ want) & key_mask) == 0) { hit = true; e = sp[q]; }
            }
        }
        if (hit) {
            // no return value: nothing waits for the atomic.  The clamp to 255 happens at read-out; the host pulls counters
            // far above it back (xclamp_kernel, every 2^31 submitted bytes) so that none can ever wrap
            __hip_atomic_fetch_add(t.counts + (e >> t.id_shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (!full) return;
        line = line + 1 == t.n_lines ? 0 : line + 1;
    }
}

// id of a k-mer (either strand), or 0xFFFFFFFF: the lookup of xt_count without the count
__device__ __forceinline__ uint32_t xt_find(const XTableView& t, uint64_t kmer)
{
    uint64_t line, want;
    xt_key(t, kmer, 0, line, want);
    const uint64_t key_mask = (1ULL << t.id_shift) - 1;
    const uint32_t j = (uint32_t)want & 15u;
    for (;;) {
        const unsigned long long* L = t.lines + (line << 4);
        const unsigned long long e = L[j];
        if (e == XT_EMPTY) return 0xFFFFFFFFu;
        if (((e ^ want) & key_mask) == 0) return (uint32_t)(e >> t.id_shift);
        bool full = true;
        for (int q = 12; q < 16; ++q) {
            const unsigned long long s = L[q];
            if (s == XT_EMPTY) full = false;
            else if (((s ^ want) & key_mask) == 0) return (uint32_t)(s >> t.id_shift);
        }
        if (!full) return 0xFFFFFFFFu;
        line = line + 1 == t.n_lines ? 0 : line + 1;
    }
}

}  // namespace vgk
#endif
