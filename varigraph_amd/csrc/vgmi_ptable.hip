// vgmi_ptable.hip -- the PATH TABLE of small graphs (k = 27, <= 65 536 k-mers: BASELINE config 2, the bench; round 5: odd k = 19 .. 25
// as well, TableView::k): what count27s_kernel (vgmi_kernels.hip) looks candidate runs up in.  (The text below speaks of k = 27.)
//
// Same reference behaviour as every count kernel here: src/kmer.cpp:140-142 (exact membership of the canonical k-mer),
// src/fastq_kmer.cpp:128-139 (saturating count).
//
// Why.  A candidate run is 16 consecutive k-mers of a read (the windows around one grid 12-mer X, vgmi_device.h).  Looked up
// one by one in the exact hash table they are 16 probes with 16 canonical forms and 16 hashes, on 16 lanes -- and every lane
// of a drain step pays the step's whole instruction stream: the drain cost more than the scan (6.0 ms per 1e8 reads, 2.7 of them
// scan; measured with the lookups compiled out).  But the 16 k-mers are not independent: they are CONSECUTIVE k-mers of a
// haplotype path, and so are the graph's k-mers they may equal.  So the graph's k-mers are laid out along their unitigs
// (chains of k-mers that follow each other uniquely in the key set: the 27 k-mers across one allele of a SNP are one chain of
// 53 bases) as a SEQUENCE, S, once per orientation, and an index maps a canonical 12-mer to the (up to two: two alleles)
// places of S where it occurs.  ONE LANE then checks a whole run: it fetches the 42 bases of S around the place, XORs them
// with the run's 42 bases, and the first mismatch on either side of the 12-mer bounds the windows that are graph k-mers --
// all 16 answers from two find-first-bit instructions, exact because the compare is on every base.  Per position of S a bit
// says whether a graph k-mer starts there (VB), another whether its counter is saturated (SB, per sample), and SLOT holds its
// hash-table slot: counters stay per table slot, so read-out, the generic kernels (ragged tails, k != 27) and the image format
// do not change.  This file orders the k-mers (device kernels, as vgmi_xtable.hip numbers counters along paths);
// build_ptable (vgmi_api_table.cpp) lays S, VB, SLOT and the index out on the host -- at most 65 536 k-mers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_device.h"
#include "vgmi_kernels.h"

namespace vgk {

#define PT_NONE 0xFFFFFFFFu
#define PT_MASK54 VG_SLOT_KMER_MASK      // (the k-mer bits of a compact slot: 56, k <= 28)

// slot of a canonical k-mer in the compact table, or PT_NONE
__device__ __forceinline__ uint32_t pt_find(const TableView& t, uint64_t canon)
{
    // (large graphs -- the context table's numbering -- keep their home slots in minimiser buckets)
    uint64_t s = (t.home_bucket_log2 ? vg_thash_local(canon, vg_revcomp(canon, t.k), t.home_bucket_log2, t.home_by_offset != 0) : vg_thash(canon)) & t.cap_mask;
    for (;;) {
        const uint64_t c = t.slots8[s];
        if (c == VG_EMPTY) return PT_NONE;
        if ((c & VG_SLOT_KMER_MASK) == canon) return (uint32_t)s;
        if (!(c & VG_SLOT_CHAIN)) return PT_NONE;
        s = (s + 1) & t.cap_mask;
    }
}

__global__ void pt_key_of_slot_kernel(const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key_of_slot[key_slot[i]] = (uint32_t)i;
}

// The key set as a bidirected de Bruijn graph.  (key i, side): side 1 = right of the canonical k-mer (its successors), 0 = left.
// link = the unique neighbour on that side | the side of the neighbour we arrive at << 31, or PT_NONE.
__global__ void pt_links_kernel(TableView t, const uint32_t* key_slot, const uint32_t* key_of_slot, uint64_t n, uint32_t* link)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    const uint64_t i = g >> 1;
    const uint32_t side = (uint32_t)g & 1u;
    const uint64_t K = t.slots8[key_slot[i]] & PT_MASK54;
    const uint64_t kmask = (1ULL << (2 * t.k)) - 1;      // (k = 27 for the context table's numbering; 19 .. 27 for the path table)
    uint32_t found = PT_NONE, cnt = 0, enter = 0;
    for (uint64_t b = 0; b < 4; ++b) {
        const uint64_t N = side ? ((K << 2) | b) & kmask : (K >> 2) | (b << (2 * t.k - 2));
        const uint64_t rc = vg_revcomp(N, t.k);
        const bool flipped = N > rc;
        const uint32_t s = pt_find(t, flipped ? rc : N);
        if (s == PT_NONE) continue;
        ++cnt;
        found = key_of_slot[s];
        enter = side ? (flipped ? 1u : 0u) : (flipped ? 0u : 1u);
    }
    // a k-mer that follows itself (homopolymers, K next to its own reverse complement) stays a chain end
    link[g] = (cnt == 1 && found != (uint32_t)i && found < 0x7FFFFFFFu) ? (found | enter << 31) : PT_NONE;
}

// keep a link only when the neighbour sees us the same way (unitigs)
__global__ void pt_mutual_kernel(const uint32_t* link, uint32_t* link2, uint64_t n)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    const uint32_t l = link[g];
    uint32_t out = PT_NONE;
    if (l != PT_NONE) {
        const uint64_t nb = l & 0x7FFFFFFFu, es = l >> 31;
        const uint32_t back = link[2 * nb + es];
        if (back != PT_NONE && (back & 0x7FFFFFFFu) == (uint32_t)(g >> 1) && (back >> 31) == ((uint32_t)g & 1u)) out = l;
    }
    link2[g] = out;
}

// one thread per (key, side) that is a chain end on that side: the end with the smaller key index lays the chain out.
// pos_of_key[i] = place in P's first half | (the walk reads the canonical k-mer as it stands) << 31
__global__ void pt_walk_kernel(const uint32_t* link2, uint64_t n, uint32_t* pos_of_key, unsigned long long* cursor, uint32_t align)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    if (link2[g] != PT_NONE) return;               // not an end on this side
    const uint32_t start = (uint32_t)(g >> 1), s0 = (uint32_t)g & 1u;
    uint32_t cur = start, out = s0 ^ 1u;
    uint64_t len = 1;
    for (;;) {
        const uint32_t l = link2[2ull * cur + out];
        if (l == PT_NONE || len > n) break;
        cur = l & 0x7FFFFFFFu;
        out = (l >> 31) ^ 1u;
        ++len;
    }
    const bool own = start < cur || (start == cur && (s0 == 0u || link2[2ull * start] != PT_NONE));
    if (!own || len > n) return;
    // (align > 1: every chain starts at a multiple of it -- the context table's counters, a chain's first 16 in one 64-byte sector)
    const uint64_t base = atomicAdd(cursor, (unsigned long long)((len + align - 1) / align * align));
    cur = start;
    out = s0 ^ 1u;
    for (uint64_t pos = 0; pos < len; ++pos) {
        pos_of_key[cur] = (uint32_t)(base + pos) | out << 31;     // leaving through the right side: walked in canonical orientation
        const uint32_t l = link2[2ull * cur + out];
        if (l == PT_NONE) break;
        cur = l & 0x7FFFFFFFu;
        out = (l >> 31) ^ 1u;
    }
}

__global__ void pt_rest_kernel(uint64_t n, uint32_t* pos_of_key, unsigned long long* cursor)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (pos_of_key[i] == PT_NONE) pos_of_key[i] = (uint32_t)atomicAdd(cursor, 1ULL) | 1u << 31;    // keys on cycles: chains of one
}

// every key a place of its own in [0, total)?  (total = n: a permutation.  status bit 16 otherwise: the caller then lays the keys
// out by key index, chains of one)
__global__ void pt_check_kernel(const uint32_t* pos_of_key, uint64_t n, uint64_t total, uint32_t* mark, uint32_t* status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pos = pos_of_key[i] & 0x7FFFFFFFu;
    if (pos >= total || atomicAdd(&mark[pos], 1u) != 0u) atomicOr(status, 16u);
}

__global__ void pt_fill_kernel(TableView t, const uint32_t* key_slot, const uint32_t* pos_of_key, uint64_t n, ulonglong2* P)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = key_slot[i];
    const uint64_t K = t.slots8[slot] & PT_MASK54, R = vg_revcomp(K, t.k);
    const uint32_t pk = pos_of_key ? pos_of_key[i] : ((uint32_t)i | 1u << 31);
    const uint64_t pos = pk & 0x7FFFFFFFu;
    const bool as_is = (pk >> 31) != 0;
    P[pos] = make_ulonglong2(as_is ? K : R, slot);
    P[2 * n - 1 - pos] = make_ulonglong2(as_is ? R : K, slot);
}

// mark == nullptr: no check here (a caller that aligns the chains checks with launch_ptable_check once it knows the total)
hipError_t launch_ptable_order(const TableView& t, const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot, uint32_t* link, uint32_t* link2,
                               uint32_t* pos_of_key, unsigned long long* cursor, uint32_t* mark, uint32_t* status, hipStream_t st, uint32_t align)
{
    if (n == 0) return hipSuccess;
    const uint32_t g2 = (uint32_t)((2 * n + 255) / 256), g1 = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(pt_key_of_slot_kernel, dim3(g1), dim3(256), 0, st, key_slot, n, key_of_slot);
    hipLaunchKernelGGL(pt_links_kernel, dim3(g2), dim3(256), 0, st, t, key_slot, key_of_slot, n, link);
    hipLaunchKernelGGL(pt_mutual_kernel, dim3(g2), dim3(256), 0, st, link, link2, n);
    hipLaunchKernelGGL(pt_walk_kernel, dim3(g2), dim3(256), 0, st, link2, n, pos_of_key, cursor, align ? align : 1u);
    hipLaunchKernelGGL(pt_rest_kernel, dim3(g1), dim3(256), 0, st, n, pos_of_key, cursor);
    if (mark) hipLaunchKernelGGL(pt_check_kernel, dim3(g1), dim3(256), 0, st, pos_of_key, n, n, mark, status);
    return hipGetLastError();
}

hipError_t launch_ptable_check(const uint32_t* pos_of_key, uint64_t n, uint64_t total, uint32_t* mark, uint32_t* status, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_check_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, pos_of_key, n, total, mark, status);
    return hipGetLastError();
}

hipError_t launch_ptable_fill(const TableView& t, const uint32_t* key_slot, const uint32_t* pos_of_key, uint64_t n, ulonglong2* P, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_fill_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, t, key_slot, pos_of_key, n, P);
    return hipGetLastError();
}

}  // namespace vgk
