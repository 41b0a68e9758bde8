// vgmi_ptable.hip -- the PATH TABLE of small graphs (k = 27, <= 65 536 k-mers: BASELINE config 2, the bench): what
// count27s_kernel (vgmi_kernels.hip) looks candidate runs up in.
//
// Same reference behaviour as every count kernel here: src/kmer.cpp:140-142 (exact membership of the canonical k-mer),
// src/fastq_kmer.cpp:128-139 (saturating count).
//
// Why.  A candidate run is 16 consecutive k-mers of a read (the windows around one grid 12-mer X, vgmi_device.h).  Looked up
// one by one in the exact hash table they are 16 random 8-byte probes; 5.4e7 runs per 1e8-read sample make 8.6e8 probes,
// which is all an XCD's L2 delivers in 3.3 ms (tools/ubench_mem: 263 G random requests/s below 4 MiB) -- measured: the scan
// alone takes 2.7 ms of the kernel's 6.0 (VGMI_DBG=1).  But the 16 k-mers are not random: they are CONSECUTIVE k-mers of a
// haplotype path, and so are the graph's k-mers they may equal.  So:
//     P      the graph's k-mers laid out along their unitigs (chains of k-mers that follow each other uniquely in the key set:
//            the 27 k-mers across one allele of a SNP are one chain), each chain once in every orientation: entry i + 1 is entry
//            i shifted by one base.  16 bytes per entry: {k-mer word | saturation flag, table slot}.  2 n entries, P[2n - 1 - i]
//            is the reverse complement of P[i].
//     index  canonical 12-mer cx -> up to two places p0 in P such that "the k-mer that has cx w bases before its end" is
//            P[p0 + w] (for every w at which such a k-mer exists); the two alleles of a site give two places.  2^17 buckets of
//            two 8-byte entries {cx : 24, p0a : 19, p0b : 19}, exact compare on cx.
// A run costs one 16-byte index load (one request for its 16 lanes) and one or two loads of 16 CONSECUTIVE entries of P (one or
// two lines each): 3-5 requests instead of 16, all of them in 0.9 MiB + 2 MiB that stay L2-resident.  The compare is on the
// whole k-mer, so a hit is exact; every (12-mer, k-mer) incidence of the graph is in the index unless it is marked as
// overflowed (a third place for one 12-mer: neighbouring sites, repeats; a third 12-mer in one bucket) -- runs that meet a
// mark take the exact hash table as before.  Counters stay per table slot, so read-out, the generic kernels (ragged tails,
// k != 27) and the image format do not change; the path table is derived from the image after upload / import / clone.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_device.h"
#include "vgmi_kernels.h"

namespace vgk {

#define PT_NONE 0xFFFFFFFFu
#define PT_MASK54 ((1ULL << 54) - 1)

// slot of a canonical k-mer in the compact table, or PT_NONE
__device__ __forceinline__ uint32_t pt_find(const TableView& t, uint64_t canon)
{
    uint64_t s = vg_thash(canon) & t.cap_mask;
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
    uint32_t found = PT_NONE, cnt = 0, enter = 0;
    for (uint64_t b = 0; b < 4; ++b) {
        const uint64_t N = side ? ((K << 2) | b) & PT_MASK54 : (K >> 2) | (b << 52);
        const uint64_t rc = vg_revcomp(N, 27);
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
__global__ void pt_walk_kernel(const uint32_t* link2, uint64_t n, uint32_t* pos_of_key, unsigned long long* cursor)
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
    const uint64_t base = atomicAdd(cursor, (unsigned long long)len);
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

// every place in [0, n) exactly once?  (status bit 16 otherwise: the caller then lays the keys out by key index, chains of one)
__global__ void pt_check_kernel(const uint32_t* pos_of_key, uint64_t n, uint32_t* mark, uint32_t* status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pos = pos_of_key[i] & 0x7FFFFFFFu;
    if (pos >= n || atomicAdd(&mark[pos], 1u) != 0u) atomicOr(status, 16u);
}

__global__ void pt_fill_kernel(TableView t, const uint32_t* key_slot, const uint32_t* pos_of_key, uint64_t n, ulonglong2* P)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = key_slot[i];
    const uint64_t K = t.slots8[slot] & PT_MASK54, R = vg_revcomp(K, 27);
    const uint32_t pk = pos_of_key ? pos_of_key[i] : ((uint32_t)i | 1u << 31);
    const uint64_t pos = pk & 0x7FFFFFFFu;
    const bool as_is = (pk >> 31) != 0;
    P[pos] = make_ulonglong2(as_is ? K : R, slot);
    P[2 * n - 1 - pos] = make_ulonglong2(as_is ? R : K, slot);
}

// index entry: cx | (p0a + 1 biased place, 0: the entry is empty) << 24 | (p0b, 0: none, PT_OVF: more than two places) << 43;
// bit 62 of a bucket's FIRST entry: some 12-mer of this bucket found no entry (lookups that miss then take the hash table)
#define PT_P0_MASK 0x7FFFFu
#define PT_OVF 0x7FFFFu
#define PT_BUCKET_OVF (1ULL << 62)
#define PT_BIAS 16u

__global__ void pt_index_kernel(const ulonglong2* P, uint64_t n2, unsigned long long* index, uint32_t bucket_log2)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n2 * 16) return;
    const uint64_t idx = g >> 4;
    const uint32_t o = (uint32_t)g & 15u;                     // X = bases o .. o + 11 of the k-mer
    const uint64_t kmer = P[idx].x & PT_MASK54;
    const uint32_t x = (uint32_t)(kmer >> (2 * (15 - o))) & 0xFFFFFFu;
    if (x > vg_revcomp12(x)) return;                          // this incidence is listed from the other orientation's entry
    const uint32_t w = 15u - o;                               // X lies w bases before the k-mer's end
    const uint64_t p0 = idx + PT_BIAS - w;                    // >= 1: P[p0 - PT_BIAS + w] is this k-mer
    unsigned long long* B = index + ((uint64_t)(vg_mul24(x, 0x9E3779u) >> (32 - bucket_log2)) << 1);
    for (int e = 0; e < 2; ++e) {
        for (;;) {
            const unsigned long long raw = *reinterpret_cast<volatile unsigned long long*>(&B[e]);
            const unsigned long long flag = raw & PT_BUCKET_OVF, cur = raw & ~PT_BUCKET_OVF;
            const uint32_t pa = (uint32_t)(cur >> 24) & PT_P0_MASK;
            if (pa == 0) {                                     // empty: claim it
                if (atomicCAS(&B[e], raw, (unsigned long long)x | p0 << 24 | flag) == raw) return;
                continue;                                      // somebody else wrote here first: look again
            }
            if (((uint32_t)cur & 0xFFFFFFu) != x) break;       // another 12-mer lives here: next entry
            const uint32_t pb = (uint32_t)(cur >> 43) & PT_P0_MASK;
            if (pa == p0 || pb == p0 || pb == PT_OVF) return;  // listed (the same place from a neighbouring k-mer), or given up on
            const unsigned long long want = (cur & ~((unsigned long long)PT_P0_MASK << 43)) | (unsigned long long)(pb == 0 ? p0 : PT_OVF) << 43 | flag;
            if (atomicCAS(&B[e], raw, want) == raw) return;
        }
    }
    atomicOr(&B[0], PT_BUCKET_OVF);                            // both entries belong to other 12-mers
}

hipError_t launch_ptable_order(const TableView& t, const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot, uint32_t* link, uint32_t* link2,
                               uint32_t* pos_of_key, unsigned long long* cursor, uint32_t* mark, uint32_t* status, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint32_t g2 = (uint32_t)((2 * n + 255) / 256), g1 = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(pt_key_of_slot_kernel, dim3(g1), dim3(256), 0, st, key_slot, n, key_of_slot);
    hipLaunchKernelGGL(pt_links_kernel, dim3(g2), dim3(256), 0, st, t, key_slot, key_of_slot, n, link);
    hipLaunchKernelGGL(pt_mutual_kernel, dim3(g2), dim3(256), 0, st, link, link2, n);
    hipLaunchKernelGGL(pt_walk_kernel, dim3(g2), dim3(256), 0, st, link2, n, pos_of_key, cursor);
    hipLaunchKernelGGL(pt_rest_kernel, dim3(g1), dim3(256), 0, st, n, pos_of_key, cursor);
    hipLaunchKernelGGL(pt_check_kernel, dim3(g1), dim3(256), 0, st, pos_of_key, n, mark, status);
    return hipGetLastError();
}

hipError_t launch_ptable_fill(const TableView& t, const uint32_t* key_slot, const uint32_t* pos_of_key, uint64_t n, ulonglong2* P,
                              unsigned long long* index, uint32_t bucket_log2, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_fill_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, t, key_slot, pos_of_key, n, P);
    const uint64_t m = 2 * n * 16;
    hipLaunchKernelGGL(pt_index_kernel, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, st, P, 2 * n, index, bucket_log2);
    return hipGetLastError();
}

// per-sample reset: the saturation flags of P
__global__ void pt_reset_kernel(ulonglong2* P, uint64_t n2)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) P[i].x &= ~VG_SLOT_SAT;
}

hipError_t launch_ptable_reset(ulonglong2* P, uint64_t n2, hipStream_t st)
{
    if (n2 == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_reset_kernel, dim3((uint32_t)((n2 + 255) / 256)), dim3(256), 0, st, P, n2);
    return hipGetLastError();
}

}  // namespace vgk
