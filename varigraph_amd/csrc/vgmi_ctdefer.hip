// vgmi_ctdefer.hip -- the context-table count kernels' counter updates, DEFERRED (round 6; VERDICT r5 "next" #1b).
//
// Reference behaviour served: src/fastq_kmer.cpp:128-139 -- a graph k-mer's coverage is c = min(255, number of times the reads hold it).
// A sum does not depend on the order of its terms, so the increments of a launch may be applied behind its row loop.
//
// Why.  count27c_kernel (vgmi_ctable.hip) is bound by memory-side requests, and the 2.9 atomic requests a read makes -- every one of
// them leaves the XCD's L2 for the fabric (TCC_EA0_WRREQ == TCP_TCC_ATOMIC requests in profiles/r6_*: agent-scope atomics are not
// executed in a per-XCD L2) -- cost 1.8 of its ~8.4 ms at chr20 class (VGMI_DBG=2 ablation).  With DEFER the row loop writes its runs
// of hits {id0, windows | dir << 12} out as 8-byte records, 64 per coalesced store; two small kernels behind it turn them into counts:
//   1. ctd_scatter_kernel: a tile sort in LDS (histogram by LDS atomics whose return value is the record's rank in its bin, exclusive
//      scan, records placed by bin, whole runs written out -- the machinery of vgmi_bloom_bin.hip) partitions the records by REGION
//      of up to 32 768 consecutive counters; every workgroup has a room of its own in every region, so nothing is reserved in global
//      memory.  A record leaves as 27 bits: the lowest counter it touches, inside its region, and the up to twelve counters from there on
//      as a bit mask (a run's counters are neighbours either way: id0 - s or id0 + s, vgmi_ctable.h).
//   2. ctd_accumulate_kernel: a workgroup per region adds its records up in LDS (twelve ds_add_u32 at constant offsets a record, no
//      return value) and hands the non-zero sums to the counters with one atomic per counter and launch -- coalesced, 1.9e6 atomic
//      requests a chr20-class launch where the row loop made 7.0e7.
// Exact under every load: a record whose room is full (a sample whose reads pile onto one region: a room is a workgroup's share of a
// FULL record buffer plus slack), a counter beyond its record's region and a run that found the record buffer full are counted by
// plain atomics where they are met.  Tables of more than CTD_MAX_BINS regions (6.7e7 counters: whole-genome class, where a read makes
// ~1 run and the atomics are a tenth of the kernel's requests) and blocks below VGMI_CT_DEFER_MIN (512 MiB: the pieces of a FASTQ
// stream do not earn the second pass's fixed cost back) keep the plain kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vgmi_kernels.h"

namespace vgk {

#define CTD_TILE 16384u                 // records a workgroup sorts at once: the longer a bin's run in a tile, the fuller the lines it writes
#define CTD_WG 1024u                    // (8 192 records and two workgroups of 512 a CU: 0.40 against 0.36 ms at chr20 class)
#define CTD_PER (CTD_TILE / CTD_WG)
#define CTD_EPT (CTD_MAX_BINS / CTD_WG) // histogram entries a thread scans
#define CTD_REGION_MAX 32768u

struct CtdTile {
    uint32_t hist[CTD_MAX_BINS];
    uint16_t off[CTD_MAX_BINS];
    uint32_t gbase[CTD_MAX_BINS];
    uint32_t mine[CTD_MAX_BINS];      // records this workgroup has put into its room of each bin so far
    uint32_t scan[CTD_WG / 64u];
    uint32_t rec[CTD_TILE];
    uint16_t bin[CTD_TILE];
};

__device__ __forceinline__ void ctd_count_direct(uint32_t* counts, uint32_t lo, uint32_t m)
{
    while (m) {
        const uint32_t j = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        __hip_atomic_fetch_add(counts + lo + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// region of a counter: lo / d.region (d.inv = 2^32 / region rounded up: the quotient is at most one too large)
__device__ __forceinline__ uint32_t ctd_region_of(const CtDefer& d, uint32_t lo)
{
    uint32_t bi = __umulhi(lo, d.inv);
    if (bi * d.region > lo) --bi;
    return bi;
}

// Every workgroup has a room of its own in every bin (d.room records at (bin * gridDim.x + workgroup) * d.room), so placing a tile's
// records needs no reservation in global memory.
__global__ __launch_bounds__(CTD_WG) void ctd_scatter_kernel(XTableView xt, CtDefer d)
{
    __shared__ CtdTile s;
    const uint32_t t = threadIdx.x;
    const uint32_t cur = *d.cursor;
    const uint32_t n = cur < d.cap ? cur : d.cap;      // chunks are reserved whole and a chunk that would pass cap is not written: everything below min(cursor, cap) is
    const uint32_t n_tiles = (n + CTD_TILE - 1u) / CTD_TILE;
    for (uint32_t i = t; i < d.n_bins; i += CTD_WG) s.mine[i] = 0;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (uint32_t i = t; i < CTD_MAX_BINS; i += CTD_WG) s.hist[i] = 0;
        __syncthreads();
        uint32_t rec[CTD_PER], tag[CTD_PER];
#pragma unroll
        for (uint32_t i = 0; i < CTD_PER; ++i) {
            const uint32_t at = tile * CTD_TILE + i * CTD_WG + t;
            tag[i] = 0xFFFFFFFFu;
            rec[i] = 0;
            if (at < n) {
                const uint2 r = d.rec[at];
                const uint32_t m = r.y & 0xFFFu;
                if (m) {
                    // the run's counters in ascending order: lo and the mask from lo on (bit 0 set)
                    uint32_t lo, up;
                    if (r.y & 0x1000u) {
                        const uint32_t z = (uint32_t)__builtin_ctz(m);
                        lo = r.x + z;
                        up = m >> z;
                    } else {
                        const uint32_t h = 31u - (uint32_t)__builtin_clz(m);
                        lo = r.x - h;
                        up = __builtin_bitreverse32(m) >> (31u - h);
                    }
                    const uint32_t bi = ctd_region_of(d, lo);
                    if (bi < d.n_bins) {
                        rec[i] = (lo - bi * d.region) | up << 15;
                        tag[i] = bi << 16 | atomicAdd(&s.hist[bi], 1u);
                    }
                }
            }
        }
        __syncthreads();
        // exclusive scan of hist: CTD_EPT entries a thread (inclusive scan of the threads' sums inside the wavefront by DPP, the wavefronts' totals through LDS)
        uint32_t h[CTD_EPT], sum = 0;
#pragma unroll
        for (uint32_t e = 0; e < CTD_EPT; ++e) {
            h[e] = s.hist[CTD_EPT * t + e];
            sum += h[e];
        }
        uint32_t incl = sum;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);      // row_shr:1
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);      // row_shr:2
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);      // row_shr:4
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);      // row_shr:8
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
        if ((t & 63u) == 63u) s.scan[t >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < CTD_WG / 64u; ++w) {
            const uint32_t v = s.scan[w];
            if (w < (t >> 6)) before += v;
            total += v;
        }
        uint32_t excl = before + incl - sum;
        // ... and where the tile's records of a bin go in this workgroup's room of the bin (the thread that scans a bin owns it: no atomics)
#pragma unroll
        for (uint32_t e = 0; e < CTD_EPT; ++e) {
            const uint32_t bi = CTD_EPT * t + e;
            s.off[bi] = (uint16_t)excl;
            excl += h[e];
            const uint32_t g = s.mine[bi];
            s.gbase[bi] = g;
            s.mine[bi] = g + h[e];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < CTD_PER; ++i)
            if (tag[i] != 0xFFFFFFFFu) {
                const uint32_t bi = tag[i] >> 16, at = s.off[bi] + (tag[i] & 0xFFFFu);
                s.rec[at] = rec[i];
                s.bin[at] = (uint16_t)bi;
            }
        __syncthreads();
        for (uint32_t i = t; i < total; i += CTD_WG) {
            const uint32_t bi = s.bin[i], r = s.rec[i];
            const uint32_t slot = s.gbase[bi] + (i - s.off[bi]);
            if (slot < d.room) d.binned[((size_t)bi * gridDim.x + blockIdx.x) * d.room + slot] = r;
            else ctd_count_direct(xt.counts, bi * d.region + (r & 0x7FFFu), r >> 15);      // no room: counted here
        }
        __syncthreads();
    }
    __syncthreads();
    for (uint32_t bi = t; bi < d.n_bins; bi += CTD_WG) d.bin_cursor[(size_t)bi * gridDim.x + blockIdx.x] = s.mine[bi];
}

// A workgroup per region: the region's records -- n_wg rooms, a wavefront two neighbouring rooms at a time -- added up in LDS, then the
// sums handed to the counters.  A record's twelve possible counters take twelve ds_add_u32 at constant offsets from one address, each
// adding the record's bit (0 or 1): no branch, no loop over set bits (that loop, 10 instructions a bit, was 0.54 ms at chr20 class).
// The twelve words behind the region's last counter take what runs reach into the next region.
__global__ __launch_bounds__(1024) void ctd_accumulate_kernel(XTableView xt, CtDefer d, uint32_t n_wg)
{
    __shared__ uint32_t lds[CTD_REGION_MAX + 16u];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    for (uint32_t b = blockIdx.x; b < d.n_bins; b += gridDim.x) {
        for (uint32_t w = t; w < d.region + 16u; w += 1024u) lds[w] = 0;
        __syncthreads();
        for (uint32_t g = 2u * wave; g < n_wg; g += 32u) {
            const size_t room = (size_t)b * n_wg + g;
            const uint32_t ca = d.bin_cursor[room], cb = g + 1u < n_wg ? d.bin_cursor[room + 1] : 0u;
            const uint32_t na = ca < d.room ? ca : d.room, nb = cb < d.room ? cb : d.room;
            const uint32_t* const ra = d.binned + room * d.room;
            for (uint32_t i = lane; i < na + nb; i += 64u) {
                const uint32_t v = i < na ? ra[i] : ra[d.room + (i - na)], m = v >> 15;
                uint32_t* const at = &lds[v & 0x7FFFu];
#pragma unroll
                for (uint32_t j = 0; j < 12u; ++j) (void)__hip_atomic_fetch_add(at + j, (m >> j) & 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        const uint32_t base = b * d.region;
        for (uint32_t w = t; w < d.region + 16u; w += 1024u) {
            const uint32_t v = lds[w];
            if (v && (uint64_t)base + w < d.n_counts) __hip_atomic_fetch_add(xt.counts + base + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

// Geometry for a block of n_bytes over a table of n_counts counters.  Returns the scratch a stream needs (0: not served) and fills
// everything of *d but the pointers.
size_t ctd_scratch_bytes(uint64_t n_bytes, uint64_t n_counts, uint32_t n_cu, CtDefer* d)
{
    if (n_counts == 0 || n_counts > (uint64_t)CTD_MAX_BINS * CTD_REGION_MAX || n_cu == 0) return 0;
    // regions of equal size, as many as keep every CU of the accumulate kernel equally busy: a multiple of n_cu (1 040 regions of 32 768
    // counters over 256 CUs are five rounds for sixteen CUs and four for the rest)
    uint64_t per_cu = (n_counts + (uint64_t)n_cu * CTD_REGION_MAX - 1) / ((uint64_t)n_cu * CTD_REGION_MAX);
    uint64_t n_bins = per_cu * n_cu;
    if (n_bins > CTD_MAX_BINS) n_bins = (n_counts + CTD_REGION_MAX - 1) / CTD_REGION_MAX;
    uint64_t region = (n_counts + n_bins - 1) / n_bins;
    if (region < 256) region = 256;      // (tables of a few thousand counters: fewer regions than CUs)
    n_bins = (n_counts + region - 1) / region;
    const uint32_t n_wg = n_cu & ~1u;      // workgroups of the scatter kernel (even: the accumulate kernel takes rooms in pairs)
    // a read of 150 bases makes ~3.4 runs at chr20 class: room for one run per 16 bytes of text (9.4 a read) + a chunk per wavefront of the largest grid
    uint64_t cap = n_bytes / 16 + 8192ull * CTD_CHUNK;
    cap = (cap + CTD_CHUNK - 1) / CTD_CHUNK * CTD_CHUNK;
    if (cap > 0xC0000000ull) cap = 0xC0000000ull;      // the cursor is 32 bits (the runs beyond leave as atomics)
    if (const char* e = getenv("VGMI_CT_DEFER_CAP")) cap = ((uint64_t)atoll(e) + CTD_CHUNK - 1) / CTD_CHUNK * CTD_CHUNK + CTD_CHUNK;      // tests: a buffer that fills up
    // a workgroup's room in a bin: its share of a FULL record buffer (2.7 x the mean at chr20 class) + slack for the spread of small means
    uint64_t room = cap / (n_bins * n_wg) + 64;
    if (const char* e = getenv("VGMI_CT_DEFER_ROOM")) room = (uint64_t)atoll(e) + 1;                                                     // tests: rooms that fill up
    d->cap = (uint32_t)cap;
    d->n_bins = (uint32_t)n_bins;
    d->room = (uint32_t)room;
    d->n_counts = n_counts;
    d->region = (uint32_t)region;
    d->inv = (uint32_t)(((1ull << 32) + region - 1) / region);
    d->n_wg = n_wg;
    return (size_t)(256 + (((n_bins * n_wg + 1) * 4 + 255) & ~255ull) + cap * 8 + n_bins * n_wg * room * 4);
}

// scratch: [cursor | records the workgroups put into their rooms | records | rooms]
void ctd_layout(uint8_t* scratch, CtDefer* d)
{
    const uint32_t n_wg = d->n_wg;
    d->cursor = reinterpret_cast<unsigned int*>(scratch);
    d->bin_cursor = d->cursor + 1;
    uint8_t* const recs = scratch + 256 + ((((size_t)d->n_bins * n_wg + 1) * 4 + 255) & ~(size_t)255);
    d->rec = reinterpret_cast<uint2*>(recs);
    d->binned = reinterpret_cast<uint32_t*>(recs + (size_t)d->cap * 8);
}

// in front of the count kernel: the cursor back to zero (the rooms' counts are written whole by the scatter kernel)
hipError_t launch_ctd_reset(const CtDefer& d, hipStream_t st) { return hipMemsetAsync(d.cursor, 0, 4, st); }

// behind it: the records into the counters
hipError_t launch_ctd_apply(const XTableView& t, const CtDefer& d, uint32_t n_cu, hipStream_t st)
{
    hipLaunchKernelGGL(ctd_scatter_kernel, dim3(d.n_wg), dim3(CTD_WG), 0, st, t, d);
    hipLaunchKernelGGL(ctd_accumulate_kernel, dim3(n_cu < d.n_bins ? n_cu : d.n_bins), dim3(1024), 0, st, t, d, d.n_wg);
    return hipGetLastError();
}

}  // namespace vgk
