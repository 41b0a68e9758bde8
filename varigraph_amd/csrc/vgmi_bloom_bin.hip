// vgmi_bloom_bin.hip -- K3, the counting-Bloom update, with the updates BINNED by filter chunk (round 4).
//
// Reference behaviour served: BloomFilter::add (src/counting_bloom_filter.cpp:28-36: n_hash MurmurHash3 positions per k-mer,
// each byte counter incremented up to 255) as ConstructIndex::make_mbf drives it (src/construct_index.cpp:258-335), and the
// CUDA twin's sort of the positions in front of its update (src/counting_bloom_filter.cu:68-106).
//
// Why.  The direct form (vgmi_kernels.hip: rows_kernel<MODE_BLOOM>) issues one 32-bit compare-and-swap per (k-mer, hash) into a
// filter far larger than the caches: 2.3e10 a second, which IS what the device does with atomics anywhere -- 27 G/s, in HBM or in
// an L2-resident slice alike (profiles/r2_ubench_mem3.json) -- for 15 bytes of algorithmic traffic per k-mer.  Binning by 2 MiB
// slices with the slice kept in L2 would still pay that rate.  So the counters are bumped in LDS:
//   1. the positions of all k-mers of a call (rows_kernel<MODE_KEYS> gives the k-mers, this file hashes them) are partitioned by
//      the 128 KiB CHUNK of the filter they fall in, in two levels (level 1: bins of B2 chunks, level 2: the chunks of a bin),
//      each level a tile sort in LDS -- histogram, exclusive scan, scatter -- with one reservation per (tile, bin) in the bin's
//      fixed room and whole runs written out;
//   2. one workgroup per chunk loads the chunk into LDS, applies its positions with LDS compare-and-swaps (a saturating byte
//      increment), and writes it back.
// Saturating increments commute, so the bytes are those of the reference whatever the order.  A bin's room is its expected load
// plus slack; positions are hashes, so only a call dominated by one repeated k-mer can overflow it -- the call is then redone
// the direct way (nothing has touched the filter before the overflow flag is read).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vgmi_device.h"
#include "vgmi_kernels.h"

namespace vgk {

#define BB_CH_LOG 17u                   // a chunk: 128 KiB of the filter
#define BB_MAX_BINS 512u                // bins of a level
#ifndef BB_TILE
#define BB_TILE 7168u                   // records a workgroup sorts at once
#endif
#define BB_KEYS (BB_TILE / 7u)           // level 1: keys per tile (n_hash <= 7)
#define BB_PER (BB_TILE / 256u)

__device__ __forceinline__ uint64_t bb_mod(uint64_t x, uint64_t m, uint64_t magic)
{
    const uint64_t q = __umul64hi(x, magic);
    uint64_t r = x - q * m;
    while (r >= m) r -= m;
    return r;
}

struct BbTile {
    uint32_t hist[BB_MAX_BINS], off[BB_MAX_BINS], gbase[BB_MAX_BINS];
    uint32_t scan[256];
    uint32_t rec[BB_TILE];
    uint16_t bin[BB_TILE];
};

// the tail every scatter shares: histogram in s.hist (complete), this thread's records in (rec[i], tag[i] = bin << 16 | rank, or
// ~0u), n_bins <= BB_MAX_BINS.  Reserves room in the bins, writes the runs out.
__device__ __forceinline__ void bb_tile_out(BbTile& s, const uint32_t (&rec)[BB_PER], const uint32_t (&tag)[BB_PER], uint32_t n_bins, uint32_t cap, uint32_t first_bin,
                                            uint32_t* __restrict__ cursor, uint32_t* __restrict__ out, uint32_t* __restrict__ overflow)
{
    const uint32_t t = threadIdx.x;
    // exclusive scan of hist[0 .. n_bins): two entries a thread, then a scan of the 256 pair sums
    const uint32_t a = 2u * t < n_bins ? s.hist[2u * t] : 0u, b = 2u * t + 1u < n_bins ? s.hist[2u * t + 1u] : 0u;
    // (inclusive scan inside the wavefront by DPP, the four wavefronts' totals through LDS: two barriers instead of eighteen)
    uint32_t incl = a + b;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);      // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);      // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);      // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);      // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
    if ((t & 63u) == 63u) s.scan[t >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (t >> 6); ++w) before += s.scan[w];
    const uint32_t total = s.scan[0] + s.scan[1] + s.scan[2] + s.scan[3];
    const uint32_t excl = before + incl - (a + b);
    if (2u * t < n_bins) s.off[2u * t] = excl;
    if (2u * t + 1u < n_bins) s.off[2u * t + 1u] = excl + a;
    // room in the bins
    for (uint32_t bi = t; bi < n_bins; bi += 256u) {
        const uint32_t h = s.hist[bi];
        uint32_t g = 0xFFFFFFFFu;
        if (h) {
            g = atomicAdd(&cursor[first_bin + bi], h);
            if (g + h > cap) {
                atomicExch(overflow, 1u);
                g = 0xFFFFFFFFu;
            }
        }
        s.gbase[bi] = g;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < BB_PER; ++i)
        if (tag[i] != 0xFFFFFFFFu) {
            const uint32_t bi = tag[i] >> 16, at = s.off[bi] + (tag[i] & 0xFFFFu);
            s.rec[at] = rec[i];
            s.bin[at] = (uint16_t)bi;
        }
    __syncthreads();
    for (uint32_t i = t; i < total; i += 256u) {
        const uint32_t bi = s.bin[i], g = s.gbase[bi];
        if (g != 0xFFFFFFFFu) out[(size_t)(first_bin + bi) * cap + g + (i - s.off[bi])] = s.rec[i];
    }
    __syncthreads();
}

// level 1: keys -> positions -> bins of (1 << bin_shift) filter bytes; a record is the position inside its bin
__global__ __launch_bounds__(256) void bb_scatter1_kernel(BloomView b, const uint64_t* __restrict__ keys, uint64_t n_keys, uint32_t bin_shift, uint32_t n_bins, uint32_t cap,
                                                          uint32_t* __restrict__ cursor, uint32_t* __restrict__ out, uint32_t* __restrict__ overflow)
{
    __shared__ BbTile s;
    const uint32_t t = threadIdx.x;
    const uint32_t keys_per_tile = BB_KEYS;                                 // four keys a thread, n_hash <= 7 positions each
    const uint64_t n_tiles = (n_keys + keys_per_tile - 1) / keys_per_tile;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        for (uint32_t i = t; i < n_bins; i += 256u) s.hist[i] = 0;
        __syncthreads();
        uint32_t rec[BB_PER], tag[BB_PER];
#pragma unroll
        for (uint32_t i = 0; i < BB_PER; ++i) tag[i] = 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t q = 0; q < BB_KEYS / 256u; ++q) {
            const uint64_t ki = tile * keys_per_tile + q * 256u + t;
            const uint64_t key = ki < n_keys ? keys[ki] : ~0ULL;
#pragma unroll
            for (uint32_t h = 0; h < 7u; ++h) {
                if (key != ~0ULL && h < b.n_hash) {
                    const uint64_t pos = bb_mod(vg_murmur_sum(key, b.seeds[h]), b.m, b.magic);
                    const uint32_t bi = (uint32_t)(pos >> bin_shift);
                    rec[q * 7u + h] = (uint32_t)(pos & ((1ull << bin_shift) - 1ull));
                    tag[q * 7u + h] = bi << 16 | atomicAdd(&s.hist[bi], 1u);
                }
            }
        }
        __syncthreads();
        bb_tile_out(s, rec, tag, n_bins, cap, 0u, cursor, out, overflow);
    }
}

// level 2: the records of level-1 bin blockIdx.y -> the chunks of that bin; a record is the position inside its chunk
__global__ __launch_bounds__(256) void bb_scatter2_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ cursor1, uint32_t cap1, uint32_t n_sub, uint32_t cap2,
                                                          uint32_t* __restrict__ cursor2, uint32_t* __restrict__ out, uint32_t* __restrict__ overflow)
{
    __shared__ BbTile s;
    const uint32_t t = threadIdx.x, b1 = blockIdx.y;
    const uint32_t n = cursor1[b1] < cap1 ? cursor1[b1] : cap1;
    const uint32_t* const src = in + (size_t)b1 * cap1;
    for (uint32_t start = blockIdx.x * BB_TILE; start < n; start += gridDim.x * BB_TILE) {
        for (uint32_t i = t; i < n_sub; i += 256u) s.hist[i] = 0;
        __syncthreads();
        uint32_t rec[BB_PER], tag[BB_PER];
#pragma unroll
        for (uint32_t i = 0; i < BB_PER; ++i) {
            const uint32_t at = start + i * 256u + t;
            tag[i] = 0xFFFFFFFFu;
            if (at < n) {
                const uint32_t r = src[at], bi = r >> BB_CH_LOG;
                rec[i] = r & ((1u << BB_CH_LOG) - 1u);
                tag[i] = bi << 16 | atomicAdd(&s.hist[bi], 1u);
            }
        }
        __syncthreads();
        bb_tile_out(s, rec, tag, n_sub, cap2, b1 * n_sub, cursor2, out, overflow);
    }
}

// a chunk of the filter through LDS: saturating byte increments at the chunk's recorded positions
__global__ __launch_bounds__(1024) void bb_accumulate_kernel(uint8_t* __restrict__ filter, uint64_t m_padded, const uint32_t* __restrict__ recs,
                                                             const uint32_t* __restrict__ cursor2, uint32_t cap2, uint32_t n_chunks)
{
    __shared__ uint32_t lds[1u << (BB_CH_LOG - 2u)];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t n = cursor2[c] < cap2 ? cursor2[c] : cap2;
        if (!n) continue;
        const uint64_t base = (uint64_t)c << BB_CH_LOG;
        const uint32_t n_words = (uint32_t)((m_padded - base < (1ull << BB_CH_LOG) ? m_padded - base : (1ull << BB_CH_LOG)) >> 2);
        uint32_t* const g = reinterpret_cast<uint32_t*>(filter + base);
        for (uint32_t w = threadIdx.x; w < n_words; w += 1024u) lds[w] = g[w];
        __syncthreads();
        const uint32_t* const r = recs + (size_t)c * cap2;
        for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
            const uint32_t p = r[i], w = p >> 2, sh = (p & 3u) * 8u;
            uint32_t old = lds[w];
            for (;;) {
                if (((old >> sh) & 0xFFu) == 0xFFu) break;
                const uint32_t prev = atomicCAS(&lds[w], old, old + (1u << sh));
                if (prev == old) break;
                old = prev;
            }
        }
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < n_words; w += 1024u) g[w] = lds[w];
        __syncthreads();
    }
}

// Geometry of a call: chunks, bins, room.  n_rec_max: positions at most (k-mer positions x n_hash).
BloomBinPlan bloom_bin_plan(uint64_t m, uint32_t n_hash, uint64_t n_keys_max)
{
    BloomBinPlan p{};
    const uint64_t m_padded = (m + 3) & ~3ULL;
    const uint64_t n_chunks = (m_padded + (1ull << BB_CH_LOG) - 1) >> BB_CH_LOG;
    if (n_hash < 1 || n_hash > 7 || n_chunks > (uint64_t)BB_MAX_BINS * BB_MAX_BINS) return p;
    uint32_t sub = 64;
    if (const char* e = getenv("VGMI_BLOOM_SUB")) sub = (uint32_t)atoi(e) >= 512 ? 512u : (uint32_t)atoi(e) >= 256 ? 256u : (uint32_t)atoi(e) >= 128 ? 128u : 64u;      // tests
    while ((n_chunks + sub - 1) / sub > BB_MAX_BINS) sub <<= 1;       // chunks of a level-1 bin: a power of two
    p.n_chunks = (uint32_t)n_chunks;
    p.n_sub = sub;
    p.n_bins = (uint32_t)((n_chunks + sub - 1) / sub);
    p.bin_shift = BB_CH_LOG + (uint32_t)__builtin_ctz(sub);
    const double n_rec = (double)n_keys_max * n_hash;
    // (the last bin / chunk may be partial and gets less; a full one gets its share of the positions)
    const double per_bin = n_rec * (double)((uint64_t)sub << BB_CH_LOG) / (double)m, per_chunk = n_rec * (double)(1ull << BB_CH_LOG) / (double)m;
    const double c1 = per_bin * 1.10 + 65536.0, c2 = per_chunk * 1.25 + 8192.0;
    if (c1 >= 4.0e9 || c2 >= 4.0e9) return p;
    p.cap1 = (uint32_t)c1;
    p.cap2 = (uint32_t)c2;
    p.scratch_bytes = ((size_t)p.n_bins * p.cap1 + (size_t)p.n_bins * p.n_sub * p.cap2) * 4 + ((size_t)p.n_bins + (size_t)p.n_bins * p.n_sub + 64) * 4;
    p.ok = 1;
    return p;
}

// scratch: plan.scratch_bytes.  *overflowed (host) is set when a bin ran out of room: nothing has been applied then.
hipError_t launch_bloom_binned(const BloomView& b, const uint64_t* keys, uint64_t n_keys, const BloomBinPlan& plan, uint8_t* scratch, int n_cu, hipStream_t st,
                               int* overflowed)
{
    uint32_t* const out1 = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* const out2 = out1 + (size_t)plan.n_bins * plan.cap1;
    uint32_t* const cur1 = out2 + (size_t)plan.n_bins * plan.n_sub * plan.cap2;
    uint32_t* const cur2 = cur1 + plan.n_bins;
    uint32_t* const flag = cur2 + (size_t)plan.n_bins * plan.n_sub;
    hipError_t e = hipMemsetAsync(cur1, 0, ((size_t)plan.n_bins + (size_t)plan.n_bins * plan.n_sub + 1) * 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bb_scatter1_kernel, dim3((uint32_t)n_cu * 3u), dim3(256), 0, st, b, keys, n_keys, plan.bin_shift, plan.n_bins, plan.cap1, cur1, out1, flag);
    const uint32_t tiles = (plan.cap1 + BB_TILE - 1) / BB_TILE;
    hipLaunchKernelGGL(bb_scatter2_kernel, dim3(tiles < 64u ? tiles : 64u, plan.n_bins), dim3(256), 0, st, out1, cur1, plan.cap1, plan.n_sub, plan.cap2, cur2, out2, flag);
    uint32_t h_flag = 0;
    e = hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    *overflowed = h_flag != 0;
    if (h_flag) return hipSuccess;
    hipLaunchKernelGGL(bb_accumulate_kernel, dim3((uint32_t)n_cu), dim3(1024), 0, st, b.filter, (b.m + 3) & ~3ULL, out2, cur2, plan.cap2, plan.n_chunks);
    return hipGetLastError();
}

}  // namespace vgk
