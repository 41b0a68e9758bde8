// vgmi_kernels.hip -- hand-written gfx950 kernels of the varigraph genotyping hot path.
//
// Reference behaviour implemented (file:line under the reference tree):
//   K1  rolling canonical k-mer emitter            src/kmer.cpp:110-149 (and :20-53 for the Bloom driver)
//   K2  exact table membership + saturating count  src/kmer.cpp:140-142, src/fastq_kmer.cpp:128-139
//   K3  counting-Bloom update                      src/counting_bloom_filter.cpp:28-36,90-98
//   K4  counting-Bloom count/find                  src/counting_bloom_filter.cpp:40-67
//   K5  clamp + per-node depth gather              src/genotype.cpp:546,660,1405-1408 (via node iterators)
//   K6  masked coverage histogram                  src/varigraph.cpp:253-296
//
// Design (DESIGN.md has the full story).  The read block is a '\n'-joined ASCII byte stream.
// A wavefront walks contiguous rows of it; a lane owns a fixed chunk of the row (one coalesced
// load per lane), encodes it to 2 bits/base through an LDS-resident copy of seq_nt4_table, and
// receives the bases before its chunk from the lanes before it (previous row for the first
// lanes) by cross-lane permutes -- no per-read state, no halo recomputation.  For odd k a k-mer
// can never equal its own reverse complement, so the reference's state machine reduces exactly
// to "emit iff the last k bases are all valid" (even k takes the sequential
// kernel below, which restates the machine literally).
//   rows_kernel     generic odd k: 1 KiB rows, 16 bytes per lane; every canonical k-mer is tested
//                   against a blocked Bloom prefilter (LDS when the graph is small enough),
//                   survivors are compacted per wavefront into an LDS queue and probed 64 at a
//                   time against the exact open-addressing table.
//   count27_kernel  k = 27 (every BASELINE configuration): 768-byte rows, 12 bytes per lane, ONE
//                   grid-filter probe per lane and row, candidate RUNS of 12 k-mers, pipelined
//                   table probes with hand-scheduled vector memory.
// Hits bump 32-bit counters with atomicAdd (skipped once a counter has reached the 255 clamp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_device.h"
#include "vgmi_kernels.h"
#include "vgmi_xtable.h"

namespace vgk {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
    // (hi:lo) >> sh, low 32 bits; sh in [0,31]  (v_alignbit_b32)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// reverse the order of the sixteen 2-bit fields of x and complement them
__device__ __forceinline__ uint32_t rc_word(uint32_t x)
{
    uint32_t r = __builtin_bitreverse32(x);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return ~r;
}

__device__ __forceinline__ uint64_t mod_u64(uint64_t x, uint64_t m, uint64_t magic)
{
    uint64_t q = __umul64hi(x, magic);
    uint64_t r = x - q * m;
    while (r >= m) r -= m;
    return r;
}

// saturating byte increment: filter[pos] = min(255, filter[pos] + 1)
// (src/counting_bloom_filter.cpp:32-34; the reference CUDA twin does the same CAS, counting_bloom_filter.cu:5-17)
__device__ __forceinline__ void bloom_inc(uint8_t* filter, uint64_t pos)
{
    uint32_t* w = reinterpret_cast<uint32_t*>(filter + (pos & ~3ULL));
    const uint32_t sh = (uint32_t)(pos & 3) * 8;
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (((old >> sh) & 0xFFu) == 0xFFu) return;
        const uint32_t prev = atomicCAS(w, old, old + (1u << sh));
        if (prev == old) return;
        old = prev;
    }
}

__device__ __forceinline__ void bloom_add_key(const BloomView& b, uint64_t key)
{
    for (uint32_t i = 0; i < b.n_hash; ++i) {
        const uint64_t pos = mod_u64(vg_murmur_sum(key, b.seeds[i]), b.m, b.magic);
        bloom_inc(b.filter, pos);
    }
}

// n occurrences of one key at once: filter[pos] = min(255, filter[pos] + n) per hash (n increments of the reference)
__device__ __forceinline__ void bloom_add_key_n(const BloomView& b, uint64_t key, uint32_t n)
{
    for (uint32_t i = 0; i < b.n_hash; ++i) {
        const uint64_t pos = mod_u64(vg_murmur_sum(key, b.seeds[i]), b.m, b.magic);
        uint32_t* w = reinterpret_cast<uint32_t*>(b.filter + (pos & ~3ULL));
        const uint32_t sh = (uint32_t)(pos & 3) * 8;
        uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const uint32_t v = (old >> sh) & 0xFFu;
            if (v == 0xFFu) break;
            const uint32_t nv = v + n > 255u ? 255u : v + n;
            const uint32_t prev = atomicCAS(w, old, (old & ~(0xFFu << sh)) | (nv << sh));
            if (prev == old) break;
            old = prev;
        }
    }
}

// Per-wavefront privatisation of the Bloom update (K3): the k-mers the 64 lanes present in one step go through a
// 128-cell table in LDS first, so a key that several lanes hold -- tandem repeats, homopolymers, satellites put the same
// k-mer at many of a wave's 1 024 consecutive positions -- reaches the filter once, with its multiplicity, instead of
// as up to 64 compare-and-swap loops fighting over the same seven bytes.
#define VG_BPRIV_CELLS 128u
__device__ __forceinline__ void bloom_add_wave(const BloomView& b, unsigned long long* cell_key, uint32_t* cell_cnt, bool active,
                                               uint64_t key)
{
    uint32_t slot = 0;
    bool owner = false;
    if (active) {
        slot = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> 57);
        for (;;) {
            const unsigned long long old = atomicCAS(&cell_key[slot], ~0ULL, (unsigned long long)key);
            if (old == ~0ULL) { owner = true; break; }
            if (old == key) break;
            slot = (slot + 1) & (VG_BPRIV_CELLS - 1);
        }
        atomicAdd(&cell_cnt[slot], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (owner) {
        const uint32_t n = cell_cnt[slot];
        cell_cnt[slot] = 0;
        cell_key[slot] = ~0ULL;
        bloom_add_key_n(b, key, n);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// counter cell of the key held by slot s (16-byte format: v = the slot; compact format: counts are per slot)
__device__ __forceinline__ uint32_t* count_cell(const TableView& t, uint64_t s, uint32_t key_index)
{
    if (t.slots8) return &t.counts[s];
    return t.counts ? &t.counts[key_index] : &t.slots[s].count;
}

// exact-table probe + saturating count of one canonical k-mer (generic kernels; either table format)
__device__ __forceinline__ uint64_t table_home(const TableView& t, uint64_t canon)
{
    if (t.home_bucket_log2) return vg_thash_local(canon, vg_revcomp(canon, 27), t.home_bucket_log2, t.home_by_offset != 0) & t.cap_mask;
    return vg_thash(canon) & t.cap_mask;
}

__device__ __forceinline__ void table_count(const TableView& t, uint64_t canon)
{
    if (t.xt.cb) {      // context table: the k-mer as a one-window context
        ct_count(t.xt, canon);
        return;
    }
    if (t.xt.lines) {   // grid-16-mer table: the k-mer's own first 16-mer (offset 0) is as good as any of the twelve
        xt_count(t.xt, canon, 0);
        return;
    }
    uint64_t s = table_home(t, canon);
    if (t.slots8) {   // compact format: k-mer words, per-slot counters
        for (;;) {
            const uint64_t c = t.slots8[s];
            if (c == VG_EMPTY) return;
            if ((c & VG_SLOT_KMER_MASK) == canon) {
                if (!(c & VG_SLOT_SAT) && atomicAdd(&t.counts[s], 1u) == 254u) {
                    // the increment that takes the counter to the clamp flags the k-mer as the fast kernels do: in the slot,
                    // for the per-sample reset, and -- small graphs -- at both of its places in the path table
                    atomicOr(reinterpret_cast<unsigned int*>(&t.slots8[s]) + 1, (unsigned int)(VG_SLOT_SAT >> 32));
                    if (t.sat_dirty) t.sat_dirty[s >> 11] = 1;      // (VG_SAT_REGION_LOG2)
                    if (t.pt.PLACE) {
                        const uint32_t pos = t.pt.PLACE[s];
                        if (pos != 0u) {
                            const uint32_t mir = t.pt.Tp - t.k - pos;
                            atomicOr(&t.pt.SB[pos >> 5], 1u << (pos & 31u));
                            atomicOr(&t.pt.SB[mir >> 5], 1u << (mir & 31u));
                        }
                    }
                }
                return;
            }
            if (!(c & VG_SLOT_CHAIN)) return;
            s = (s + 1) & t.cap_mask;
        }
    }
    for (;;) {
        const uint4 v = *reinterpret_cast<const uint4*>(&t.slots[s]);
        const uint64_t c = ((uint64_t)v.y << 32) | v.x;
        if (c == VG_EMPTY) return;
        if ((c & VG_SLOT_KMER_MASK) == canon) {
            if (t.counts) atomicAdd(&t.counts[v.w], 1u);
            else if (v.z < 255u) atomicAdd(&t.slots[s].count, 1u);
            return;
        }
        if (!(c & VG_SLOT_CHAIN)) return;
        s = (s + 1) & t.cap_mask;
    }
}

// the count of one canonical k-mer taken back by one (even k on the fast paths: even_debit_kernel)
__device__ __forceinline__ void table_debit(const TableView& t, uint64_t canon)
{
    if (t.xt.cb) {      // large graphs of even k (round 5): the counter of the context table's id
        const uint32_t id = ct_find(t.xt, canon);
        if (id != 0xFFFFFFFFu) atomicSub(t.xt.counts + id, 1u);
        return;
    }
    if (!t.slots8) return;
    uint64_t s = table_home(t, canon);
    for (;;) {
        const uint64_t c = t.slots8[s];
        if (c == VG_EMPTY) return;
        if ((c & VG_SLOT_KMER_MASK) == canon) {
            atomicSub(&t.counts[s], 1u);
            return;
        }
        if (!(c & VG_SLOT_CHAIN)) return;
        s = (s + 1) & t.cap_mask;
    }
}

__device__ __forceinline__ bool filter_test_global(const TableView& t, uint64_t canon)
{
    const uint32_t w = t.filter[vg_fhash_word(canon) >> t.filter_shift];
    const uint32_t m = vg_fhash_bits(canon, t.filter_words_log2);
    return (w & m) == m;
}

// ------------------------------------------------------------------------------------------
// row kernel: position-parallel emitter for ODD k, three sinks
// ------------------------------------------------------------------------------------------
enum { MODE_COUNT = 0, MODE_KEYS = 1, MODE_BLOOM = 2, MODE_DEBIT = 3 };

#define VG_QCAP 128u  // per-wave pass queue entries (power of two, >= 2*64)

__device__ __forceinline__ uint4 load_chunk(const uint8_t* bases, uint64_t n_bytes, uint64_t off)
{
    if (off + 16 <= n_bytes) return *reinterpret_cast<const uint4*>(bases + off);
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint64_t o = off + 4 * i + b;
            const uint32_t c = o < n_bytes ? bases[o] : (uint32_t)'\n';
            x |= c << (8 * b);
        }
        w[i] = x;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// 16 ASCII bytes -> be: base t at bits [2(15-t), +2);  inv: bit t set iff byte t is not a base.
// One 512-byte LDS table (seq_nt4_table, include/seq_nt4_table.hpp:5-22): [c] = 2-bit code,
// [256 + c] = invalid flag, so both reads share one address register (the second uses the DS
// offset field) and each result is merged with a single shift-or.
__device__ __forceinline__ void encode16(const uint4 raw, const uint8_t* lut, uint32_t& be, uint32_t& inv)
{
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t acc = 0, flags = 0;  // flags: bit (15 - t) = invalid(t), reversed below
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint8_t* e = lut + ((w[i] >> (8 * b)) & 0xFFu);
            acc = (acc << 2) | e[0];
            flags = (flags << 1) | e[256];
        }
    }
    be = acc;
    inv = __builtin_bitreverse32(flags) >> 16;
}

__device__ __forceinline__ void stage_lut(uint8_t* lut, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t i = tid; i < 256; i += nthreads) {
        const uint32_t c = vg_nt4(i);
        lut[i] = (uint8_t)(c & 3u);
        lut[256 + i] = (uint8_t)(c >> 2);
    }
}

template <int MODE>
__device__ __forceinline__ void drain_queue(const TableView& t, uint64_t* queue, uint32_t head, uint32_t n, uint32_t lane)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < n) {
        const uint64_t canon = queue[(head + lane) & (VG_QCAP - 1)];
        table_count(t, canon);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int MODE, bool FLDS>
__global__ __launch_bounds__(1024) void rows_kernel(RowParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t nwaves = blockDim.x >> 6;

    // ---- LDS carve: [filter][queues][lut_code][lut_inv]
    size_t off = 0;
    uint32_t* s_filter = reinterpret_cast<uint32_t*>(smem);
    if (FLDS) off += (size_t)4 << p.table.filter_words_log2;
    uint64_t* s_queue = reinterpret_cast<uint64_t*>(smem + off) + (size_t)wave * VG_QCAP;
    if (MODE == MODE_COUNT) off += (size_t)nwaves * VG_QCAP * 8;
    unsigned long long* s_bkey = reinterpret_cast<unsigned long long*>(smem + off) + (size_t)wave * VG_BPRIV_CELLS;
    if (MODE == MODE_BLOOM) off += (size_t)nwaves * VG_BPRIV_CELLS * 8;
    uint32_t* s_bcnt = reinterpret_cast<uint32_t*>(smem + off) + (size_t)wave * VG_BPRIV_CELLS;
    if (MODE == MODE_BLOOM) off += (size_t)nwaves * VG_BPRIV_CELLS * 4;
    if (MODE == MODE_BLOOM)
        for (uint32_t i = lane; i < VG_BPRIV_CELLS; i += 64) {
            s_bkey[i] = ~0ULL;
            s_bcnt[i] = 0;
        }
    uint8_t* s_lut = smem + off;
    stage_lut(s_lut, tid, blockDim.x);
    if (FLDS) {
        const uint32_t nq = 1u << (p.table.filter_words_log2 - 2);  // uint4 count (>= 1: words >= 4)
        const uint4* src = reinterpret_cast<const uint4*>(p.table.filter);
        uint4* dst = reinterpret_cast<uint4*>(s_filter);
        for (uint32_t i = tid; i < nq; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    // ---- block length: host value, or (device-side FASTQ parser) read from device memory together with what follows from it
    uint64_t n_bytes = p.n_bytes, row_begin = p.row_begin, emit_from = p.emit_from;
    if (p.n_bytes_dev) {
        n_bytes = *p.n_bytes_dev;
        emit_from = 0;
        if (p.tail27) {   // behind count27_kernel: the ends it does not cover (see launch_count in vgmi_api.cpp)
            // 1: count27_kernel (pairs of 768-byte rows), 2: count27x_kernel (single rows), 3: count27s_kernel (pairs of 1 024-byte rows)
            const uint64_t row_bytes = p.tail27 >= 3 ? 1024u : 768u;
            const uint64_t row_end27 = p.tail27 == 2 ? n_bytes / 768 : (n_bytes / (2 * row_bytes)) * 2;
            emit_from = row_end27 ? row_end27 * row_bytes - (p.tail27 == 4 ? 0u : 1u) : 0;
        }
        row_begin = emit_from >> 10;
        if (emit_from >= n_bytes) return;
    }
    // ---- this wave's contiguous row range within [row_begin, total_rows)
    const uint64_t total_rows = (n_bytes + 1023) >> 10;
    if (row_begin >= total_rows) return;
    const uint64_t span = total_rows - row_begin;
    const uint64_t total_waves = (uint64_t)gridDim.x * nwaves;
    const uint64_t rpw = (span + total_waves - 1) / total_waves;
    const uint64_t gw = (uint64_t)blockIdx.x * nwaves + wave;
    uint64_t r0 = row_begin + gw * rpw;
    uint64_t r1 = r0 + rpw < total_rows ? r0 + rpw : total_rows;
    if (r0 >= r1) return;

    const uint32_t K = p.k;                      // odd, 1..27
    const uint64_t mask = (1ULL << (2 * K)) - 1; // 2k-bit mask
    const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
    const uint32_t rc_off = 2 * (33 - K);        // bit offset of the rc window for j = 0
    const uint32_t src1 = (lane + 63u) & 63u, src2 = (lane + 62u) & 63u;

    // carry from the row before r0 (values already rotated by 1 and 2 lanes)
    uint32_t pr1_be = 0, pr2_be = 0, pr1_rc = ~0u, pr2_rc = ~0u, pr1_inv = 0xFFFFu, pr2_inv = 0xFFFFu;
    if (r0 > 0) {
        const uint4 raw = load_chunk(p.bases, n_bytes, ((r0 - 1) << 10) + lane * 16);
        uint32_t be, inv;
        encode16(raw, s_lut, be, inv);
        const uint32_t rcw = rc_word(be);
        pr1_be = __shfl(be, src1);   pr2_be = __shfl(be, src2);
        pr1_rc = __shfl(rcw, src1);  pr2_rc = __shfl(rcw, src2);
        pr1_inv = __shfl(inv, src1); pr2_inv = __shfl(inv, src2);
    }

    uint32_t qhead = 0, qtail = 0;  // wave-uniform ring indices (MODE_COUNT)

    uint4 raw_next = load_chunk(p.bases, n_bytes, (r0 << 10) + lane * 16);
    for (uint64_t r = r0; r < r1; ++r) {
        const uint4 raw = raw_next;
        if (r + 1 < r1) raw_next = load_chunk(p.bases, n_bytes, ((r + 1) << 10) + lane * 16);

        uint32_t be, inv;
        encode16(raw, s_lut, be, inv);
        const uint32_t rcw = rc_word(be);
        // neighbours: lane-1 / lane-2 of this row, or the tail of the previous row
        const uint32_t r1_be = __shfl(be, src1), r2_be = __shfl(be, src2);
        const uint32_t r1_rc = __shfl(rcw, src1), r2_rc = __shfl(rcw, src2);
        const uint32_t r1_inv = __shfl(inv, src1), r2_inv = __shfl(inv, src2);
        const uint32_t F0 = be;
        const uint32_t F1 = lane >= 1 ? r1_be : pr1_be;
        const uint32_t F2 = lane >= 2 ? r2_be : pr2_be;
        const uint32_t R2 = rcw;
        const uint32_t R1 = lane >= 1 ? r1_rc : pr1_rc;
        const uint32_t R0 = lane >= 2 ? r2_rc : pr2_rc;
        const uint32_t i1 = lane >= 1 ? r1_inv : pr1_inv;
        const uint32_t i2 = lane >= 2 ? r2_inv : pr2_inv;
        pr1_be = r1_be; pr2_be = r2_be; pr1_rc = r1_rc; pr2_rc = r2_rc; pr1_inv = r1_inv; pr2_inv = r2_inv;

        // invalid-base mask of the 48-base window (bit i = window base i), smeared forward by
        // k-1 so that bit (32+j) says "some base of the k-mer ending at own base j is invalid"
        uint64_t sm = ((uint64_t)inv << 32) | ((uint64_t)i1 << 16) | (uint64_t)i2;
        {
            uint32_t span = 1;
            while (2 * span <= K) { sm |= sm << span; span *= 2; }
            if (span < K) sm |= sm << (K - span);
        }
        const uint32_t bad = (uint32_t)(sm >> 32);  // bit j

        // empty-read check (reference: assert(len > 0), src/kmer.cpp:124).  Cheap necessary
        // condition first (two adjacent non-base bytes), exact test on the raw bytes only then.
        {
            const uint32_t prev_bit = (i1 >> 15) & 1u;
            const uint32_t adj = inv & ((inv << 1) | prev_bit);
            if (__builtin_expect(__ballot(adj != 0) != 0, 0)) {
                if (adj) {
                    const uint64_t base_off = (r << 10) + lane * 16;
                    for (uint32_t t = 0; t < 16; ++t) {
                        if (!((adj >> t) & 1u)) continue;
                        const uint64_t o = base_off + t;
                        if (o >= n_bytes) continue;
                        if (p.bases[o] == '\n' && (o == 0 || p.bases[o - 1] == '\n')) atomicOr(p.status, 1u);
                    }
                }
            }
        }
        for (uint32_t j = 0; j < 16; ++j) {
            // forward k-mer ending at own base j: bits [2(15-j), +2k) of F2:F1:F0
            const uint32_t fs = 2 * (15 - j);
            const uint32_t f_lo = funnel(F1, F0, fs) & mask_lo;
            const uint32_t f_hi = funnel(F2, F1, fs) & mask_hi;
            // reverse complement: bits [rc_off + 2j, +2k) of R2:R1:R0
            const uint32_t rs = rc_off + 2 * j;
            uint32_t r_lo, r_hi;
            if (rs < 32) {
                r_lo = funnel(R1, R0, rs);
                r_hi = funnel(R2, R1, rs);
            } else if (rs < 64) {
                r_lo = funnel(R2, R1, rs - 32);
                r_hi = R2 >> (rs - 32);
            } else {
                r_lo = R2 >> (rs - 64);
                r_hi = 0;
            }
            r_lo &= mask_lo;
            r_hi &= mask_hi;
            const uint64_t fwd = ((uint64_t)f_hi << 32) | f_lo;
            const uint64_t rc = ((uint64_t)r_hi << 32) | r_lo;
            const uint64_t canon = fwd < rc ? fwd : rc;
            const bool valid = ((bad >> j) & 1u) == 0;

            if (MODE == MODE_KEYS) {
                const uint64_t pos = (r << 10) + lane * 16 + j;
                if (pos < n_bytes) p.keys_out[pos] = valid ? (vg_hash64(canon, mask) << 8 | K) : ~0ULL;
            } else if (MODE == MODE_BLOOM) {
                const uint64_t pos = (r << 10) + lane * 16 + j;
                if (VG_DBG(p.dbg) & 128u) {   // A/B: every lane updates the filter itself
                    if (valid && pos < n_bytes) bloom_add_key(p.bloom, vg_hash64(canon, mask) << 8 | K);
                } else {
                    bloom_add_wave(p.bloom, s_bkey, s_bcnt, valid && pos < n_bytes, vg_hash64(canon, mask) << 8 | K);
                }
            } else {
                const bool emit = valid && (r << 10) + lane * 16 + j >= emit_from;
                bool pass;
                if (FLDS) {
                    const uint32_t h = vg_fhash_word(canon);
                    const uint32_t w = s_filter[h >> p.table.filter_shift];
                    const uint32_t m = vg_fhash_bits_small(h);
                    pass = emit && ((w & m) == m);
                } else {
                    pass = emit && filter_test_global(p.table, canon);
                }
                const uint64_t ball = __ballot(pass);
                if (ball) {
                    if (pass) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                              __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
                        s_queue[(qtail + rank) & (VG_QCAP - 1)] = canon;
                    }
                    qtail += (uint32_t)__popcll(ball);
                    if (qtail - qhead >= 64u) {
                        drain_queue<MODE>(p.table, s_queue, qhead, 64u, lane);
                        qhead += 64u;
                    }
                }
            }
        }
    }
    if (MODE == MODE_COUNT) {
        if (qtail != qhead) drain_queue<MODE>(p.table, s_queue, qhead, qtail - qhead, lane);
    }
}

// ------------------------------------------------------------------------------------------
// fast count kernels: k = 27 (every BASELINE.json configuration)
//
// Row walk: 768-byte rows, 12 bytes (one dwordx3) per lane, neighbours by ds_bpermute.  Every lane owns
// exactly one position of the GRID FILTER (vgmi_device.h; VG_GRID_STEP = 12), so all shifts are constants:
//   * the 16-mer ending at the last base of the PREVIOUS lane's chunk is canonicalised and tested against the
//     blocked Bloom filter (LDS for small graphs, global memory for large ones);
//   * a hit makes the 12 k-mers ending at that base .. 11 bases later candidates.  They are all substrings of
//     38 bases the lane holds (own 12 + 36 halo), so the lane queues ONE entry for the whole RUN: 76 bits of
//     bases + a 12-bit validity mask (windows containing a non-base are dropped);
//   * drain: whenever >= 5 runs are queued, lanes 0..59 expand 5 runs into 60 k-mers (lane -> run = lane / 12,
//     window = lane % 12), canonicalise (min with the v_bfrev reverse complement), hash and ISSUE one 16-byte
//     exact-table load each; the batch issued by the previous drain step is FINISHED first: compare, atomic
//     add for the hits, and a lane whose slot holds another k-mer re-queues its k-mer with the probe distance
//     advanced (a dependent load would stall the whole wave); re-queued k-mers ride in a second small ring
//     and get their own batches.
//
// Vector memory of the hot loop (row prefetch, table loads, counter atomics) is issued and waited for BY HAND
// (inline asm + explicit "s_waitcnt vmcnt(n)"): which of them are outstanding depends on wave-uniform run-time
// state (did the previous row drain?), and the compiler, which has to pick one immediate per program point,
// can only answer vmcnt(0) -- every row would then wait for the table loads and atomics issued a moment ago.
// The kernel instead counts, in SGPRs, how many of its own operations were issued after the one it needs and
// branches to the matching immediate.  Atomics are the returning kind: returning operations complete in issue
// order, which is what makes the counts exact.  The registers written asynchronously (vm_buf, vm_tv, vm_ret) are
// operands of every wait, so nothing that reads them can move above it.
// ------------------------------------------------------------------------------------------
#define VG_Q_KMER_MASK ((1ULL << 54) - 1)
#define VG_SAT_REGION_LOG2 11u   // TableView::sat_dirty granularity
#define VG_RUNQ 80u       // run ring: 16-byte entries {bases[31:0], bases[63:32], bases[75:64] | valid12 << 12, -};
                          // a row adds <= 64 runs, <= 4 are left over and the kernel drains between the two rows of
                          // a pair when needed, so it cannot overflow (LDS: 128 KiB filter + 16 waves x (80 x 16 +
                          // 64 x 8) B + 4 KiB LUT = 163 840 B = all of the 160 KiB)
#define VG_RUN_BATCH 5u   // runs per probe batch (60 k-mers)
#define VG_REQ 64u        // re-queue ring: 64-bit entries (canonical k-mer | probe distance << 54)

// The asynchronously written registers are FIXED physical VGPRs at the top of the 128-register budget both
// variants run with (occupancy is LDS- resp. workgroup-limited to 4 waves per SIMD): the compiler never learns
// that a value lives there, so it cannot copy or spill it while the data is still in flight; naming them as
// clobbers makes it allocate the full budget and keep its own values (about 60 VGPRs) far below.
// tools/check_hot_vgprs.py verifies on the generated ISA that nothing else touches them.
//   v[116:118] / v[120:122] row prefetch (first / second row of a pair)     v123 atomic return (never read)
//   v[124:127] table slots of the batch in flight
#define VG_HOT_CLOBBERS "v116", "v117", "v118", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define VG_HOT_CLOBBERS_S "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

template <int R>   // R = 0: first row of the pair -> v[116:118], 1: second row -> v[120:122]
__device__ __forceinline__ void vm_load_row(const uint8_t* ptr)
{
    if (R == 0) asm volatile("global_load_dwordx3 v[116:118], %0, off nt ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS);
    else asm volatile("global_load_dwordx3 v[120:122], %0, off nt ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS);
}
// count27s_kernel: 16 bytes per lane, v[116:119] first / v[120:123] second row of a pair; table word v[124:125], atomic return v126
template <int R>
__device__ __forceinline__ void vms_load_row(const uint8_t* ptr)
{
    if (R == 0) asm volatile("global_load_dwordx4 v[116:119], %0, off nt ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS_S);
    else asm volatile("global_load_dwordx4 v[120:123], %0, off nt ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS_S);
}
__device__ __forceinline__ void vms_load_slot(const void* ptr)
{
    asm volatile("global_load_dwordx2 v[124:125], %0, off ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS_S);
}
__device__ __forceinline__ void vms_atomic_inc(uint32_t* ptr, uint32_t one)
{
    asm volatile("global_atomic_add v126, %0, %1, off sc0 ; VGHOT" : : "v"(ptr), "v"(one) : "memory", VG_HOT_CLOBBERS_S);
}
__device__ __forceinline__ uint2 vms_slot_value()   // after the wait
{
    uint2 v;
    asm volatile("v_mov_b32 %0, v124 ; VGHOT\n\tv_mov_b32 %1, v125 ; VGHOT" : "=v"(v.x), "=v"(v.y));
    return v;
}
__device__ __forceinline__ uint32_t vms_atomic_old()   // return value of the last vms_atomic_inc, after the wait
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, v126 ; VGHOT" : "=v"(v));
    return v;
}

// path-table drain of count27s_kernel<true>: the index bucket's two entries v[108:111], v[112:115]; per place (0..3) 16 bytes of
// sequence v[92 + 4 c : 95 + 4 c], the words with the k-mer-start bits v[76 + 4 c : 77 + 4 c] and with the saturation bits v[78 + 4 c : 79 + 4 c]
#define VG_HOT_CLOBBERS_P "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", \
                          "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", \
                          "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", VG_HOT_CLOBBERS_S
__device__ __forceinline__ void vmp_load_index(const void* ptr)     // 32 bytes: both entries of the bucket
{
    asm volatile("global_load_dwordx4 v[108:111], %0, off ; VGHOT\n\tglobal_load_dwordx4 v[112:115], %0, off offset:16 ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS_P);
}
#define VG_SEQ_ASM(C, SQ, VBR, SBR) \
    if (c == C) asm volatile("global_load_dwordx4 " SQ ", %0, off ; VGHOT\n\tglobal_load_dwordx2 " VBR ", %1, off ; VGHOT\n\tglobal_load_dwordx2 " SBR ", %2, off ; VGHOT" \
                             : : "v"(sq), "v"(vb), "v"(sb) : VG_HOT_CLOBBERS_P)
template <int c>
__device__ __forceinline__ void vmp_load_place(const void* sq, const void* vb, const void* sb)     // three loads
{
    VG_SEQ_ASM(0, "v[92:95]", "v[76:77]", "v[78:79]");
    VG_SEQ_ASM(1, "v[96:99]", "v[80:81]", "v[82:83]");
    VG_SEQ_ASM(2, "v[100:103]", "v[84:85]", "v[86:87]");
    VG_SEQ_ASM(3, "v[104:107]", "v[88:89]", "v[90:91]");
}
#define VG_MOV4(A, B, C, D) asm volatile("v_mov_b32 %0, " A " ; VGHOT\n\tv_mov_b32 %1, " B " ; VGHOT\n\tv_mov_b32 %2, " C " ; VGHOT\n\tv_mov_b32 %3, " D " ; VGHOT" \
                                         : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w))
template <int e>
__device__ __forceinline__ uint4 vmp_index_value()
{
    uint4 v;
    if (e == 0) VG_MOV4("v108", "v109", "v110", "v111");
    else VG_MOV4("v112", "v113", "v114", "v115");
    return v;
}
template <int c>
__device__ __forceinline__ uint4 vmp_seq_value()
{
    uint4 v;
    if (c == 0) VG_MOV4("v92", "v93", "v94", "v95");
    if (c == 1) VG_MOV4("v96", "v97", "v98", "v99");
    if (c == 2) VG_MOV4("v100", "v101", "v102", "v103");
    if (c == 3) VG_MOV4("v104", "v105", "v106", "v107");
    return v;
}
template <int c>
__device__ __forceinline__ uint4 vmp_bits_value()     // {start bits lo, hi, saturation bits lo, hi}
{
    uint4 v;
    if (c == 0) VG_MOV4("v76", "v77", "v78", "v79");
    if (c == 1) VG_MOV4("v80", "v81", "v82", "v83");
    if (c == 2) VG_MOV4("v84", "v85", "v86", "v87");
    if (c == 3) VG_MOV4("v88", "v89", "v90", "v91");
    return v;
}
// rare path of the path-table drain: four independent dword loads / four returning atomic increments, one wait for all
__device__ __forceinline__ void vm_load_dword4_sync(const uint32_t* p0, const uint32_t* p1, const uint32_t* p2, const uint32_t* p3, uint32_t (&v)[4])
{
    asm volatile("global_load_dword %0, %4, off ; VGHOT\n\tglobal_load_dword %1, %5, off ; VGHOT\n\tglobal_load_dword %2, %6, off ; VGHOT\n\t"
                 "global_load_dword %3, %7, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void vm_atomic_add4_sync(uint32_t* p0, uint32_t* p1, uint32_t* p2, uint32_t* p3, const uint32_t (&add)[4], uint32_t (&old)[4])
{
    asm volatile("global_atomic_add %0, %4, %8, off sc0 ; VGHOT\n\tglobal_atomic_add %1, %5, %9, off sc0 ; VGHOT\n\t"
                 "global_atomic_add %2, %6, %10, off sc0 ; VGHOT\n\tglobal_atomic_add %3, %7, %11, off sc0 ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT"
                 : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(add[0]), "v"(add[1]), "v"(add[2]), "v"(add[3]) : "memory");
}
__device__ __forceinline__ uint32_t vm_load_dword_sync(const uint32_t* ptr)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : "=&v"(v) : "v"(ptr) : "memory");
    return v;
}
// rare path: a returning atomic waited for on the spot
__device__ __forceinline__ uint32_t vm_atomic_inc_sync(uint32_t* ptr, uint32_t one)
{
    uint32_t old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0 ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : "=&v"(old) : "v"(ptr), "v"(one) : "memory");
    return old;
}
// n = hot operations issued after the one whose result is needed: waits with vmcnt(min(n, 4)) (a drain step of the path-table
// variant puts up to four operations in flight)
__device__ __forceinline__ void vm_wait4(uint32_t n)
{
    asm volatile("s_cmp_lt_u32 %0, 4 ; VGHOT\n\t"
                 "s_cbranch_scc1 1f ; VGHOT\n\t"
                 "s_waitcnt vmcnt(4) ; VGHOT\n\t"
                 "s_branch 9f ; VGHOT\n"
                 "1:\n\t"
                 "s_cmp_lt_u32 %0, 2 ; VGHOT\n\t"
                 "s_cbranch_scc1 2f ; VGHOT\n\t"
                 "s_cmp_eq_u32 %0, 2 ; VGHOT\n\t"
                 "s_cbranch_scc1 3f ; VGHOT\n\t"
                 "s_waitcnt vmcnt(3) ; VGHOT\n\t"
                 "s_branch 9f ; VGHOT\n"
                 "3:\n\t"
                 "s_waitcnt vmcnt(2) ; VGHOT\n\t"
                 "s_branch 9f ; VGHOT\n"
                 "2:\n\t"
                 "s_cmp_eq_u32 %0, 0 ; VGHOT\n\t"
                 "s_cbranch_scc1 4f ; VGHOT\n\t"
                 "s_waitcnt vmcnt(1) ; VGHOT\n\t"
                 "s_branch 9f ; VGHOT\n"
                 "4:\n\t"
                 "s_waitcnt vmcnt(0) ; VGHOT\n"
                 "9:"
                 : : "s"(__builtin_amdgcn_readfirstlane((int)n)) : "scc", "memory");
}

template <bool COMPACT>   // compact table format: the 8-byte k-mer word only
__device__ __forceinline__ void vm_load_slot(const void* ptr)
{
    if (COMPACT) asm volatile("global_load_dwordx2 v[124:125], %0, off ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS);
    else asm volatile("global_load_dwordx4 v[124:127], %0, off ; VGHOT" : : "v"(ptr) : VG_HOT_CLOBBERS);
}
__device__ __forceinline__ void vm_atomic_inc(uint32_t* ptr, uint32_t one)
{
    asm volatile("global_atomic_add v123, %0, %1, off sc0 ; VGHOT" : : "v"(ptr), "v"(one) : "memory", VG_HOT_CLOBBERS);
}
template <int N>
__device__ __forceinline__ void vm_wait_imm()
{
    asm volatile("s_waitcnt vmcnt(%0) ; VGHOT" : : "n"(N) : "memory");
}
// n = number of hot operations issued after the one whose result is needed (wave-uniform, SGPR): waits with
// vmcnt(min(n, 2)) -- an immediate below the exact count is only stricter.  One asm block: the compiler's own
// lowering of the equivalent if-chain costs ~20 scalar instructions.
__device__ __forceinline__ void vm_wait(uint32_t n)
{
    asm volatile("s_cmp_lt_u32 %0, 2 ; VGHOT\n\t"
                 "s_cbranch_scc1 1f ; VGHOT\n\t"
                 "s_waitcnt vmcnt(2) ; VGHOT\n\t"
                 "s_branch 3f ; VGHOT\n"
                 "1:\n\t"
                 "s_cmp_eq_u32 %0, 0 ; VGHOT\n\t"
                 "s_cbranch_scc1 2f ; VGHOT\n\t"
                 "s_waitcnt vmcnt(1) ; VGHOT\n\t"
                 "s_branch 3f ; VGHOT\n"
                 "2:\n\t"
                 "s_waitcnt vmcnt(0) ; VGHOT\n"
                 "3:"
                 : : "s"(__builtin_amdgcn_readfirstlane((int)n)) : "scc", "memory");
}
// rare paths: synchronous loads / fire-and-forget atomics, also by hand -- a single compiler-visible memory operation
// in the loop makes the compiler add its own vmcnt(0) waits in the hot path
template <bool COMPACT>
__device__ __forceinline__ uint4 vm_load_slot_sync(const void* ptr)
{
    if (COMPACT) {
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        u32x2_t v;
        asm volatile("global_load_dwordx2 %0, %1, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : "=&v"(v) : "v"(ptr) : "memory");
        return make_uint4(v.x, v.y, 0u, 0u);
    }
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : "=&v"(v) : "v"(ptr) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t vm_load_byte_sync(const uint8_t* ptr)
{
    uint32_t v;
    asm volatile("global_load_ubyte %0, %1, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : "=&v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void vm_atomic_or_sync(uint32_t* ptr, uint32_t bits)
{
    asm volatile("global_atomic_or %0, %1, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : : "v"(ptr), "v"(bits) : "memory");
}
__device__ __forceinline__ void vm_store_byte_sync(uint8_t* ptr, uint32_t v)
{
    asm volatile("global_store_byte %0, %1, off ; VGHOT\n\ts_waitcnt vmcnt(0) ; VGHOT" : : "v"(ptr), "v"(v) : "memory");
}
template <bool COMPACT>
__device__ __forceinline__ uint4 vm_slot_value()   // after the wait
{
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (COMPACT) asm volatile("v_mov_b32 %0, v124 ; VGHOT\n\tv_mov_b32 %1, v125 ; VGHOT" : "=v"(v.x), "=v"(v.y));
    else asm volatile("v_mov_b32 %0, v124 ; VGHOT\n\tv_mov_b32 %1, v125 ; VGHOT\n\tv_mov_b32 %2, v126 ; VGHOT\n\tv_mov_b32 %3, v127 ; VGHOT"
                      : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w));
    return v;
}
__device__ __forceinline__ uint32_t vm_atomic_old()   // return value of the last vm_atomic_inc, after the wait
{
    uint32_t v;
    asm volatile("v_mov_b32 %0, v123 ; VGHOT" : "=v"(v));
    return v;
}

// Row geometry of the k = 27 kernels
#define VG_ROW27 768u

// Base LUT of the k = 27 kernels, at LDS byte offset 0: two sets (A: dwords 0 and 2 of a lane's chunk, B: dword 1)
// of 4 position tables of 256 u16.  Entry for byte b of a dword = code << 2(3 - b) | invalid << (8 + b) (set A) or
// invalid << (12 + b) (set B); the OR of a dword's four entries is {4 packed codes, first base most significant} in
// byte 0 and its 4 invalid bits above.
#define VG_LUT27_BYTES 4096u
typedef __attribute__((address_space(3))) const uint16_t lds_u16;
typedef __attribute__((address_space(3))) const uint32_t lds_u32;

__device__ __forceinline__ void stage_lut27(uint32_t tid, uint32_t nthreads)
{
    typedef __attribute__((address_space(3))) uint16_t lds_u16_rw;
    for (uint32_t i = tid; i < 2048; i += nthreads) {
        const uint32_t set = i >> 10, b = (i >> 8) & 3u, c = vg_nt4(i & 255u);
        reinterpret_cast<lds_u16_rw*>((uintptr_t)0)[i] =
            (uint16_t)(((c & 3u) << (2 * (3 - b))) | ((c >> 2) << ((set ? 12 : 8) + b)));
    }
}

struct Addr4 { uint32_t a0, a1, a2, a3; };   // LDS byte offsets of a dword's four LUT entries (before the table offsets)

// 2 * byte b of dword D of landed row R (v116 + 4 R + D), one SDWA op each (the LUT holds u16)
#define VG_SDWA_X2(REG, B) "v_lshlrev_b32_sdwa %" #B ", %4, " REG " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #B " ; VGHOT\n\t"
#define VG_SDWA_ROW(REG) asm volatile(VG_SDWA_X2(REG, 0) VG_SDWA_X2(REG, 1) VG_SDWA_X2(REG, 2) VG_SDWA_X2(REG, 3) \
                                      : "=&v"(a.a0), "=&v"(a.a1), "=&v"(a.a2), "=&v"(a.a3) : "v"(one))
template <int R, int D>
__device__ __forceinline__ Addr4 lut_addr4(const uint32_t one)
{
    Addr4 a;
    if (R == 0 && D == 0) VG_SDWA_ROW("v116");
    if (R == 0 && D == 1) VG_SDWA_ROW("v117");
    if (R == 0 && D == 2) VG_SDWA_ROW("v118");
    if (R == 1 && D == 0) VG_SDWA_ROW("v120");
    if (R == 1 && D == 1) VG_SDWA_ROW("v121");
    if (R == 1 && D == 2) VG_SDWA_ROW("v122");
    return a;
}

template <int R, int D>   // count27s_kernel: dword D (0..3) of landed row R (v116 + 4 R + D)
__device__ __forceinline__ Addr4 luts_addr4(const uint32_t one)
{
    Addr4 a;
    if (R == 0 && D == 0) VG_SDWA_ROW("v116");
    if (R == 0 && D == 1) VG_SDWA_ROW("v117");
    if (R == 0 && D == 2) VG_SDWA_ROW("v118");
    if (R == 0 && D == 3) VG_SDWA_ROW("v119");
    if (R == 1 && D == 0) VG_SDWA_ROW("v120");
    if (R == 1 && D == 1) VG_SDWA_ROW("v121");
    if (R == 1 && D == 2) VG_SDWA_ROW("v122");
    if (R == 1 && D == 3) VG_SDWA_ROW("v123");
    return a;
}

template <int SET>
__device__ __forceinline__ uint32_t encode4(const Addr4& a)
{
    const uint32_t e0 = *reinterpret_cast<lds_u16*>((uintptr_t)(a.a0 + SET * 2048u));
    const uint32_t e1 = *reinterpret_cast<lds_u16*>((uintptr_t)(a.a1 + SET * 2048u + 512u));
    const uint32_t e2 = *reinterpret_cast<lds_u16*>((uintptr_t)(a.a2 + SET * 2048u + 1024u));
    uint32_t e3 = *reinterpret_cast<lds_u16*>((uintptr_t)(a.a3 + SET * 2048u + 1536u));
    asm("" : "+v"(e3));   // hide the value range: keeps the ORs 32-bit (v_or3_b32) instead of 16-bit ops + re-extension
    return e0 | e1 | e2 | e3;
}

// LDS_BM: grid filter (2^15 words) staged in LDS (small graphs, one 1024-thread workgroup per CU)
//         or probed in global memory (large graphs, 256-thread workgroups).
template <bool LDS_BM, bool COMPACT>
__global__ __launch_bounds__(LDS_BM ? 1024 : 256) void count27_kernel(RowParams p)
{
    constexpr uint32_t MASK_HI = (1u << (2 * 27 - 32)) - 1;  // 54-bit k-mer: low word full, 22 bits high
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t nwaves = blockDim.x >> 6;

    // LDS carve: [base LUT, at byte offset 0][grid filter (LDS_BM)][run rings][re-queue rings]; everything is
    // addressed by absolute 32-bit LDS offsets (address_space(3)), the rings' bases are wave-uniform
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 lds_u4_rw;
    typedef __attribute__((address_space(3))) uint64_t lds_u64_rw;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t rings0 = VG_LUT27_BYTES + (LDS_BM ? VG_GRID_LDS_WORDS * 4u : 0u);
    const uint32_t runs_base = rings0 + wave_u * (VG_RUNQ * 16u);
    const uint32_t req_base = rings0 + nwaves * (VG_RUNQ * 16u) + wave_u * (VG_REQ * 8u);
    stage_lut27(tid, blockDim.x);
    if (LDS_BM) {
        const uint4* src = reinterpret_cast<const uint4*>(p.table.grid);
        uint4* dst = reinterpret_cast<uint4*>(smem + VG_LUT27_BYTES);
        for (uint32_t i = tid; i < VG_GRID_LDS_WORDS / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t* g_grid = p.table.grid;
    const uint32_t gwl = LDS_BM ? VG_GRID_LDS_WORDS_LOG2 : p.table.grid_words_log2;
    VgSlot* const slots = p.table.slots;                 // 16-byte format (global-filter variant)
    unsigned long long* const slots8 = p.table.slots8;   // compact format (LDS-filter variant)
    uint32_t* const counts = p.table.counts;
    const uint64_t cap_mask = p.table.cap_mask;
    const uint32_t hb_log2 = p.table.home_bucket_log2;
    const bool hb_off = p.table.home_by_offset != 0;

    // rows [0, row_end) are complete 768-byte rows (row_end is even, see vgmi_api.cpp), so every load below is an
    // unconditional, perfectly coalesced dwordx3 (the ragged tail goes to rows_kernel).  A wave walks a
    // contiguous range of row PAIRS: two rows per iteration give every wave two independent dependency chains
    // (LUT reads -> permutes -> filter word), which four waves per SIMD need to keep the VALU fed.
    const uint64_t total_pairs = p.n_bytes_dev ? *p.n_bytes_dev / (2 * VG_ROW27) : p.row_end >> 1;   // device-side block length: vgmi_fastq.hip
    const uint64_t total_waves = (uint64_t)gridDim.x * nwaves;
    const uint64_t ppw = (total_pairs + total_waves - 1) / total_waves;
    // the wave index is uniform: keep the row counters in SGPRs
    const uint64_t gw = (uint64_t)blockIdx.x * nwaves + wave_u;
    const uint64_t r0v = gw * ppw;
    const uint64_t r1v = r0v + ppw < total_pairs ? r0v + ppw : total_pairs;
    const uint64_t r0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r0v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r0v);
    const uint64_t r1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r1v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r1v);
    if (r0 >= r1) return;
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));   // SDWA takes no inline constant for the shift amount
    const uint32_t lane_off = lane * 12u;
    const uint8_t* const bases = p.bases;

    // previous row's words, already rotated by 1 / 2 / 3 lanes (lanes 0..2 take them)
    uint32_t pr1_be = 0, pr2_be = 0, pr3_be = 0, pr1_inv = 0xFFFu, pr2_inv = 0xFFFu, pr3_inv = 0xFFFu;
    // the pair before the range is walked first as a warm-up iteration: it only provides the halo
    const uint64_t rs = r0 > 0 ? r0 - 1 : r0;

    // ---- asynchronous state (see the header comment) ----
    uint32_t n_after_row = 0;             // hot operations issued after the prefetch of the row buffer
    uint32_t n_after_slot = 2;            // ... after the table loads of the batch in flight
    uint64_t b_canon = 0, b_slot = 0;     // per lane: canonical k-mer | probe distance << 54, slot probed
    bool b_active = false;                // per lane: takes part in the batch in flight (none at the start: the
                                          // first step "finishes" an empty batch, so the step needs no special case)
    uint64_t p_slot = 0;                  // compact format: slot whose counter this lane bumped in the previous step
    bool p_bumped = false;

    // drain bookkeeping: which run / window of a 5-run batch this lane expands.  Ring state is wave-uniform
    // (SGPR) and kept reduced mod the ring size; a lane's slot wraps with one subtract + min.
    const uint32_t my_run = lane / 12u, my_win = lane % 12u;
    const uint32_t my_sh = 2 * (11 - my_win), my_vbit = 12 + my_win;
    uint32_t run_head = 0, run_n = 0, req_head = 0, req_n = 0;
    auto ring_slot = [](uint32_t pos) -> uint32_t {  // pos < 2 * VG_RUNQ
        const uint32_t w = pos - VG_RUNQ;
        return w < pos ? w : pos;
    };

    // canonicalise + hash for every lane (cheap), load only for the lanes that take part (>= 1 by construction,
    // so the load below is always issued and the counters stay exact)
    auto probe_issue = [&](bool act, uint64_t kmer, uint64_t dist) __attribute__((always_inline)) {
        const uint64_t rc = vg_revcomp(kmer, 27);
        const uint64_t canon = kmer < rc ? kmer : rc;   // idempotent for re-queued (already canonical) entries
        b_canon = canon | (dist << 54);
        const uint64_t home = (!LDS_BM && hb_log2) ? vg_thash_local(canon, canon == kmer ? rc : kmer, hb_log2, hb_off) : vg_thash(canon);
        b_slot = (home + dist) & cap_mask;
        b_active = act;
        if (act) vm_load_slot<COMPACT>(COMPACT ? (const void*)&slots8[b_slot] : (const void*)&slots[b_slot]);
        ++n_after_row;
        n_after_slot = 0;
    };
    // one drain step: finish the batch in flight (compare, atomic add for the hits, re-queue the collisions),
    // then issue the next batch: 32+ re-queued k-mers first, else up to 5 runs
    auto drain_step = [&]() __attribute__((always_inline)) {
        vm_wait(n_after_slot);
        if (COMPACT) {
            // the previous step's atomics have returned (they were issued before the loads just waited for): a
            // counter that has reached the clamp gets its slot flagged, later hits skip their atomic
            const uint32_t old = vm_atomic_old();
            // exactly one increment takes a counter from 254 to 255: that lane flags the slot and puts it on the reset
            // list (every lane that sees an older value >= 254 doing so would serialise thousands of same-address
            // atomics on the list counter at deep coverage)
            const bool sat = p_bumped && old == 254u;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(sat) != 0, 0)) {
                if (sat) {
                    vm_atomic_or_sync(reinterpret_cast<uint32_t*>(&slots8[p_slot]) + 1, (uint32_t)(VG_SLOT_SAT >> 32));
                    // the slot's region is marked for the per-sample reset (a plain store of 1: no contention)
                    vm_store_byte_sync(p.table.sat_dirty + (p_slot >> VG_SAT_REGION_LOG2), 1u);
                }
            }
        }
        const uint4 tv = vm_slot_value<COMPACT>();
        bool again = false;
        uint32_t* bump = nullptr;
        if (b_active) {
            const uint64_t c = ((uint64_t)tv.y << 32) | tv.x;
            const uint64_t canon = b_canon & VG_Q_KMER_MASK;
            if (c != VG_EMPTY && (c & VG_SLOT_KMER_MASK) == canon) {
                if (COMPACT) { if (!(c & VG_SLOT_SAT)) bump = &counts[b_slot]; }
                else if (counts) bump = &counts[tv.w];               // dense counters, clamped at read-out
                else if (tv.z < 255u) bump = &slots[b_slot].count;
            } else if (c != VG_EMPTY && (c & VG_SLOT_CHAIN)) {
                again = !(VG_DBG(p.dbg) & 64u);   // 64: ablation (wrong counts): collisions are dropped instead of re-queued
            }
        }
        const uint64_t ball = __builtin_amdgcn_ballot_w64(again);
        if (ball != 0) {
            const uint32_t n = (uint32_t)__builtin_popcountll(ball);
            const bool fits = req_n + n <= VG_REQ && (b_canon >> 54) < 1023u;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(again && !fits) == 0, 1)) {
                if (again) {
                    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                         __builtin_amdgcn_mbcnt_lo((uint32_t)ball, req_head + req_n));
                    *reinterpret_cast<lds_u64_rw*>((uintptr_t)(req_base + (pos & (VG_REQ - 1)) * 8u)) = b_canon + (1ULL << 54);
                }
                req_n += n;
            } else if (again) {
                // ring full (or probe distance field exhausted): chase the chain right here, synchronously
                const uint64_t canon = b_canon & VG_Q_KMER_MASK;
                uint64_t sl = b_slot;
                for (;;) {
                    sl = (sl + 1) & cap_mask;
                    const uint4 v = vm_load_slot_sync<COMPACT>(COMPACT ? (const void*)&slots8[sl] : (const void*)&slots[sl]);
                    const uint64_t c = ((uint64_t)v.y << 32) | v.x;
                    if (c != VG_EMPTY && (c & VG_SLOT_KMER_MASK) == canon) {
                        if (COMPACT) { if (!(c & VG_SLOT_SAT)) { bump = &counts[sl]; b_slot = sl; } }
                        else if (counts) bump = &counts[v.w];
                        else if (v.z < 255u) bump = &slots[sl].count;
                        break;
                    }
                    if (c == VG_EMPTY || !(c & VG_SLOT_CHAIN)) break;
                }
            }
        }
        p_bumped = bump != nullptr;
        p_slot = b_slot;
        if (__builtin_amdgcn_ballot_w64(bump != nullptr)) {   // wave-uniform: the atomic is issued iff some lane hit
            if (bump) vm_atomic_inc(bump, one);
            ++n_after_row;
        }
        __builtin_amdgcn_wave_barrier();
        // ring bookkeeping first, branch-free (both rings' counters are updated on both paths: a common tail the
        // optimiser could sink through a pointer select would push them into scratch memory)
        const bool do_req = req_n >= 32u || run_n == 0;
        const uint32_t take_req = do_req ? (req_n < 64u ? req_n : 64u) : 0u;   // may be 0 (final flush)
        const uint32_t take_run = do_req ? 0u : (run_n < VG_RUN_BATCH ? run_n : VG_RUN_BATCH);
        const uint32_t rq_head = req_head, rn_head = run_head;
        req_head += take_req;
        req_n -= take_req;
        run_head += take_run;
        if (run_head >= VG_RUNQ) run_head -= VG_RUNQ;
        run_n -= take_run;
        if (do_req) {
            const bool have = lane < take_req;
            uint64_t e = 0;
            if (have) e = *reinterpret_cast<lds_u64_rw*>((uintptr_t)(req_base + ((rq_head + lane) & (VG_REQ - 1)) * 8u));
            if (take_req) probe_issue(have, e & VG_Q_KMER_MASK, e >> 54);
            else b_active = false;   // an all-idle batch, no load
        } else {
            const bool have = my_run < take_run;
            u32x4 e = {0u, 0u, 0u, 0u};
            if (have) e = *reinterpret_cast<lds_u4_rw*>((uintptr_t)(runs_base + ring_slot(rn_head + my_run) * 16u));
            // k-mer ending at window my_win of the run: bits [2(11 - win), +54) of the 76 run bits
            const uint32_t lo = funnel(e.y, e.x, my_sh);
            const uint32_t hi = funnel(e.z, e.y, my_sh) & MASK_HI;
            // dbg 32: ablation (wrong counts): one probing lane per run instead of up to 12
            probe_issue(have && ((e.z >> my_vbit) & 1u) && (!(VG_DBG(p.dbg) & 32u) || my_win == (__builtin_ctz(e.z >> 12) & 15u)),
                        ((uint64_t)hi << 32) | lo, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };

    // scan of one row once its words and its neighbours' are known: window, validity, grid probe; returns the
    // hit predicate and leaves the run entry in d0..d2
    struct RowScan { uint32_t W0, W1, W2, vm, gm, gw32; bool ok16; };
    auto scan_probe = [&](uint32_t be, uint32_t inv, uint32_t be1, uint32_t be2, uint32_t be3, uint32_t i1, uint32_t i2,
                          uint32_t i3) __attribute__((always_inline)) -> RowScan {
        RowScan r;
        // 48-base window, base q = 0..47 at bits 2(47 - q) of W2:W1:W0: q = 36..47 own chunk, 0..35 the three
        // chunks before.  This lane's grid position is g = q 35 (the last base of the previous chunk):
        //   grid 16-mer  q 20..35                       bits [24, 56)
        //   run          q  9..46 (k-mers ending at q 35..46, i.e. g .. g + 11)   bits [2, 78)
        r.W0 = (be1 << 24) | be;
        r.W1 = (be2 << 16) | (be1 >> 8);
        r.W2 = (be3 << 8) | (be2 >> 16);
        // non-base bits.  B bit i = base q 9 + i (i = 0..26, the span of the k-mer ending at g): the k-mer
        // ending at g + j is spoilt by B iff B >> j != 0, and by the own chunk iff one of own bases 0..j-1 is
        // a non-base (prefix-OR = x | -x).
        const uint32_t B = (i3 >> 9) | (i2 << 3) | (i1 << 15);
        const uint32_t a = (inv << 1) & 0xFFFu;
        const uint32_t bad_b = B ? (0xFFFFFFFFu >> __builtin_clz(B)) : 0u;
        r.vm = ~(a | (0u - a) | bad_b) & 0xFFFu;
        r.ok16 = B < 2048u;   // no non-base in q 20..35 (the 16-mer itself)
        const uint32_t mer = funnel(r.W1, r.W0, 24);
        uint64_t gx;
        if (LDS_BM) {
            vg_grid_probe(mer, gwl, gx, r.gm);
            r.gw32 = *reinterpret_cast<lds_u32*>((uintptr_t)(((uint32_t)gx << 2) + VG_LUT27_BYTES));
        } else {
            uint32_t rot;
            bool as_is;
            vg_grid_probe(mer, gwl, gx, r.gm, rot, as_is);
            const uint2 g = reinterpret_cast<const uint2*>(g_grid)[gx];
            r.gw32 = g.x;
            // offset bits of the 12 windows: window w has the 16-mer w bases before the k-mer's end
            const uint32_t rr = funnel(g.y, g.y, rot);   // rotate right: bit b -> bit 0 + b
            r.vm &= as_is ? rr : (__builtin_bitreverse32(rr) >> 20);
        }
        return r;
    };
    auto enqueue = [&](const RowScan& r, uint64_t ball, uint32_t tail) __attribute__((always_inline)) {
        if (__builtin_amdgcn_inverse_ballot_w64(ball)) {
            const uint32_t d0 = funnel(r.W1, r.W0, 2);
            const uint32_t d1 = funnel(r.W2, r.W1, 2);
            const uint32_t d2 = ((r.W2 >> 2) & 0xFFFu) | (r.vm << 12);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, tail));
            *reinterpret_cast<lds_u4_rw*>((uintptr_t)(runs_base + ring_slot(pos) * 16u)) = u32x4{d0, d1, d2, 0u};
        }
    };
    auto empty_read_check = [&](uint32_t adj, const uint8_t* row) __attribute__((always_inline)) {
        // reference: assert(len > 0), src/kmer.cpp:124 -- two adjacent non-bases are necessary; exact test on the bytes
        if (adj) {
            const uint64_t base_off = (uint64_t)(row - bases) + lane_off;
            for (uint32_t t = 0; t < 12; ++t) {
                if (!((adj >> t) & 1u)) continue;
                const uint64_t o = base_off + t;
                if (vm_load_byte_sync(bases + o) == '\n' && (o == 0 || vm_load_byte_sync(bases + o - 1) == '\n'))
                    vm_atomic_or_sync(p.status, 1u);
            }
        }
    };

    // 32-bit trip count and a scalar row pointer: the loop control stays on the SALU
    const uint32_t n_it = (uint32_t)(r1 - rs), n_warm = (uint32_t)(r0 - rs);
    const uint8_t* rowp = bases + rs * (2 * VG_ROW27);
    vm_load_row<0>(rowp + lane_off);
    vm_load_row<1>(rowp + VG_ROW27 + lane_off);
    for (uint32_t it = 0; it < n_it; ++it) {
        // both rows have landed once at most n_after_row younger operations are outstanding
        vm_wait(n_after_row);
        const Addr4 a0 = lut_addr4<0, 0>(one), a1 = lut_addr4<0, 1>(one), a2 = lut_addr4<0, 2>(one);
        const Addr4 c0 = lut_addr4<1, 0>(one), c1 = lut_addr4<1, 1>(one), c2 = lut_addr4<1, 2>(one);
        const uint8_t* const cur = rowp;
        if (it + 1 < n_it) rowp += 2 * VG_ROW27;
        vm_load_row<0>(rowp + lane_off);            // prefetch (the last iteration re-reads its own pair)
        vm_load_row<1>(rowp + VG_ROW27 + lane_off);
        n_after_row = 0;
        n_after_slot += 2;
        // first drain point of the iteration (the second one follows the scan): steps half an iteration apart
        // never wait for table loads that were issued a moment ago
        if (run_n >= VG_RUN_BATCH || req_n >= 32u) drain_step();

        // own 12 bases of each row: be = 24 bits, first base most significant; inv bit t = base t is not a base
        const uint32_t g0 = encode4<0>(a0), g1 = encode4<1>(a1), g2 = encode4<0>(a2);
        const uint32_t h0 = encode4<0>(c0), h1 = encode4<1>(c1), h2 = encode4<0>(c2);
        const uint32_t beA = __builtin_amdgcn_perm(g0, __builtin_amdgcn_perm(g1, g2, 0x0c0c0400u), 0x0c040100u);
        const uint32_t invA = ((g0 | g1) >> 8) | (g2 & 0xF00u);   // g0: bits 8..11, g1: 12..15, g2: 8..11
        const uint32_t beB = __builtin_amdgcn_perm(h0, __builtin_amdgcn_perm(h1, h2, 0x0c0c0400u), 0x0c040100u);
        const uint32_t invB = ((h0 | h1) >> 8) | (h2 & 0xF00u);
        // lane l <- lane l - 1 (wrapping) as a DPP wave rotate: a VALU move instead of an LDS-pipe ds_bpermute (the
        // LDS pipe carries the 24 LUT reads of the pair and is the busier unit); chained for l - 2 and l - 3
        auto ror1 = [](uint32_t v) __attribute__((always_inline)) -> uint32_t {
            return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
        };
        const uint32_t a1_be = ror1(beA), a1_inv = ror1(invA), b1_be = ror1(beB), b1_inv = ror1(invB);
        const uint32_t a2_be = ror1(a1_be), a2_inv = ror1(a1_inv), b2_be = ror1(b1_be), b2_inv = ror1(b1_inv);
        const uint32_t a3_be = ror1(a2_be), a3_inv = ror1(a2_inv), b3_be = ror1(b2_be), b3_inv = ror1(b2_inv);
        // neighbours: lanes l-1..l-3 of the same row; the first lanes take the tail of the row before (row A: second
        // row of the previous pair, row B: row A -- the permutes wrap, so lanes 0..2 of a*_ hold exactly that tail)
        const uint32_t beA1 = lane >= 1 ? a1_be : pr1_be, beA2 = lane >= 2 ? a2_be : pr2_be, beA3 = lane >= 3 ? a3_be : pr3_be;
        const uint32_t iA1 = lane >= 1 ? a1_inv : pr1_inv, iA2 = lane >= 2 ? a2_inv : pr2_inv, iA3 = lane >= 3 ? a3_inv : pr3_inv;
        const uint32_t beB1 = lane >= 1 ? b1_be : a1_be, beB2 = lane >= 2 ? b2_be : a2_be, beB3 = lane >= 3 ? b3_be : a3_be;
        const uint32_t iB1 = lane >= 1 ? b1_inv : a1_inv, iB2 = lane >= 2 ? b2_inv : a2_inv, iB3 = lane >= 3 ? b3_inv : a3_inv;
        pr1_be = b1_be; pr2_be = b2_be; pr3_be = b3_be; pr1_inv = b1_inv; pr2_inv = b2_inv; pr3_inv = b3_inv;
        if (it < n_warm) continue;

        {   // empty-read check (rare path)
            const uint32_t adjA = invA & ((invA << 1) | (iA1 >> 11));
            const uint32_t adjB = invB & ((invB << 1) | (iB1 >> 11));
            if (__builtin_expect(__ballot((adjA | adjB) != 0) != 0, 0)) {
                empty_read_check(adjA, cur);
                empty_read_check(adjB, cur + VG_ROW27);
            }
        }

        const RowScan sa = scan_probe(beA, invA, beA1, beA2, beA3, iA1, iA2, iA3);
        const RowScan sb = scan_probe(beB, invB, beB1, beB2, beB3, iB1, iB2, iB3);
        const uint64_t ballA = __builtin_amdgcn_ballot_w64(sa.ok16 && (sa.gw32 & sa.gm) == sa.gm && sa.vm != 0);
        const uint64_t ballB = __builtin_amdgcn_ballot_w64(sb.ok16 && (sb.gw32 & sb.gm) == sb.gm && sb.vm != 0);
        const uint32_t nA = (uint32_t)__builtin_popcountll(ballA), nB = (uint32_t)__builtin_popcountll(ballB);
        if (VG_DBG(p.dbg) & 1u) continue;
        enqueue(sa, ballA, run_head + run_n);
        run_n += nA;
        // the ring holds VG_RUNQ runs: make room for row B's (only rows dense in candidates ever need this)
        while (run_n + nB > VG_RUNQ) drain_step();
        enqueue(sb, ballB, run_head + run_n);
        run_n += nB;
        if (run_n >= VG_RUN_BATCH || req_n >= 32u) drain_step();
        while (run_n >= 3 * VG_RUN_BATCH || req_n >= 48u) drain_step();   // dense stretches: keep the rings short
    }
    // flush: until the rings are empty and the last step issued nothing (finishing a batch can re-queue)
    do {
        drain_step();
    } while (run_n != 0 || req_n != 0 || __builtin_amdgcn_ballot_w64(b_active) != 0);
    vm_wait_imm<0>();
}

// ------------------------------------------------------------------------------------------
// count27s_kernel: k = 27, graphs of <= 65 536 k-mers (BASELINE config 2, the bench) -- round 3.
//
// The same pipeline as count27_kernel<true, true> on a coarser grid (vgmi_device.h, VG_GRID12_*): 1 024-byte rows, 16 bytes
// (one dwordx4) per lane, ONE grid 12-mer per lane and row, candidate runs of 16 k-mers, four runs (64 lanes) per probe
// batch.  What is paid per grid position -- two neighbour exchanges instead of three, validity, canonical form, hash, filter
// word, ballot, enqueue -- is paid once per 16 bases instead of once per 12, the drain's lanes are all busy (4 x 16 = 64
// instead of 5 x 12 = 60), and no 32-bit multiply (quarter rate) is left in the loop.  The vector memory is scheduled by
// hand exactly as there (see the header of count27_kernel); fixed registers: v[116:119] / v[120:123] the two rows of a
// pair, v[124:125] the table word in flight, v126 the atomic's return value.
//
// A lane's 48-base window, base q = 0..47 at bits 2(47 - q) of W2:W1:W0: q 32..47 own chunk (W0), q 16..31 lane - 1 (W1),
// q 0..15 lane - 2 (W2).  Grid position g = q 31 (the last base of the previous lane's chunk):
//   grid 12-mer  q 20..31 = the low 24 bits of W1
//   run          q  5..46 (k-mers ending at q 31..46, i.e. g .. g + 15) = bits [2, 86)
// ------------------------------------------------------------------------------------------
#define VG_ROW27S 1024u
#define VG_RUN_BATCH_S 4u

// K < 27 (odd, 19 .. 25; round 5, path table only): a k-mer of fewer than 27 bases need not contain a grid 12-mer of a grid of 16, so
// the grid is 8 -- the SAME rows, loads, encode and neighbour exchange, but every lane looks at TWO grid positions per row, g = its
// own base 0 and its own base 8, and each owns the 8 k-mers that end at g .. g + 7 (all of them contain the 12-mer that ends at g:
// K - 12 >= 7).  A run is the K + 7 bases q[g - K + 1, g + 7] (<= 32 bases: two words), K - 12 in front of the 12-mer and 7 behind
// it; the path table lists a 12-mer's places K - 12 bases in front of it (build_ptable), a read that carries it reversed is
// compared at Tp - (2 K - 12) - place.  A lane of a row covers the ends 16 L .. 16 L + 15: every end inside the rows (tail27 = 4).
template <bool PT, uint32_t K = 27>   // PT: candidate runs are looked up in the path table (vgmi_ptable.hip); else k-mer by k-mer in the hash table
__global__ __launch_bounds__(1024) void count27s_kernel(RowParams p)
{
    // (even k = 20 .. 24 too: the windows of k bases are counted, as for odd k -- a k-mer that is its own reverse complement is never
    // emitted by the reference and never counted here: its start bit is not set in the path table, the hash-table fallback skips
    // it -- and seq_kernel<MODE_DEBIT> has taken back, ahead of this kernel, the few windows the reference's run counter suppresses)
    static_assert(K == 27 || (PT && K >= 19 && K <= 25), "count27s_kernel: k = 27, or the path-table form for k = 19 .. 25 (a run of k + 7 bases is two words)");
    constexpr uint32_t MASK_HI = (1u << (2 * K - 32)) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t nwaves = blockDim.x >> 6;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 lds_u4_rw;
    typedef __attribute__((address_space(3))) uint64_t lds_u64_rw;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t rings0 = VG_LUT27_BYTES + VG_GRID_LDS_WORDS * 4u;
    const uint32_t runs_base = rings0 + wave_u * (VG_RUNQ * 16u);
    const uint32_t req_base = rings0 + nwaves * (VG_RUNQ * 16u) + wave_u * (VG_REQ * 8u);
    stage_lut27(tid, blockDim.x);
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.table.grid);
        uint4* dst = reinterpret_cast<uint4*>(smem + VG_LUT27_BYTES);
        for (uint32_t i = tid; i < VG_GRID_LDS_WORDS / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    unsigned long long* const slots8 = p.table.slots8;
    uint32_t* const counts = p.table.counts;
    const uint64_t cap_mask = p.table.cap_mask;

    // rows [0, row_end) are complete 1 024-byte rows, row_end even: a wave walks a contiguous range of row PAIRS
    const uint64_t total_pairs = p.n_bytes_dev ? *p.n_bytes_dev / (2 * VG_ROW27S) : p.row_end >> 1;
    const uint64_t total_waves = (uint64_t)gridDim.x * nwaves;
    const uint64_t ppw = (total_pairs + total_waves - 1) / total_waves;
    const uint64_t gw = (uint64_t)blockIdx.x * nwaves + wave_u;
    const uint64_t r0v = gw * ppw;
    const uint64_t r1v = r0v + ppw < total_pairs ? r0v + ppw : total_pairs;
    const uint64_t r0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r0v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r0v);
    const uint64_t r1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r1v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r1v);
    if (r0 >= r1) return;
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));
    const uint32_t lane_off = lane * 16u;
    const uint8_t* const bases = p.bases;

    // previous row's words, already rotated by 1 / 2 lanes (lanes 0 and 1 take them); nothing lies in front of the block
    uint32_t pr1_be = 0, pr2_be = 0, pr1_inv = 0xFFFFu, pr2_inv = 0xFFFFu;
    const uint64_t rs = r0 > 0 ? r0 - 1 : r0;   // the pair before the range is walked first: it only provides the halo

    uint32_t n_after_row = 0, n_after_slot = 2;
    uint64_t b_canon = 0, b_slot = 0;
    bool b_active = false;
    uint64_t p_slot = 0;
    bool p_bumped = false;

    const uint32_t my_run = lane >> 4, my_win = lane & 15u;
    const uint32_t my_sh = 2 * (15 - my_win);
    uint32_t run_head = 0, run_n = 0, req_head = 0, req_n = 0;
    auto ring_slot = [](uint32_t pos) -> uint32_t {  // pos < 2 * VG_RUNQ
        const uint32_t w = pos - VG_RUNQ;
        return w < pos ? w : pos;
    };

    auto probe_issue = [&](bool act, uint64_t kmer, uint64_t dist) __attribute__((always_inline)) {
        const uint64_t rc = vg_revcomp(kmer, K);
        const uint64_t canon = kmer < rc ? kmer : rc;   // idempotent for re-queued (already canonical) entries
        b_canon = canon | (dist << 54);
        b_slot = (vg_thash(canon) + dist) & cap_mask;
        b_active = act;
        if (act) vms_load_slot(&slots8[b_slot]);
        ++n_after_row;
        n_after_slot = 0;
    };
    // ================= path-table drain (PT): one lane per run =================
    // Once per <= 64 queued runs, every lane takes ONE run through three phases, an iteration of the row loop apart (each
    // phase's loads have that long to arrive):
    //   1  the run leaves the ring, its index bucket is asked for
    //   2  exact compare on the 12-mer: a run that is not in the index is dropped (the filter's false positives end here);
    //      for each of its (up to two) places: 16 bytes of the unitig sequence S, the words that hold the 16 k-mer-start
    //      bits and the 16 saturation bits.  A read that carries the 12-mer reversed needs no turning round: S holds every
    //      unitig in both orientations, mirrored, so its run is compared at Tp - 42 - place
    //   3  run XOR sequence: the first mismatch on either side of the 12-mer bounds the windows that equal the unitig's k-mers
    //      (two find-first-bit instructions give all 16 answers); AND the start bits AND the run's own validity bits = the hits.
    //      Hits whose saturation bit is set are done (the steady state of a deep sample); the others fetch their slot and bump
    //      its counter, bit by bit.
    constexpr uint32_t RQ = 112;      // run ring: all of the wavefront's 1 792 bytes (a round takes up to 64 runs, rows keep adding)
    const uint32_t rq_base = rings0 + wave_u * (VG_RUNQ * 16u + VG_REQ * 8u);
    auto rq_wrap = [](uint32_t pos) -> uint32_t { return pos >= RQ ? pos - RQ : pos; };      // pos < 2 * RQ
    const unsigned long long* const pt_index = p.table.pt.index;
    const uint32_t* const ptS = p.table.pt.S;
    const uint32_t* const ptVB = p.table.pt.VB;
    uint32_t* const ptSB = p.table.pt.SB;
    const uint32_t* const ptSLOT = p.table.pt.SLOT;
    const uint32_t pt_Tp = p.table.pt.Tp, pt_bshift = 32u - p.table.pt.bucket_log2;
    const uint32_t l1_min = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.l1_min);
    uint32_t n_after_idx = 8, n_after_seq = 8;            // hot operations issued after the index load / after the last load of phase 2
    u32x4 l1_run = {0u, 0u, 0u, 0u};                      // the run this lane works on
    uint32_t l1_n = 0, l1_phase = 0;                      // wave-uniform: runs in the round, 0 idle / 1 bucket in flight / 2 sequence in flight
    uint32_t pl[4] = {0u, 0u, 0u, 0u};                    // places (0: none)
    bool l1_slow = false;
    auto hot = [&](uint32_t k) __attribute__((always_inline)) {
        n_after_row += k;
        n_after_idx += k;
        n_after_seq += k;
    };
    auto slow_count = [&](uint32_t klo, uint32_t khi) __attribute__((always_inline)) {
        // a window of a run whose 12-mer the index does not cover: through the hash table, on the spot (rare)
        const uint64_t kmer = (uint64_t)khi << 32 | klo, rc = vg_revcomp(kmer, K);
        if (!(K & 1u) && kmer == rc) return;      // src/kmer.cpp:134
        const uint64_t canon = kmer < rc ? kmer : rc;
        uint64_t sl = vg_thash(canon) & cap_mask;
        for (;;) {
            const uint4 v = vm_load_slot_sync<true>(&slots8[sl]);
            const uint64_t c = ((uint64_t)v.y << 32) | v.x;
            if (c == VG_EMPTY) break;
            if ((c & VG_SLOT_KMER_MASK) == canon) {
                if (!(c & VG_SLOT_SAT) && vm_atomic_inc_sync(&counts[sl], one) == 254u) {
                    vm_atomic_or_sync(reinterpret_cast<uint32_t*>(&slots8[sl]) + 1, (uint32_t)(VG_SLOT_SAT >> 32));
                    vm_store_byte_sync(p.table.sat_dirty + (sl >> VG_SAT_REGION_LOG2), 1u);
                    // ... and at both of the k-mer's places in the path table, as bump_all does: later hits through the table skip their atomic
                    const uint32_t pos = vm_load_dword_sync(p.table.pt.PLACE + sl);
                    if (pos != 0u) {
                        const uint32_t mir = pt_Tp - K - pos;
                        vm_atomic_or_sync(ptSB + (pos >> 5), 1u << (pos & 31u));
                        vm_atomic_or_sync(ptSB + (mir >> 5), 1u << (mir & 31u));
                    }
                }
                break;
            }
            if (!(c & VG_SLOT_CHAIN)) break;
            sl = (sl + 1) & cap_mask;
        }
    };
    // phase 1
    auto resolve_issue = [&]() __attribute__((always_inline)) {
        l1_n = run_n < 64u ? run_n : 64u;
        if (lane < l1_n && !(VG_DBG(p.dbg) & 2048u)) {      // 2048: ablation (wrong counts): the runs are popped and forgotten
            l1_run = *reinterpret_cast<lds_u4_rw*>((uintptr_t)(rq_base + rq_wrap(run_head + lane) * 16u));
            vmp_load_index(pt_index + ((uint64_t)(vg_idx_hash(l1_run.w & 0xFFFFFFu) >> pt_bshift) << 2));
        }
        run_head = rq_wrap(run_head + l1_n);
        run_n -= l1_n;
        hot(2);
        n_after_idx = 0;
        l1_phase = 1;
        __builtin_amdgcn_wave_barrier();
    };
    // phase 2
    auto resolve_places = [&]() __attribute__((always_inline)) {
        vm_wait4(n_after_idx);
        const uint4 e0 = vmp_index_value<0>(), e1 = vmp_index_value<1>();
        const uint32_t cx = l1_run.w & 0xFFFFFFu;
        const bool act = lane < l1_n && !(VG_DBG(p.dbg) & (1024u | 2048u));
        const bool as_is = ((l1_run.w >> 24) & 1u) != 0;
        // entry: x[0:24) the 12-mer, places 0 and 1 in x[24:32) : y[0:11) and y[11:30), y[30] "some 12-mer found no entry in this
        // bucket" (first entry only); places 2 and 3 in z[0:19) and z[19:32) : w[0:6), w[6] "more than four places"
        const bool m0 = ((e0.x ^ cx) & 0xFFFFFFu) == 0 && (((e0.x >> 24) | (e0.y << 8)) & 0x7FFFFu) != 0;
        const bool m1 = ((e1.x ^ cx) & 0xFFFFFFu) == 0 && (((e1.x >> 24) | (e1.y << 8)) & 0x7FFFFu) != 0;
        const uint4 e = m0 ? e0 : e1;
        const bool found = m0 || m1;
        const uint32_t q0 = ((e.x >> 24) | (e.y << 8)) & 0x7FFFFu, q1 = (e.y >> 11) & 0x7FFFFu;
        const uint32_t q2 = e.z & 0x7FFFFu, q3 = ((e.z >> 19) | (e.w << 13)) & 0x7FFFFu;
        const bool more = ((e.w >> 6) & 1u) != 0;
        // more places than the entry holds, or a 12-mer that found no entry in its bucket: the hash table decides, window by window
        l1_slow = act && (found ? more : ((e0.y >> 30) & 1u) != 0);
        const bool ok = act && found && !more;
        const uint32_t flip = pt_Tp - (K == 27 ? 42u : 2u * K - 12u);      // the run's length less the lead's asymmetry (build_ptable)
        pl[0] = ok ? (as_is ? q0 : flip - q0) : 0u;
        pl[1] = ok && q1 != 0 ? (as_is ? q1 : flip - q1) : 0u;
        pl[2] = ok && q2 != 0 ? (as_is ? q2 : flip - q2) : 0u;
        pl[3] = ok && q3 != 0 ? (as_is ? q3 : flip - q3) : 0u;
        if (__builtin_amdgcn_ballot_w64(pl[0] != 0)) {
            if (pl[0] != 0) vmp_load_place<0>(ptS + (pl[0] >> 4), ptVB + (pl[0] >> 5), ptSB + (pl[0] >> 5));
            hot(3);
        }
        if (__builtin_amdgcn_ballot_w64(pl[1] != 0)) {
            if (pl[1] != 0) vmp_load_place<1>(ptS + (pl[1] >> 4), ptVB + (pl[1] >> 5), ptSB + (pl[1] >> 5));
            hot(3);
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pl[2] != 0) != 0, 0)) {     // neighbouring sites: a third and a fourth allele combination
            if (pl[2] != 0) vmp_load_place<2>(ptS + (pl[2] >> 4), ptVB + (pl[2] >> 5), ptSB + (pl[2] >> 5));
            hot(3);
            if (__builtin_amdgcn_ballot_w64(pl[3] != 0)) {
                if (pl[3] != 0) vmp_load_place<3>(ptS + (pl[3] >> 4), ptVB + (pl[3] >> 5), ptSB + (pl[3] >> 5));
                hot(3);
            }
        }
        n_after_seq = 0;
        l1_phase = 2;
    };
    // phase 3, per place: the windows of the run that equal the k-mers starting at place .. place + 15, less the saturated ones
    auto windows_at = [&](uint32_t place, const uint4 sq, const uint4 bits, uint32_t vm16, uint32_t& hits) __attribute__((always_inline)) -> uint32_t {
        if constexpr (K != 27) {
            // the K + 7 bases from `place` on, aligned like the run's 2 (K + 7) bits (base r at bits 2 (K + 6 - r)): the 128-bit
            // big-endian stream sq.x : y : z : w shifted right by 128 - 2 (K + 7) - 2 (place mod 16) = 34 .. 76
            const uint32_t sh = 128u - 2u * (K + 7u) - 2u * (place & 15u);
            const bool far = sh >= 64u;
            const uint32_t x0 = funnel(far ? sq.x : sq.y, far ? sq.y : sq.z, sh & 31u);
            const uint32_t x1 = funnel(far ? 0u : sq.x, far ? sq.x : sq.y, sh & 31u);
            const uint32_t d0 = x0 ^ l1_run.x, d1 = (x1 ^ l1_run.y) & ((K + 7u == 32u) ? 0xFFFFFFFFu : (1u << ((2u * (K + 7u) - 32u) & 31u)) - 1u);
            // the 7 bases behind the 12-mer are bits [0, 14): the mismatch nearest to it (highest bit) ends the windows
            const uint32_t tr = (d0 | (d0 >> 1)) & 0x1555u;
            const uint32_t mask_r = tr ? (1u << (7u - ((31u - (uint32_t)__builtin_clz(tr)) >> 1))) - 1u : 0xFFu;
            // the K - 12 bases in front of it are bits [38, 2 K + 14): the mismatch nearest to it (lowest bit) starts them
            const uint32_t fl = d1 >> 6;
            const uint32_t tl = (fl | (fl >> 1)) & 0x1555555u;
            const uint32_t mask_l = tl ? ~((2u << (K - 13u - ((uint32_t)__builtin_ctz(tl) >> 1))) - 1u) & 0xFFu : 0xFFu;
            const uint32_t starts = funnel(bits.y, bits.x, place & 31u) & 0xFFu;
            const uint32_t sats = funnel(bits.w, bits.z, place & 31u) & 0xFFu;
            hits = mask_l & mask_r & starts & vm16;
            return hits & ~sats;
        }
        // the 42 bases from `place` on, aligned like the run's 84 bits: the 128-bit big-endian stream sq.x : y : z : w shifted
        // right by 44 - 2 (place mod 16)
        const uint32_t sh = 44u - 2u * (place & 15u);
        const bool far = sh >= 32u;
        const uint32_t x0 = funnel(far ? sq.y : sq.z, far ? sq.z : sq.w, sh & 31u);
        const uint32_t x1 = funnel(far ? sq.x : sq.y, far ? sq.y : sq.z, sh & 31u);
        const uint32_t x2 = funnel(far ? 0u : sq.x, far ? sq.x : sq.y, sh & 31u);
        const uint32_t d0 = x0 ^ l1_run.x, d1 = x1 ^ l1_run.y, d2 = (x2 ^ l1_run.z) & 0xFFFFFu;
        // bases r27 .. r41 (behind the 12-mer) are bits [0, 30): the mismatch nearest to the 12-mer (highest bit) ends the windows
        const uint32_t tr = (d0 | (d0 >> 1)) & 0x15555555u;
        const uint32_t mask_r = tr ? (1u << (15u - ((31u - (uint32_t)__builtin_clz(tr)) >> 1))) - 1u : 0xFFFFu;
        // bases r0 .. r14 (in front of it) are bits [54, 84): the mismatch nearest to the 12-mer (lowest bit) starts them
        const uint32_t fl = funnel(d2, d1, 22) & 0x3FFFFFFFu;
        const uint32_t tl = (fl | (fl >> 1)) & 0x15555555u;
        const uint32_t mask_l = tl ? ~((1u << (15u - ((uint32_t)__builtin_ctz(tl) >> 1))) - 1u) & 0xFFFFu : 0xFFFFu;
        const uint32_t starts = funnel(bits.y, bits.x, place & 31u) & 0xFFFFu;
        const uint32_t sats = funnel(bits.w, bits.z, place & 31u) & 0xFFFFu;
        hits = mask_l & mask_r & starts & vm16;
        return hits & ~sats;
    };
    auto bump_all = [&](uint32_t todo, uint32_t place) __attribute__((always_inline)) {
        // the unsaturated hits of this lane's run, up to four windows per round (every lane with a hit left takes part): their
        // slots in one round trip, their counters in a second.  A lane with fewer hits left pads the round by adding 0 to the
        // counter of its first window (its own address: a shared dummy would serialise every padded lane of the device).
#pragma unroll 1
        while (__builtin_amdgcn_ballot_w64(todo != 0) != 0) {
            if (todo != 0) {
                uint32_t pos[4], sl[4], old[4];
                bool live[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    live[q] = todo != 0;
                    pos[q] = place + (live[q] ? (uint32_t)__builtin_ctz(todo) : 0u);
                    todo &= todo - 1u;      // 0 stays 0
                }
                vm_load_dword4_sync(ptSLOT + pos[0], ptSLOT + pos[1], ptSLOT + pos[2], ptSLOT + pos[3], sl);
                const uint32_t add[4] = {1u, live[1] ? 1u : 0u, live[2] ? 1u : 0u, live[3] ? 1u : 0u};
                vm_atomic_add4_sync(&counts[sl[0]], &counts[live[1] ? sl[1] : sl[0]], &counts[live[2] ? sl[2] : sl[0]], &counts[live[3] ? sl[3] : sl[0]], add, old);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (live[q] && old[q] == 254u) {
                        // this increment took the counter to the clamp: flag the k-mer in the hash table (the generic kernels
                        // and the slow path look there) and at both of its places in the path table
                        vm_atomic_or_sync(reinterpret_cast<uint32_t*>(&slots8[sl[q]]) + 1, (uint32_t)(VG_SLOT_SAT >> 32));
                        vm_store_byte_sync(p.table.sat_dirty + (sl[q] >> VG_SAT_REGION_LOG2), 1u);
                        const uint32_t mir = pt_Tp - K - pos[q];
                        vm_atomic_or_sync(ptSB + (pos[q] >> 5), 1u << (pos[q] & 31u));
                        vm_atomic_or_sync(ptSB + (mir >> 5), 1u << (mir & 31u));
                    }
                }
            }
        }
    };
    auto resolve_compare = [&]() __attribute__((always_inline)) {
        vm_wait4(n_after_seq);
        const uint32_t vm16 = K == 27 ? (l1_run.z >> 20) | ((l1_run.w >> 28) << 12) : l1_run.z & 0xFFu;      // (K < 27: 8 windows, kept in z)
        uint32_t hits = 0, todo[4] = {0u, 0u, 0u, 0u};
        if (pl[0] != 0) todo[0] = windows_at(pl[0], vmp_seq_value<0>(), vmp_bits_value<0>(), vm16, hits);
        if (pl[1] != 0) todo[1] = windows_at(pl[1], vmp_seq_value<1>(), vmp_bits_value<1>(), vm16, hits);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pl[2] != 0) != 0, 0)) {
            if (pl[2] != 0) todo[2] = windows_at(pl[2], vmp_seq_value<2>(), vmp_bits_value<2>(), vm16, hits);
            if (pl[3] != 0) todo[3] = windows_at(pl[3], vmp_seq_value<3>(), vmp_bits_value<3>(), vm16, hits);
        }
        if (VG_DBG(p.dbg) & 512u) todo[0] = todo[1] = todo[2] = todo[3] = 0;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64((todo[0] | todo[1] | todo[2] | todo[3]) != 0) != 0, 0)) {
            bump_all(todo[0], pl[0]);
            bump_all(todo[1], pl[1]);
            if (__builtin_amdgcn_ballot_w64((todo[2] | todo[3]) != 0) != 0) {
                bump_all(todo[2], pl[2]);
                bump_all(todo[3], pl[3]);
            }
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(l1_slow && !(VG_DBG(p.dbg) & 4096u)) != 0, 0)) {     // 4096: ablation: no slow path
            // runs the index does not cover, one after the other, their 16 windows side by side on lanes 0..15
            uint64_t left = __builtin_amdgcn_ballot_w64(l1_slow);
            while (left) {
                const uint32_t src = (uint32_t)__builtin_ctzll(left);
                left &= left - 1;
                const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)l1_run.x, (int)src), ry = (uint32_t)__builtin_amdgcn_readlane((int)l1_run.y, (int)src);
                const uint32_t rz = (uint32_t)__builtin_amdgcn_readlane((int)l1_run.z, (int)src), rw = (uint32_t)__builtin_amdgcn_readlane((int)l1_run.w, (int)src);
                if constexpr (K != 27) {
                    // (window j = the K bases from run base j on: bits [2 (7 - j), + 2 K) of the run's two words)
                    if (lane < 8u && (((rz & 0xFFu) >> lane) & 1u)) {
                        const uint32_t sh2 = 2u * (7u - lane);
                        slow_count(funnel(ry, rx, sh2), (ry >> sh2) & MASK_HI);
                    }
                    continue;
                }
                const uint32_t v16 = (rz >> 20) | ((rw >> 28) << 12);
                if (lane < 16u && ((v16 >> lane) & 1u)) {
                    const uint32_t sh2 = 2u * (15u - lane);
                    slow_count(funnel(ry, rx, sh2), funnel(rz, ry, sh2) & MASK_HI);
                }
            }
        }
        l1_phase = 0;
    };
    auto l1_advance = [&]() __attribute__((always_inline)) {
        if (l1_phase == 2) resolve_compare();
        else if (l1_phase == 1) resolve_places();
    };

    auto drain_hash = [&]() __attribute__((always_inline)) {
        vm_wait(n_after_slot);
        {
            // the previous step's atomics have returned (issued before the loads just waited for): the one increment that
            // takes a counter from 254 to 255 flags the slot (later hits skip their atomic, as the reference skips its
            // increment) and marks its region for the per-sample reset
            const uint32_t old = vms_atomic_old();
            const bool sat = p_bumped && old == 254u;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(sat) != 0, 0)) {
                if (sat) {
                    vm_atomic_or_sync(reinterpret_cast<uint32_t*>(&slots8[p_slot]) + 1, (uint32_t)(VG_SLOT_SAT >> 32));
                    vm_store_byte_sync(p.table.sat_dirty + (p_slot >> VG_SAT_REGION_LOG2), 1u);
                }
            }
        }
        const uint2 tv = vms_slot_value();
        bool again = false;
        uint32_t* bump = nullptr;
        if (b_active) {
            const uint64_t c = ((uint64_t)tv.y << 32) | tv.x;
            const uint64_t canon = b_canon & VG_Q_KMER_MASK;
            if (c != VG_EMPTY && (c & VG_SLOT_KMER_MASK) == canon) {
                if (!(c & VG_SLOT_SAT)) bump = &counts[b_slot];
            } else if (c != VG_EMPTY && (c & VG_SLOT_CHAIN)) {
                again = !(VG_DBG(p.dbg) & 64u);
            }
        }
        const uint64_t ball = __builtin_amdgcn_ballot_w64(again);
        if (ball != 0) {
            const uint32_t n = (uint32_t)__builtin_popcountll(ball);
            const bool fits = req_n + n <= VG_REQ && (b_canon >> 54) < 1023u;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(again && !fits) == 0, 1)) {
                if (again) {
                    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32),
                                         __builtin_amdgcn_mbcnt_lo((uint32_t)ball, req_head + req_n));
                    *reinterpret_cast<lds_u64_rw*>((uintptr_t)(req_base + (pos & (VG_REQ - 1)) * 8u)) = b_canon + (1ULL << 54);
                }
                req_n += n;
            } else if (again) {
                // ring full (or probe distance field exhausted): chase the chain right here, synchronously
                const uint64_t canon = b_canon & VG_Q_KMER_MASK;
                uint64_t sl = b_slot;
                for (;;) {
                    sl = (sl + 1) & cap_mask;
                    const uint4 v = vm_load_slot_sync<true>(&slots8[sl]);
                    const uint64_t c = ((uint64_t)v.y << 32) | v.x;
                    if (c != VG_EMPTY && (c & VG_SLOT_KMER_MASK) == canon) {
                        if (!(c & VG_SLOT_SAT)) { bump = &counts[sl]; b_slot = sl; }
                        break;
                    }
                    if (c == VG_EMPTY || !(c & VG_SLOT_CHAIN)) break;
                }
            }
        }
        p_bumped = bump != nullptr;
        p_slot = b_slot;
        if (__builtin_amdgcn_ballot_w64(bump != nullptr)) {   // wave-uniform: the atomic is issued iff some lane hit
            if (bump) vms_atomic_inc(bump, one);
            ++n_after_row;
        }
        __builtin_amdgcn_wave_barrier();
        const bool do_req = req_n >= 32u || run_n == 0;
        const uint32_t take_req = do_req ? (req_n < 64u ? req_n : 64u) : 0u;   // may be 0 (final flush)
        const uint32_t take_run = do_req ? 0u : (run_n < VG_RUN_BATCH_S ? run_n : VG_RUN_BATCH_S);
        const uint32_t rq_head = req_head, rn_head = run_head;
        req_head += take_req;
        req_n -= take_req;
        run_head += take_run;
        if (run_head >= VG_RUNQ) run_head -= VG_RUNQ;
        run_n -= take_run;
        if (do_req) {
            const bool have = lane < take_req;
            uint64_t e = 0;
            if (have) e = *reinterpret_cast<lds_u64_rw*>((uintptr_t)(req_base + ((rq_head + lane) & (VG_REQ - 1)) * 8u));
            if (take_req) probe_issue(have, e & VG_Q_KMER_MASK, e >> 54);
            else b_active = false;   // an all-idle batch, no load
        } else {
            const bool have = my_run < take_run;
            u32x4 e = {0u, 0u, 0u, 0u};
            if (have) e = *reinterpret_cast<lds_u4_rw*>((uintptr_t)(runs_base + ring_slot(rn_head + my_run) * 16u));
            // k-mer ending at window my_win of the run: bits [2(15 - win), +54) of the 84 run bits
            const uint32_t lo = funnel(e.y, e.x, my_sh);
            const uint32_t hi = funnel(e.z, e.y, my_sh) & MASK_HI;
            probe_issue(have && ((e.w >> my_win) & 1u) && (!(VG_DBG(p.dbg) & 32u) || my_win == (__builtin_ctz(e.w) & 15u)),
                        ((uint64_t)hi << 32) | lo, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };

    auto drain_step = [&]() __attribute__((always_inline)) { drain_hash(); };
    // path table: the run ring is emptied by level 1 in one go (after the round before it has been finished)
    auto make_room = [&]() __attribute__((always_inline)) {
        while (l1_phase != 0) l1_advance();
        resolve_issue();
    };

    // scan of one row once its words and its neighbours' are known: validity of the 16 windows, grid probe
    struct RowScan { uint32_t W0, W1, W2, vm, gm, gw32, cm; };   // cm: canonical 12-mer | (the read carries it as it stands) << 24
    auto scan_probe = [&](uint32_t be, uint32_t inv, uint32_t be1, uint32_t be2, uint32_t i1, uint32_t i2) __attribute__((always_inline)) -> RowScan {
        RowScan r;
        r.W0 = be;
        r.W1 = be1;
        r.W2 = be2;
        // non-base bits.  B bit i = base q 5 + i (i = 0..26, the span of the k-mer ending at g): the k-mer ending at g + j is
        // spoilt by B iff B >> j != 0, and by the own chunk iff one of own bases 0..j-1 is a non-base (x | -x = every bit
        // from the lowest set one up).  A non-base inside the 12-mer itself (B bits 15..26) spoils all 16 windows: vm = 0.
        const uint32_t B = (i2 >> 5) | (i1 << 11);
        const uint32_t a = (inv << 1) & 0xFFFFu;
        const uint32_t bad_b = B ? (0xFFFFFFFFu >> __builtin_clz(B)) : 0u;
        r.vm = ~(a | (0u - a) | bad_b) & 0xFFFFu;
        // vg_grid12_probe (vgmi_device.h), with the canonical form kept for the run's entry
        const uint32_t mer = be1 & 0xFFFFFFu, rcm = vg_revcomp12(mer);
        const uint32_t cm = mer < rcm ? mer : rcm;
        r.cm = cm | (mer <= rcm ? 1u << 24 : 0u);
        const uint32_t h1 = vg_mul24(cm, 0x9E3779u), h2 = vg_mul24(cm, 0x85EBCBu);
        r.gm = (1u << (h2 >> 27)) | (1u << ((h2 >> 22) & 31u)) | (1u << ((h2 >> 17) & 31u));
        r.gw32 = *reinterpret_cast<lds_u32*>((uintptr_t)(((h1 >> (32 - VG_GRID_LDS_WORDS_LOG2)) << 2) + VG_LUT27_BYTES));
        return r;
    };
    auto enqueue = [&](const RowScan& r, uint64_t ball, uint32_t tail) __attribute__((always_inline)) {
        if (__builtin_amdgcn_inverse_ballot_w64(ball)) {
            const uint32_t d0 = funnel(r.W1, r.W0, 2);
            const uint32_t d1 = funnel(r.W2, r.W1, 2);
            const uint32_t d2 = r.W2 >> 2;
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, tail));
            if (PT)   // run bits [0, 84) | validity bits 0..11 in z[20:32), 12..15 in w[28:32) | canonical 12-mer and its orientation in w[0:25)
                *reinterpret_cast<lds_u4_rw*>((uintptr_t)(rq_base + rq_wrap(pos) * 16u)) =
                    u32x4{d0, d1, (d2 & 0xFFFFFu) | (r.vm << 20), r.cm | ((r.vm >> 12) << 28)};
            else
                *reinterpret_cast<lds_u4_rw*>((uintptr_t)(runs_base + ring_slot(pos) * 16u)) = u32x4{d0, d1, d2, r.vm};
        }
    };
    // K < 27: grid position H (0: g = the lane's own base 0, 1: its base 8) of one row.  g = q 32 + 8 H of the 48-base window; B bit i =
    // base q g - K + 1 + i is no base (the span of the k-mer ending at g), a bit j = base g + j is none (j = 1 .. 7).  H is
    // wave-uniform: every shift amount is a scalar.
    auto scan_probe_k = [&](uint32_t be, uint32_t inv, uint32_t be1, uint32_t be2, uint32_t i1, uint32_t i2, uint32_t H) __attribute__((always_inline)) -> RowScan {
        RowScan r;
        r.W0 = be;
        r.W1 = be1;
        r.W2 = be2;
        const uint32_t g = 32u + 8u * H;
        const uint64_t I = (uint64_t)i2 | (uint64_t)i1 << 16 | (uint64_t)inv << 32;
        const uint32_t B = (uint32_t)(I >> (g + 1u - K)) & ((1u << K) - 1u);
        const uint32_t a = (uint32_t)(I >> g) & 0xFEu;
        const uint32_t bad_b = B ? (0xFFFFFFFFu >> __builtin_clz(B)) : 0u;
        r.vm = ~(a | (0u - a) | bad_b) & 0xFFu;
        const uint32_t mer = funnel(be1, be, 30u - 16u * H) & 0xFFFFFFu, rcm = vg_revcomp12(mer);
        const uint32_t cm = mer < rcm ? mer : rcm;
        r.cm = cm | (mer <= rcm ? 1u << 24 : 0u);
        const uint32_t h1 = vg_mul24(cm, 0x9E3779u), h2 = vg_mul24(cm, 0x85EBCBu);
        r.gm = (1u << (h2 >> 27)) | (1u << ((h2 >> 22) & 31u)) | (1u << ((h2 >> 17) & 31u));
        r.gw32 = *reinterpret_cast<lds_u32*>((uintptr_t)(((h1 >> (32 - VG_GRID_LDS_WORDS_LOG2)) << 2) + VG_LUT27_BYTES));
        return r;
    };
    // ... its run: the 2 (K + 7) bits of q[g - K + 1, g + 7] | validity bits in z | canonical 12-mer and its orientation in w[0:25)
    auto enqueue_k = [&](const RowScan& r, uint64_t ball, uint32_t tail, uint32_t H) __attribute__((always_inline)) {
        if (__builtin_amdgcn_inverse_ballot_w64(ball)) {
            const uint32_t sh = 16u - 16u * H;
            const uint32_t d0 = funnel(r.W1, r.W0, sh);
            const uint32_t d1 = funnel(r.W2, r.W1, sh) & ((K + 7u == 32u) ? 0xFFFFFFFFu : (1u << ((2u * (K + 7u) - 32u) & 31u)) - 1u);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, tail));
            *reinterpret_cast<lds_u4_rw*>((uintptr_t)(rq_base + rq_wrap(pos) * 16u)) = u32x4{d0, d1, r.vm, r.cm};
        }
    };
    auto empty_read_check = [&](uint32_t adj, const uint8_t* row) __attribute__((always_inline)) {
        // reference: assert(len > 0), src/kmer.cpp:124 -- two adjacent non-bases are necessary; exact test on the bytes
        if (adj) {
            const uint64_t base_off = (uint64_t)(row - bases) + lane_off;
            for (uint32_t t = 0; t < 16; ++t) {
                if (!((adj >> t) & 1u)) continue;
                const uint64_t o = base_off + t;
                if (vm_load_byte_sync(bases + o) == '\n' && (o == 0 || vm_load_byte_sync(bases + o - 1) == '\n'))
                    vm_atomic_or_sync(p.status, 1u);
            }
        }
    };

    const uint32_t n_it = (uint32_t)(r1 - rs), n_warm = (uint32_t)(r0 - rs);
    const uint8_t* rowp = bases + rs * (2 * VG_ROW27S);
    vms_load_row<0>(rowp + lane_off);
    vms_load_row<1>(rowp + VG_ROW27S + lane_off);
    for (uint32_t it = 0; it < n_it; ++it) {
        // both rows have landed once at most n_after_row younger operations are outstanding
        if (PT) vm_wait4(n_after_row);
        else vm_wait(n_after_row);
        const Addr4 a0 = luts_addr4<0, 0>(one), a1 = luts_addr4<0, 1>(one), a2 = luts_addr4<0, 2>(one), a3 = luts_addr4<0, 3>(one);
        const Addr4 c0 = luts_addr4<1, 0>(one), c1 = luts_addr4<1, 1>(one), c2 = luts_addr4<1, 2>(one), c3 = luts_addr4<1, 3>(one);
        const uint8_t* const cur = rowp;
        if (it + 1 < n_it) rowp += 2 * VG_ROW27S;
        vms_load_row<0>(rowp + lane_off);            // prefetch (the last iteration re-reads its own pair)
        vms_load_row<1>(rowp + VG_ROW27S + lane_off);
        n_after_row = 0;
        n_after_slot += 2;
        if (PT) {
            n_after_idx += 2;
            n_after_seq += 2;
        } else if (run_n >= VG_RUN_BATCH_S || req_n >= 32u) drain_step();

        // own 16 bases of each row: be = 32 bits, first base most significant; inv bit t = base t is not a base.
        // LUT sets (stage_lut27): A puts a dword's four non-base flags at bits 8..11, B at bits 12..15
        const uint32_t g0 = encode4<0>(a0), g1 = encode4<1>(a1), g2 = encode4<0>(a2), g3 = encode4<1>(a3);
        const uint32_t h0 = encode4<0>(c0), h1 = encode4<1>(c1), h2 = encode4<0>(c2), h3 = encode4<1>(c3);
        const uint32_t beA = __builtin_amdgcn_perm(__builtin_amdgcn_perm(g0, g1, 0x0c0c0400u), __builtin_amdgcn_perm(g2, g3, 0x0c0c0400u), 0x05040100u);
        const uint32_t invA = (((g0 | g1) >> 8) & 0xFFu) | ((g2 | g3) & 0xFF00u);
        const uint32_t beB = __builtin_amdgcn_perm(__builtin_amdgcn_perm(h0, h1, 0x0c0c0400u), __builtin_amdgcn_perm(h2, h3, 0x0c0c0400u), 0x05040100u);
        const uint32_t invB = (((h0 | h1) >> 8) & 0xFFu) | ((h2 | h3) & 0xFF00u);
        auto ror1 = [](uint32_t v) __attribute__((always_inline)) -> uint32_t {
            return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x13C /* wave_ror:1: every lane has a source, no "old" value to set up */, 0xF, 0xF, false);
        };
        const uint32_t a1_be = ror1(beA), a1_inv = ror1(invA), b1_be = ror1(beB), b1_inv = ror1(invB);
        const uint32_t a2_be = ror1(a1_be), a2_inv = ror1(a1_inv), b2_be = ror1(b1_be), b2_inv = ror1(b1_inv);
        // neighbours: lanes l-1, l-2 of the same row; the first lanes take the tail of the row before (row A: second row of
        // the previous pair, row B: row A -- the rotates wrap, so lanes 0, 1 of a*_ hold exactly that tail)
        const uint32_t beA1 = lane >= 1 ? a1_be : pr1_be, beA2 = lane >= 2 ? a2_be : pr2_be;
        const uint32_t iA1 = lane >= 1 ? a1_inv : pr1_inv, iA2 = lane >= 2 ? a2_inv : pr2_inv;
        const uint32_t beB1 = lane >= 1 ? b1_be : a1_be, beB2 = lane >= 2 ? b2_be : a2_be;
        const uint32_t iB1 = lane >= 1 ? b1_inv : a1_inv, iB2 = lane >= 2 ? b2_inv : a2_inv;
        pr1_be = b1_be; pr2_be = b2_be; pr1_inv = b1_inv; pr2_inv = b2_inv;
        if (it < n_warm) continue;

        {   // empty-read check (rare path)
            const uint32_t adjA = invA & ((invA << 1) | (iA1 >> 15));
            const uint32_t adjB = invB & ((invB << 1) | (iB1 >> 15));
            if (__builtin_expect(__ballot(((adjA | adjB) & 0xFFFFu) != 0) != 0, 0)) {
                empty_read_check(adjA & 0xFFFFu, cur);
                empty_read_check(adjB & 0xFFFFu, cur + VG_ROW27S);
            }
        }

        if constexpr (K != 27) {
            // the four grid positions of the pair, one after the other (one copy of the code: h is wave-uniform)
#pragma unroll 1
            for (uint32_t h = 0; h < 4u; ++h) {
                const bool second = h >= 2u;
                const RowScan sc = scan_probe_k(second ? beB : beA, second ? invB : invA, second ? beB1 : beA1, second ? beB2 : beA2,
                                                second ? iB1 : iA1, second ? iB2 : iA2, h & 1u);
                const uint64_t ball = __builtin_amdgcn_ballot_w64((sc.gw32 & sc.gm) == sc.gm && sc.vm != 0);
                if (VG_DBG(p.dbg) & 1u) continue;
                const uint32_t nh = (uint32_t)__builtin_popcountll(ball);
                if (run_n + nh > RQ) make_room();
                enqueue_k(sc, ball, run_head + run_n, h & 1u);
                run_n += nh;
            }
            l1_advance();
            if (l1_phase == 0 && run_n >= l1_min) resolve_issue();
            continue;
        }
        const RowScan sa = scan_probe(beA, invA, beA1, beA2, iA1, iA2);
        const RowScan sb = scan_probe(beB, invB, beB1, beB2, iB1, iB2);
        const uint64_t ballA = __builtin_amdgcn_ballot_w64((sa.gw32 & sa.gm) == sa.gm && sa.vm != 0);
        const uint64_t ballB = __builtin_amdgcn_ballot_w64((sb.gw32 & sb.gm) == sb.gm && sb.vm != 0);
        const uint32_t nA = (uint32_t)__builtin_popcountll(ballA), nB = (uint32_t)__builtin_popcountll(ballB);
        if (VG_DBG(p.dbg) & 1u) continue;
        if (PT) {
            if (run_n + nA > RQ) make_room();
            enqueue(sa, ballA, run_head + run_n);
            run_n += nA;
            if (run_n + nB > RQ) make_room();
            enqueue(sb, ballB, run_head + run_n);
            run_n += nB;
            // one phase per iteration: every phase's loads have a whole iteration to arrive
            l1_advance();
            if (l1_phase == 0 && run_n >= l1_min) resolve_issue();
            continue;
        }
        enqueue(sa, ballA, run_head + run_n);
        run_n += nA;
        while (run_n + nB > VG_RUNQ) drain_step();   // make room for row B's runs (only rows dense in candidates need this)
        enqueue(sb, ballB, run_head + run_n);
        run_n += nB;
        if (run_n >= VG_RUN_BATCH_S || req_n >= 32u) drain_step();
        while (run_n >= 3 * VG_RUN_BATCH_S || req_n >= 48u) drain_step();   // dense stretches: keep the rings short
    }
    if (PT) {   // flush: until nothing is queued or in flight
        while (run_n != 0 || l1_phase != 0) {
            while (l1_phase != 0) l1_advance();
            if (run_n != 0) resolve_issue();
        }
    } else {
        do {   // flush: until the rings are empty and the last step issued nothing (finishing a batch can re-queue)
            drain_step();
        } while (run_n != 0 || req_n != 0 || __builtin_amdgcn_ballot_w64(b_active) != 0);
    }
    vm_wait_imm<0>();
}

// ------------------------------------------------------------------------------------------
// sequential kernel: literal restatement of the reference state machine, any k in 1..28.
// One lane per read (read r = bytes [off[r], off[r+1]-1), followed by its '\n'); used for even
// k, where palindromic k-mers and stale registers make emission history-dependent
// (src/kmer.cpp:134 `continue` before ++l; :145 only l is reset).
// ------------------------------------------------------------------------------------------
#define VG_SEQ_LDS 16384u   // bytes of read text a wavefront stages (64 reads of up to ~250 bases)
template <int MODE>
__global__ __launch_bounds__(256) void seq_kernel(RowParams p, const uint64_t* read_off, uint64_t n_reads)
{
    // The 64 reads of a wavefront are one contiguous stretch of the block: it is copied into LDS with coalesced 16-byte
    // loads and every lane walks ITS read there (a lane per read straight from global memory touches 64 different
    // lines per step and thrashes the 32 KiB L1).  Stretches longer than the staging area are read in place.
    __shared__ __attribute__((aligned(16))) uint8_t s_text[4][VG_SEQ_LDS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r_first = r - lane;
    bool staged = false;
    uint64_t stage_base = 0;
    if (r_first < n_reads) {
        const uint64_t r_last = r_first + 64 < n_reads ? r_first + 64 : n_reads;
        const uint64_t b = read_off[r_first] & ~15ULL, e = read_off[r_last];
        // (the exact tail behind a fast kernel: reads that end in front of emit_from have nothing to count)
        if (MODE == MODE_COUNT && e <= p.emit_from) return;
        if (e - b <= VG_SEQ_LDS && e <= ((p.n_bytes + 15) & ~15ULL)) {
            for (uint64_t o = b + lane * 16u; o < e; o += 1024)
                *reinterpret_cast<uint4*>(&s_text[wave][o - b]) = load_chunk(p.bases, p.n_bytes, o);
            staged = true;
            stage_base = b;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (r >= n_reads) return;
    const uint64_t s = read_off[r];
    const uint64_t e = read_off[r + 1] - (MODE == MODE_BLOOM ? 0 : 1);  // reads end with '\n'; a Bloom sequence does not
    if (e <= s) {
        atomicOr(p.status, 1u);
        return;
    }
    const uint32_t K = p.k;
    const uint64_t mask = (1ULL << (2 * K)) - 1, shift1 = 2 * (uint64_t)(K - 1);
    uint64_t fwd = 0, rc = 0;
    uint32_t l = 0;
    for (uint64_t i = s; i < e; ++i) {
        const uint32_t c = vg_nt4(staged ? s_text[wave][i - stage_base] : p.bases[i]);
        uint64_t out = ~0ULL;
        if (c < 4) {
            fwd = (fwd << 2 | c) & mask;
            rc = (rc >> 2) | (uint64_t)(3u ^ c) << shift1;
            if (fwd != rc) {
                ++l;
                if (l >= K) {
                    const uint64_t canon = fwd < rc ? fwd : rc;
                    if (MODE == MODE_COUNT) {
                        if (i >= p.emit_from && filter_test_global(p.table, canon)) table_count(p.table, canon);
                    } else if (MODE == MODE_BLOOM) {
                        bloom_add_key(p.bloom, vg_hash64(canon, mask) << 8 | K);
                    } else {
                        out = vg_hash64(canon, mask) << 8 | K;
                    }
                }
            }
        } else {
            l = 0;
        }
        if (MODE == MODE_KEYS) p.keys_out[i] = out;
    }
    if (MODE == MODE_KEYS) p.keys_out[e] = ~0ULL;  // the separator position
}

// ------------------------------------------------------------------------------------------
// even k = 20 .. 24 on a small graph, ahead of count27s_kernel<true, K> (round 5): what the reference's run counter suppresses.
// That kernel counts a window iff its k bases are bases -- the rule of odd k.  The reference (src/kmer.cpp:132-146) does not advance
// l on a window that is its own reverse complement -- of the REGISTERS' content: stale bases across a non-base (:145 resets l only)
// and the zeros in front of the read included -- so behind such a position, met while l < k, l lags behind the run of bases and the
// positions with  run >= k > l  are not emitted.
//
// Only a non-base can start such a lag.  From a read's start the registers are zero, and with j + 1 < k bases shifted in the two
// registers cannot be equal: field 0 of rc is still zero, which asks for base j = A, and field k - 1 of fwd is still zero, which asks
// for base j = T.  So l counts every base up to the first whole window, and from there on a window that is not counted is a palindrome,
// which the fast kernel does not count either (no path-table bit).  Behind a non-base the registers keep the bases in front of it, l
// starts again at zero, and a palindrome of old and new bases makes the lag.  Hence this pass (second form; the first walked every
// read's first k-odd bases through the state machine, a lane per read: 4.6 of an even k's 6.0 ms per 1.5e7 reads):
//   * the text is SCANNED for bytes that are neither ACGT nor a newline, 16 bytes per lane, coalesced, four pieces in flight: per
//     word  sel = (w >> 1) & 07070707  is distinct for 'A' 'C' 'T' 'G' '\n' (0 1 2 3 5), one v_perm_b32 turns it back into the byte it
//     stands for, and any difference from w is a byte to look at.  That is the whole cost for reads of bases: the text at HBM speed.
//   * a lane that holds such a byte -- a real non-base, not a lower-case base (vg_nt4 knows those), with a base behind it -- finds its
//     read (binary search in the offsets), rebuilds the registers from the k bases in front (fewer at a read's start: zeros, as the
//     reference has them; older non-bases skipped, as the reference skips them) and walks the state machine until l = k, the next
//     non-base (which has its own lane) or the read's end.
// The k-mer of a suppressed position loses one count BEFORE the fast kernel adds it -- the counters are 32-bit and wrap; every debit
// is followed by its increment in the same stream; a saturation flag is set by the increment that takes the running value, which never
// exceeds the true one, from 254 to 255, and the read-out of even k honours the flag (cov_kernel).  Positions at or behind emit_from
// belong to the exact tail launch (seq_kernel<MODE_COUNT>).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t debit_odd_bytes(uint32_t w)      // non-zero iff a byte of w is none of 'A' 'C' 'G' 'T' '\n'
{
    // v_perm_b32: selector 0..3 = byte of the second operand, 4..7 = byte of the first; slots 4, 6, 7 hold 0, which no byte with such
    // a selector is (0 has selector 0)
    return __builtin_amdgcn_perm(0x00000A00u, 0x47544341u, (w >> 1) & 0x07070707u) ^ w;
}

// What happens behind the non-base at byte i (rare in a wavefront's 4 KiB, not in a workgroup's 16 KiB: reads carry an N per ten
// thousand bases, one read in seventy has one).  Nothing here may wait for one load after another -- found by a binary search over the
// offsets and walked byte by byte, a non-base cost ~80 dependent round trips and the pass 6 ms per 2e7 reads, as much as the state
// machine over every read's head had: the 32 bytes on either side come as four 16-byte loads in flight together (byte-aligned
// addresses: the memory path takes them), are turned into 2-bit codes and flag words, and are walked in registers.  A read's '\n' on
// either side ends the walk, so the read's offsets are not needed; only what lies further than 32 bytes away (more than eight older
// non-bases among the k bases in front, more than eight palindromes behind) or next to the text's ends is read byte by byte.
// (Inlined, once: as a call it takes RowParams by address, and the kernel then keeps its 464 bytes of arguments in scratch, written by
// every lane of the scan.)
typedef uint32_t vg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void debit_codes16(const vg_u32x4 v, uint32_t t0, bool backward, uint64_t& code, uint32_t& bad, uint32_t& nl)
{
#pragma unroll 1
    for (uint32_t q = 0; q < 4; ++q) {      // (rolled, the word picked by selects: this runs once per non-base, the registers it would take unrolled
                                            // are the occupancy of the scan)
        uint32_t w = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
#pragma unroll 1
        for (uint32_t j = 4 * q; j < 4 * q + 4; ++j, w >>= 8) {
            const uint32_t b = w & 0xFFu, c = vg_nt4(b), t = t0 + (backward ? 15 - j : j);
            code |= (uint64_t)(c & 3u) << (2 * t);
            bad |= (c >> 2) << t;
            nl |= (uint32_t)(b == '\n') << t;
        }
    }
}

__device__ __forceinline__ void debit_behind_non_base(const RowParams& p, uint64_t i, uint64_t lo, uint64_t hi)
{
    const uint8_t* const bases = p.bases;
    const uint32_t K = p.k;
    const uint64_t mask = (1ULL << (2 * K)) - 1, shift1 = 2 * (uint64_t)(K - 1);
    uint64_t fwd = 0, rc = 0;
    uint32_t l = 0, run = 0, idx = 0;
    auto older = [&](uint32_t c) -> bool {      // a base in front of i, nearest first; false: the registers are full
        fwd |= (uint64_t)c << (2 * idx);
        rc |= (uint64_t)(3u ^ c) << (2 * (K - 1 - idx));
        return ++idx < K;
    };
    auto step = [&](uint32_t c) -> bool {       // a base behind i; false: l has reached k
        fwd = (fwd << 2 | c) & mask;
        rc = (rc >> 2) | (uint64_t)(3u ^ c) << shift1;
        if (run < K) ++run;
        if (fwd == rc) return true;
        if (++l >= K) return false;
        if (run >= K) {      // the fast kernel counts this window, the reference does not
            const uint64_t canon = fwd < rc ? fwd : rc;
            if (filter_test_global(p.table, canon)) table_debit(p.table, canon);
        }
        return true;
    };
    uint64_t back_from = i, on_from = i + 1;      // where the byte-by-byte walks take over
    bool more = true;
    if (i >= 32 && i + 33 <= p.n_bytes) {
        vg_u32x4 f0, f1, b0, b1;
        __builtin_memcpy(&f0, bases + i + 1, 16);
        __builtin_memcpy(&f1, bases + i + 17, 16);
        __builtin_memcpy(&b0, bases + i - 16, 16);
        __builtin_memcpy(&b1, bases + i - 32, 16);
        uint64_t fcode = 0, bcode = 0;
        uint32_t fbad = 0, bbad = 0, fnl = 0, bnl = 0;      // bit t: the byte t + 1 behind / in front of i is no base; ... is a '\n'
        debit_codes16(f0, 0, false, fcode, fbad, fnl);
        debit_codes16(f1, 16, false, fcode, fbad, fnl);
        debit_codes16(b0, 0, true, bcode, bbad, bnl);
        debit_codes16(b1, 16, true, bcode, bbad, bnl);
        // bytes outside the text of the reads end the walks like a read's end does
        const uint64_t nf = hi - i - 1, nb = i - lo;
        if (nf < 32) fbad |= ~0u << nf;
        if (nb < 32) bnl |= ~0u << nb;
        if (fbad & 1u) return;      // a non-base or the read's end next: nothing behind it that could be counted
        // the registers as the reference has them here: the last k bases of the read in front of i (fewer: zeros in front of them)
        for (uint32_t t = 0; t < 32 && more; ++t) {
            if ((bnl >> t) & 1u) more = false;
            else if (!((bbad >> t) & 1u)) more = older((uint32_t)(bcode >> (2 * t)) & 3u);
        }
        back_from = i - 32;
        if (more) {
            for (uint64_t q = back_from; q > lo;) {
                --q;
                const uint32_t b = bases[q], c = vg_nt4(b);
                if (b == '\n') break;
                if (c < 4 && !older(c)) break;
            }
        }
        for (uint32_t t = 0; t < 32; ++t) {
            if ((fbad >> t) & 1u) return;
            if (!step((uint32_t)(fcode >> (2 * t)) & 3u)) return;
        }
        on_from = i + 33;
    } else {
        if (i + 1 >= hi || vg_nt4(bases[i + 1]) >= 4) return;
        for (uint64_t q = back_from; q > lo;) {
            --q;
            const uint32_t b = bases[q], c = vg_nt4(b);
            if (b == '\n') break;
            if (c < 4 && !older(c)) break;
        }
    }
    for (uint64_t x = on_from; x < hi; ++x) {
        const uint32_t c = vg_nt4(bases[x]);
        if (c >= 4 || !step(c)) return;
    }
}

constexpr uint32_t VG_DEBIT_PIECES = 4;      // 16-byte pieces per lane: a workgroup scans 16 KiB

// (the list: a non-base with a base behind it is work for ONE lane -- 10-20 us of it, four loads and a hundred steps -- and a workgroup of
// the scan meets 1.6 of them in its 16 KiB: walked where they are found, the scan ran at 0.6 TB/s, every workgroup waiting for a lane.
// The scan only writes their positions down, a second launch walks them a lane each; what does not fit the list is walked on the spot.)
__global__ __launch_bounds__(256) void even_debit_kernel(RowParams p, const uint64_t* __restrict__ read_off, uint64_t n_reads,
                                                         unsigned long long* __restrict__ list, uint32_t list_cap, unsigned int* list_n)
{
    const uint64_t limit = p.emit_from < p.n_bytes ? p.emit_from : p.n_bytes;
    const uint64_t base = (uint64_t)blockIdx.x * (256u * 16u * VG_DEBIT_PIECES) + threadIdx.x * 16u;
    vg_u32x4 v[VG_DEBIT_PIECES];
#pragma unroll
    for (uint32_t t = 0; t < VG_DEBIT_PIECES; ++t) {
        const uint64_t a = base + (uint64_t)t * 4096u;
        if (a + 16 <= p.n_bytes) v[t] = __builtin_nontemporal_load(reinterpret_cast<const vg_u32x4*>(p.bases + a));
        else {
            const uint4 c = a < limit ? load_chunk(p.bases, p.n_bytes, a) : make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
            v[t] = vg_u32x4{c.x, c.y, c.z, c.w};
        }
    }
    uint32_t any = 0;
#pragma unroll
    for (uint32_t t = 0; t < VG_DEBIT_PIECES; ++t)
        any |= debit_odd_bytes(v[t].x) | debit_odd_bytes(v[t].y) | debit_odd_bytes(v[t].z) | debit_odd_bytes(v[t].w);
    if (any == 0 || n_reads == 0) return;
    // the text of the reads: [lo, hi); positions at or behind emit_from are the tail launch's
    const uint64_t lo = read_off[0], hi = read_off[n_reads] < limit ? read_off[n_reads] : limit;
    static_assert(VG_DEBIT_PIECES == 4, "the selects below");
#pragma unroll 1
    for (uint32_t tq = 0; tq < 4 * VG_DEBIT_PIECES; ++tq) {      // (rolled: ONE copy of the walk in the kernel; the words picked by selects, not by address)
        const uint32_t t = tq >> 2, q = tq & 3u;
        const vg_u32x4 vt = t == 0 ? v[0] : t == 1 ? v[1] : t == 2 ? v[2] : v[3];
        const uint32_t w = q == 0 ? vt.x : q == 1 ? vt.y : q == 2 ? vt.z : vt.w;
        uint32_t m = debit_odd_bytes(w);
        while (m) {
            const uint32_t j = (uint32_t)__builtin_ctz(m) >> 3;
            m &= ~(0xFFu << (8 * j));
            const uint64_t i = base + (uint64_t)t * 4096u + 4 * q + j;
            const uint32_t b = (w >> (8 * j)) & 0xFFu;
            if (i < lo || i + 1 >= hi || vg_nt4(b) < 4) continue;      // (lower case and 'U' are bases)
            // (256 lists by workgroup number, their counters a line apart: three hundred thousand returning atomics on ONE word were 2 of the scan's 3.3 ms)
            const uint32_t sub = blockIdx.x & (VG_DEBIT_SUBLISTS - 1u), sub_cap = list_cap / VG_DEBIT_SUBLISTS;
            const uint32_t at = atomicAdd(list_n + 16u * sub, 1u);
            if (at < sub_cap) list[(size_t)sub * sub_cap + at] = i;
            else debit_behind_non_base(p, i, lo, hi);
        }
    }
}

__global__ __launch_bounds__(256) void even_debit_walk_kernel(RowParams p, const uint64_t* __restrict__ read_off, uint64_t n_reads,
                                                              const unsigned long long* __restrict__ list, uint32_t list_cap,
                                                              const unsigned int* __restrict__ list_n)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, sub_cap = list_cap / VG_DEBIT_SUBLISTS, sub = g / sub_cap;
    if (g >= list_cap || g - sub * sub_cap >= list_n[16u * sub] || n_reads == 0) return;
    const uint64_t limit = p.emit_from < p.n_bytes ? p.emit_from : p.n_bytes;
    const uint64_t lo = read_off[0], hi = read_off[n_reads] < limit ? read_off[n_reads] : limit;
    debit_behind_non_base(p, list[g], lo, hi);
}

// ------------------------------------------------------------------------------------------
// even k over ONE long sequence (K3 on a chromosome): the state machine above, one lane per segment.
// The emitter's state at a position is (fwd, rc, l): the registers hold the last k VALID bases (a non-base only
// resets l, src/kmer.cpp:145; zeros before the sequence's first bases), and only min(l, k) matters.  A lane rebuilds
// it exactly from a look-back window in front of its segment: it steps back over `want` valid bases (or to the start
// of the sequence) and replays from there with zero registers.  After k valid bases the registers are exact; l is
// exact from the first non-base after that point on, and otherwise at least the count of non-palindromic positions
// replayed with exact registers -- which settles min(l, k) once that count reaches k.  If neither happens (more than
// k palindromic positions in a row: not a real sequence) the window is doubled.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bloom_even_kernel(RowParams p, uint64_t seg_len)
{
    __shared__ unsigned long long s_bkey[4 * VG_BPRIV_CELLS];
    __shared__ uint32_t s_bcnt[4 * VG_BPRIV_CELLS];
    unsigned long long* const cell_key = s_bkey + (threadIdx.x >> 6) * VG_BPRIV_CELLS;
    uint32_t* const cell_cnt = s_bcnt + (threadIdx.x >> 6) * VG_BPRIV_CELLS;
    for (uint32_t i = threadIdx.x & 63u; i < VG_BPRIV_CELLS; i += 64) {
        cell_key[i] = ~0ULL;
        cell_cnt[i] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint64_t seg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t s = seg * seg_len;
    if (s >= p.n_bytes) return;
    const uint64_t e = s + seg_len < p.n_bytes ? s + seg_len : p.n_bytes;
    const uint32_t K = p.k;
    const uint64_t mask = (1ULL << (2 * K)) - 1, shift1 = 2 * (uint64_t)(K - 1);
    // nothing to emit in a segment without bases (long runs of N): no look-back either
    {
        bool any = false;
        for (uint64_t i = s; i < e && !any; ++i) any = vg_nt4(p.bases[i]) < 4;
        if (!any) return;
    }
    uint64_t fwd = 0, rc = 0;
    uint32_t l = 0;
    for (uint64_t want = 3ull * K;; want *= 2) {
        // window start: `want` valid bases back, or the start of the sequence
        uint64_t b = s;
        uint64_t got = 0;
        while (b > 0 && got < want) {
            --b;
            if (vg_nt4(p.bases[b]) < 4) ++got;
        }
        fwd = 0;
        rc = 0;
        l = 0;
        const bool from_start = b == 0;
        uint64_t valid = 0, exact_nonpal = 0;
        bool l_exact = from_start;   // replay from the sequence start is the reference's own run
        for (uint64_t i = b; i < s; ++i) {
            const uint32_t c = vg_nt4(p.bases[i]);
            if (c < 4) {
                fwd = (fwd << 2 | c) & mask;
                rc = (rc >> 2) | (uint64_t)(3u ^ c) << shift1;
                ++valid;
                if (fwd != rc) {
                    if (l < K) ++l;
                    if (from_start || valid > K) ++exact_nonpal;
                }
            } else {
                l = 0;
                if (from_start || valid >= K) l_exact = true;   // registers exact here: l is exact from now on
                exact_nonpal = 0;
            }
        }
        if (l_exact) break;
        if (exact_nonpal >= K) {   // the true l is at least this: saturated
            l = K;
            break;
        }
    }
    // the lanes of a wave walk their segments in step, so the keys of one step go through the per-wave table like in
    // the odd-k kernel (neighbouring segments of a low-complexity region emit the same k-mer)
    for (uint64_t t = 0; t < seg_len; ++t) {
        const uint64_t i = s + t;
        bool emit = false;
        uint64_t key = 0;
        if (i < e) {
            const uint32_t c = vg_nt4(p.bases[i]);
            if (c < 4) {
                fwd = (fwd << 2 | c) & mask;
                rc = (rc >> 2) | (uint64_t)(3u ^ c) << shift1;
                if (fwd != rc) {
                    if (l < K) ++l;
                    if (l >= K) {
                        emit = true;
                        key = vg_hash64(fwd < rc ? fwd : rc, mask) << 8 | K;
                    }
                }
            } else {
                l = 0;
            }
        }
        bloom_add_wave(p.bloom, cell_key, cell_cnt, emit, key);
    }
}

// ------------------------------------------------------------------------------------------
// table build (once per graph) and per-sample reset / read-out
// ------------------------------------------------------------------------------------------
__global__ void table_clear_kernel(TableView t)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > t.cap_mask) return;
    if (t.slots8) t.slots8[i] = VG_EMPTY;
    else *reinterpret_cast<uint4*>(&t.slots[i]) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu);
}

__global__ void table_insert_kernel(TableView t, const uint64_t* keys, uint64_t n, uint32_t k,
                                    uint32_t* key_slot, uint32_t* filter_rw, uint32_t* grid_rw, bool grid12, uint32_t* status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    if ((key & 0xFFu) != k || (key >> 8) > mask) {
        atomicOr(status, 2u);
        return;
    }
    const uint64_t canon = vg_hash64_inv(key >> 8, mask);
    uint64_t s = table_home(t, canon);
    for (;;) {
        unsigned long long* cell = t.slots8 ? &t.slots8[s] : &t.slots[s].canon;
        const unsigned long long prev = atomicCAS(cell, (unsigned long long)VG_EMPTY, (unsigned long long)canon);
        if (prev == VG_EMPTY) break;
        if ((prev & VG_SLOT_KMER_MASK) == canon) {
            atomicOr(status, 4u);  // duplicate key
            return;
        }
        atomicOr(cell, (unsigned long long)VG_SLOT_CHAIN);   // this key goes past the slot: lookups must follow
        s = (s + 1) & t.cap_mask;
    }
    if (!t.slots8) t.slots[s].key_index = (uint32_t)i;
    key_slot[i] = (uint32_t)s;
    atomicOr(&filter_rw[vg_fhash_word(canon) >> t.filter_shift], vg_fhash_bits(canon, t.filter_words_log2));
    if (grid_rw && grid12) {   // small graphs: the k - 11 twelve-mers of the k-mer (16 at k = 27; vgmi_device.h, VG_GRID12_*)
        for (uint32_t off = 0; off + 11u < k; ++off) {
            uint32_t w, m;
            vg_grid12_probe((uint32_t)(canon >> (2 * off)) & 0xFFFFFFu, w, m);
            atomicOr(&grid_rw[w], m);
        }
    } else if (grid_rw) {   // k = 27 only: the 12 sixteen-mers of the k-mer (canonicalised inside vg_grid_probe, which
                     // makes the reverse complement's twelve the same entries)
        const bool wide = t.grid_words_log2 != VG_GRID_LDS_WORDS_LOG2;   // global variant: 64-bit entries + offset bits
        for (uint32_t off = 0; off < VG_GRID_STEP; ++off) {
            uint64_t w;
            uint32_t m, rot;
            bool as_is;
            const uint32_t mer = (uint32_t)(canon >> (2 * off));
            vg_grid_probe(mer, t.grid_words_log2, w, m, rot, as_is);
            if (!wide) {
                atomicOr(&grid_rw[w], m);
            } else {
                const uint32_t b = as_is ? off : 11u - off;
                uint32_t hi = 1u << ((rot + b) & 31u);
                if (mer == vg_revcomp16(mer)) hi |= 1u << ((rot + 11u - b) & 31u);   // palindromic 16-mer: both readings
                atomicOr(&grid_rw[2 * w], m);
                atomicOr(&grid_rw[2 * w + 1], hi);
            }
        }
    }
}

// key -> its index in the uploaded key array (graph2node's mGraphKmerHashHapStrMap.find, src/construct_index.cpp:710-751), batched:
// the compact format keeps key_slot (index -> slot) only, so the caller hands in the inverse (slot -> index) for the call
__global__ void table_key_of_slot_kernel(const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key_of_slot[key_slot[i]] = (uint32_t)i;
}

__global__ void table_lookup_kernel(TableView t, const uint64_t* keys, uint64_t n, uint32_t k, const uint32_t* key_of_slot, uint32_t* out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    uint32_t found = 0xFFFFFFFFu;
    if ((key & 0xFFu) == k && (key >> 8) <= mask) {
        const uint64_t canon = vg_hash64_inv(key >> 8, mask);
        uint64_t s = table_home(t, canon);
        for (;;) {
            const uint64_t c = t.slots8 ? t.slots8[s] : t.slots[s].canon;
            if (c == VG_EMPTY) break;
            if ((c & VG_SLOT_KMER_MASK) == canon) {
                found = t.slots8 ? key_of_slot[s] : t.slots[s].key_index;
                break;
            }
            if (!(c & VG_SLOT_CHAIN)) break;
            s = (s + 1) & t.cap_mask;
        }
    }
    out[i] = found;
}

// per-sample reset of the table side (the counter arrays are cleared by memset): in-slot counters of the 16-byte
// format, saturation flags of the compact one
__global__ __launch_bounds__(256) void counts_reset_kernel(TableView t)
{
    if (t.slots8) {
        // saturation flags: only the 2048-slot regions a flag was set in this sample (all of them for an image of unknown
        // history: the dirty bytes are then preset)
        const uint64_t n_regions = (t.cap_mask >> VG_SAT_REGION_LOG2) + 1;
        for (uint64_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
            if (!t.sat_dirty[r]) continue;
            const uint64_t b = r << VG_SAT_REGION_LOG2;
            const uint64_t e = b + (1ULL << VG_SAT_REGION_LOG2) <= t.cap_mask + 1 ? b + (1ULL << VG_SAT_REGION_LOG2) : t.cap_mask + 1;
            for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
                const unsigned long long c = t.slots8[i];
                if (c != VG_EMPTY && (c & VG_SLOT_SAT)) t.slots8[i] = c & ~VG_SLOT_SAT;
            }
            __syncthreads();
            if (threadIdx.x == 0) t.sat_dirty[r] = 0;
        }
    } else {
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= t.cap_mask; i += stride) t.slots[i].count = 0;
    }
}

// K5 part 1 + K6: cov[i] = min(255, count(key i)); hist[c] += 1 for flagged keys with c != 0
__global__ void cov_kernel(TableView t, const uint32_t* key_slot, uint64_t n, const uint8_t* flag, uint8_t* cov,
                           unsigned long long* hist)
{
    __shared__ unsigned int s_hist[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t c32 = *count_cell(t, key_slot[i], (uint32_t)i);
        uint32_t c = c32 < 255u ? c32 : 255u;
        // (even k on the fast path: the counter of a saturated k-mer may stand below the clamp -- a debit of seq_kernel<MODE_DEBIT> whose
        // increment the saturation flag then skipped; the flag says the sum reached it)
        if (t.slots8 && !(t.k & 1u) && (t.slots8[key_slot[i]] & VG_SLOT_SAT)) c = 255u;
        cov[i] = (uint8_t)c;
        if (hist && c != 0 && flag && flag[i]) atomicAdd(&s_hist[c], 1u);
    }
    __syncthreads();
    if (hist)
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}

// raw 32-bit counters <-> dense key-ordered array (all-reduce of a read-sharded sample)
__global__ void counts_xfer_kernel(TableView t, const uint32_t* key_slot, uint32_t* ext, uint64_t n, bool import)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t* cell = count_cell(t, key_slot[i], (uint32_t)i);
        if (import) *cell = ext[i];
        else {
            uint32_t v = *cell;
            if (t.slots8 && !(t.k & 1u) && (t.slots8[key_slot[i]] & VG_SLOT_SAT) && v < 255u) v = 255u;      // (even k, fast path: see cov_kernel)
            ext[i] = v;
        }
    }
}

// K5 part 2: per-node depth gather in CSR order
__global__ void node_gather_kernel(const uint8_t* cov, const uint32_t* key_index, uint64_t n, uint8_t* cov_node)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        cov_node[i] = cov[key_index[i]];
}

// K4: BloomFilter::count (min) and ::find (all non-zero) for a batch of keys
__global__ void bloom_query_kernel(BloomView b, const uint64_t* keys, uint64_t n, uint8_t* min_out, uint8_t* nz_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t mn = 255, nz = 1;
    for (uint32_t h = 0; h < b.n_hash; ++h) {
        const uint32_t v = b.filter[mod_u64(vg_murmur_sum(keys[i], b.seeds[h]), b.m, b.magic)];
        mn = v < mn ? v : mn;
        nz &= v != 0;
    }
    if (min_out) min_out[i] = (uint8_t)mn;
    if (nz_out) nz_out[i] = (uint8_t)nz;
}

// ------------------------------------------------------------------------------------------
// launch wrappers (called from vgmi_api*.cpp through vgmi_kernels.h)
// ------------------------------------------------------------------------------------------
static size_t rows_lds_bytes(int mode, bool flds, uint32_t filter_words_log2, uint32_t block)
{
    size_t b = 512;
    if (flds) b += (size_t)4 << filter_words_log2;
    if (mode == MODE_COUNT) b += (size_t)(block / 64) * VG_QCAP * 8;
    if (mode == MODE_BLOOM) b += (size_t)(block / 64) * VG_BPRIV_CELLS * 12;
    return b;
}

template <int MODE, bool FLDS>
static hipError_t launch_rows_t(const RowParams& p, uint32_t grid, uint32_t block, hipStream_t st)
{
    const size_t lds = rows_lds_bytes(MODE, FLDS, p.table.filter_words_log2, block);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_kernel<MODE, FLDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rows_kernel<MODE, FLDS>), dim3(grid), dim3(block), lds, st, p);
    return hipGetLastError();
}

template <bool LDS_BM, bool COMPACT>
static hipError_t launch_count27_t(const RowParams& p, uint32_t grid, uint32_t block, hipStream_t st)
{
    const size_t lds = (LDS_BM ? (size_t)VG_GRID_LDS_WORDS * 4 : 0) + (size_t)(block / 64) * (VG_RUNQ * 16 + VG_REQ * 8) + VG_LUT27_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&count27_kernel<LDS_BM, COMPACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((count27_kernel<LDS_BM, COMPACT>), dim3(grid), dim3(block), lds, st, p);
    return hipGetLastError();
}

template <uint32_t K>
static hipError_t launch_countks_t(const RowParams& p, uint32_t grid, size_t lds, hipStream_t st)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&count27s_kernel<true, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((count27s_kernel<true, K>), dim3(grid), dim3(1024), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_count27s(const RowParams& p, uint32_t grid, hipStream_t st)
{
    const size_t lds = (size_t)VG_GRID_LDS_WORDS * 4 + (size_t)16 * (VG_RUNQ * 16 + VG_REQ * 8) + VG_LUT27_BYTES;
    if (p.k != 27) {      // small graphs of odd k = 19 .. 25: the path-table form on the grid of 8
        if (!p.table.pt.index) return hipErrorInvalidValue;
        switch (p.k) {
        case 19: return launch_countks_t<19>(p, grid, lds, st);
        case 21: return launch_countks_t<21>(p, grid, lds, st);
        case 23: return launch_countks_t<23>(p, grid, lds, st);
        case 25: return launch_countks_t<25>(p, grid, lds, st);
        case 20: return launch_countks_t<20>(p, grid, lds, st);
        case 22: return launch_countks_t<22>(p, grid, lds, st);
        case 24: return launch_countks_t<24>(p, grid, lds, st);
        default: return hipErrorInvalidValue;
        }
    }
    const void* fn = p.table.pt.index ? reinterpret_cast<const void*>(&count27s_kernel<true>) : reinterpret_cast<const void*>(&count27s_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (p.table.pt.index) hipLaunchKernelGGL(count27s_kernel<true>, dim3(grid), dim3(1024), lds, st, p);
    else hipLaunchKernelGGL(count27s_kernel<false>, dim3(grid), dim3(1024), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_count27(bool lds_bitmap, const RowParams& p, uint32_t grid, uint32_t block, hipStream_t st)
{
    if (lds_bitmap) return launch_count27_t<true, true>(p, grid, block, st);
    return p.table.slots8 ? launch_count27_t<false, true>(p, grid, block, st) : launch_count27_t<false, false>(p, grid, block, st);
}

hipError_t launch_rows(int mode, bool flds, const RowParams& p, uint32_t grid, uint32_t block, hipStream_t st)
{
    if (mode == MODE_COUNT) return flds ? launch_rows_t<MODE_COUNT, true>(p, grid, block, st)
                                        : launch_rows_t<MODE_COUNT, false>(p, grid, block, st);
    if (mode == MODE_KEYS) return launch_rows_t<MODE_KEYS, false>(p, grid, block, st);
    return launch_rows_t<MODE_BLOOM, false>(p, grid, block, st);
}

hipError_t launch_seq(int mode, const RowParams& p, const uint64_t* read_off, uint64_t n_reads, hipStream_t st)
{
    const uint32_t block = 256;
    const uint32_t grid = (uint32_t)((n_reads + block - 1) / block);
    if (grid == 0) return hipSuccess;
    if (mode == MODE_COUNT) hipLaunchKernelGGL((seq_kernel<MODE_COUNT>), dim3(grid), dim3(block), 0, st, p, read_off, n_reads);
    else if (mode == MODE_KEYS) hipLaunchKernelGGL((seq_kernel<MODE_KEYS>), dim3(grid), dim3(block), 0, st, p, read_off, n_reads);
    else if (mode == MODE_DEBIT) return hipErrorInvalidValue;      // launch_even_debit
    else hipLaunchKernelGGL((seq_kernel<MODE_BLOOM>), dim3(grid), dim3(block), 0, st, p, read_off, n_reads);
    return hipGetLastError();
}

// list: list_cap positions and, behind them, the counters (one allocation per stream that counts even k: vgmi_api.cpp)
hipError_t launch_even_debit(const RowParams& p, const uint64_t* read_off, uint64_t n_reads, unsigned long long* list, uint32_t list_cap, hipStream_t st)
{
    const uint64_t limit = p.emit_from < p.n_bytes ? p.emit_from : p.n_bytes, per_wg = 256u * 16u * VG_DEBIT_PIECES;
    if (limit == 0 || n_reads == 0) return hipSuccess;
    unsigned int* const list_n = reinterpret_cast<unsigned int*>(list + list_cap);
    hipError_t e = hipMemsetAsync(list_n, 0, VG_DEBIT_SUBLISTS * 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(even_debit_kernel, dim3((uint32_t)((limit + per_wg - 1) / per_wg)), dim3(256), 0, st, p, read_off, n_reads, list, list_cap, list_n);
    hipLaunchKernelGGL(even_debit_walk_kernel, dim3((list_cap + 255) / 256), dim3(256), 0, st, p, read_off, n_reads, list, list_cap, list_n);
    return hipGetLastError();
}

hipError_t launch_bloom_even(const RowParams& p, hipStream_t st)
{
    const uint64_t seg_len = 1024;
    const uint64_t n_seg = (p.n_bytes + seg_len - 1) / seg_len;
    if (n_seg == 0) return hipSuccess;
    hipLaunchKernelGGL(bloom_even_kernel, dim3((uint32_t)((n_seg + 255) / 256)), dim3(256), 0, st, p, seg_len);
    return hipGetLastError();
}

static uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap)
{
    uint64_t g = (n + block - 1) / block;
    if (g == 0) g = 1;
    return (uint32_t)(g < cap ? g : cap);
}

hipError_t launch_table_clear(const TableView& t, hipStream_t st)
{
    hipLaunchKernelGGL(table_clear_kernel, dim3((uint32_t)((t.cap_mask + 256) / 256)), dim3(256), 0, st, t);
    return hipGetLastError();
}

hipError_t launch_table_insert(const TableView& t, const uint64_t* keys, uint64_t n, uint32_t k, uint32_t* key_slot,
                               uint32_t* filter_rw, uint32_t* grid_rw, bool grid12, uint32_t* status, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(table_insert_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, t, keys, n, k, key_slot,
                       filter_rw, grid_rw, grid12, status);
    return hipGetLastError();
}

hipError_t launch_table_key_of_slot(const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(table_key_of_slot_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, key_slot, n, key_of_slot);
    return hipGetLastError();
}

hipError_t launch_table_lookup(const TableView& t, const uint64_t* keys, uint64_t n, uint32_t k, const uint32_t* key_of_slot, uint32_t* out,
                               hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(table_lookup_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, t, keys, n, k, key_of_slot, out);
    return hipGetLastError();
}

hipError_t launch_counts_reset(const TableView& t, hipStream_t st)
{
    hipLaunchKernelGGL(counts_reset_kernel, dim3(grid_for(t.cap_mask + 1, 256, 4096)), dim3(256), 0, st, t);
    return hipGetLastError();
}

hipError_t launch_cov(const TableView& t, const uint32_t* key_slot, uint64_t n, const uint8_t* flag, uint8_t* cov,
                      unsigned long long* hist, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cov_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, t, key_slot, n, flag, cov, hist);
    return hipGetLastError();
}

hipError_t launch_counts_xfer(const TableView& t, const uint32_t* key_slot, uint32_t* ext, uint64_t n, bool import,
                              hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(counts_xfer_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, t, key_slot, ext, n, import);
    return hipGetLastError();
}

hipError_t launch_node_gather(const uint8_t* cov, const uint32_t* key_index, uint64_t n, uint8_t* cov_node, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(node_gather_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, cov, key_index, n, cov_node);
    return hipGetLastError();
}

hipError_t launch_bloom_query(const BloomView& b, const uint64_t* keys, uint64_t n, uint8_t* min_out, uint8_t* nz_out,
                              hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bloom_query_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, b, keys, n, min_out, nz_out);
    return hipGetLastError();
}

}  // namespace vgk
