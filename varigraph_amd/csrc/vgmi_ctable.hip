// vgmi_ctable.hip -- the large-graph read-counting kernel over the CONTEXT TABLE (k = 27, graphs of > 65 536 k-mers:
// BASELINE configs 3-5).  Entry algebra and the reasons: vgmi_ctable.h.
//
// Same reference behaviour as every count kernel here: src/kmer.cpp:110-149 (emitter, odd k: a window counts iff its 27
// bases are bases), :140-142 (membership), src/fastq_kmer.cpp:128-139 (saturating count).
//
// Memory-side requests per 150-base read (the quantity this regime is bound by): 1.2 of the row stream + 12.6
// buckets (one 64-byte request per grid position: no filter in front, no table line behind) + the counter atomics of the hits,
// against 1.2 + 12.6 filter words + 5.2 table lines + 2.8 atomic requests of count27x_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vgmi_ctable.h"
#include "vgmi_xtable.h"

namespace vgk {

#define CT_RUNQ 256u            // run ring per wavefront (records of 8 bytes): a row adds <= 256, a drain step takes 5
#define CT_PENDQ 128u           // contexts that go on to a next bucket (16 bytes): a row or a batch adds <= 64, a batch takes 64 once 64 wait
#define CT_NONE 0xFFFFFFFFu
#define CT_MASK54 VG_SLOT_KMER_MASK      // the k-mer bits of a compact slot and of an okmer word: 56 (k <= 28)
#define CT_OK_FIRST 56u               // okmer word: k-mer | first of its unitig << 56 | k-mers behind it there (capped at 15) << 57; bit 63: no k-mer at this place

// ---- build ----------------------------------------------------------------------------------------------------------
__global__ void ct_clear_kernel(uint4* cb, uint64_t n_buckets)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 4 * n_buckets; i += stride)
        cb[i] = (i & 3u) ? make_uint4(0, 0, 0, 0) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);      // CtBucket: four empty X, then the triples
}

// Per key, from the numbering of vgmi_ptable.hip (pos_of_key = place | walked-as-canonical << 31; link2 = the mutual unique
// links): the k-mer as the walk reads it, whether it is the first of its unitig, and how many k-mers follow it there (capped
// at 15) -- "unitig" meaning what the entries need: neighbours in the key set WITH consecutive places, whatever numbering was
// used (keys on cycles, the identity fallback: chains of one).
__global__ void ct_okmer_kernel(TableView t, const uint32_t* key_slot, const uint32_t* pos_of_key, const uint32_t* link2, uint64_t n,
                                unsigned long long* okmer, uint32_t* id_of_key, unsigned long long* n_unitigs)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pk = pos_of_key[i], p = pk & 0x7FFFFFFFu, out = pk >> 31;
    const uint64_t K = t.slots8[key_slot[i]] & CT_MASK54;
    const uint64_t Kw = out ? K : vg_revcomp(K, t.k);
    bool first = true;
    const uint32_t lb = link2[2 * i + (out ^ 1u)];
    if (lb != CT_NONE) {
        const uint32_t pkn = pos_of_key[lb & 0x7FFFFFFFu];
        if ((pkn >> 31) == (lb >> 31) && (pkn & 0x7FFFFFFFu) + 1u == p) first = false;
    }
    uint32_t cur = (uint32_t)i, o = out, q = p, cnt = 0;
    while (cnt < 15u) {
        const uint32_t l = link2[2ull * cur + o];
        if (l == CT_NONE) break;
        const uint32_t nx = l & 0x7FFFFFFFu, nxo = (l >> 31) ^ 1u, pkx = pos_of_key[nx];
        if ((pkx >> 31) != nxo || (pkx & 0x7FFFFFFFu) != q + 1u) break;
        cur = nx;
        o = nxo;
        ++q;
        ++cnt;
    }
    okmer[p] = Kw | (unsigned long long)first << CT_OK_FIRST | (unsigned long long)cnt << (CT_OK_FIRST + 1u);
    id_of_key[i] = p;
    if (first) atomicAdd(n_unitigs, 1ULL);
}

__global__ void ct_identity_kernel(uint32_t* pos_of_key, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos_of_key[i] = (uint32_t)i | 1u << 31;
}

// one thread per (place p of the numbering, offset o of X in the k-mer: 0 .. k - 16): the pair leads an occurrence iff o == k - 16 or the k-mer is the first of its
// unitig.  An entry that finds its home bucket and the CT_HOPS buckets behind it full sends the k-mers of its windows to the exact
// overflow table (their places go on over_list); every full bucket it passed is marked, so a lookup follows the same trail.
__global__ void ct_insert_kernel(XTableView t, const unsigned long long* okmer, uint64_t n, uint32_t* over_list, uint32_t over_cap,
                                 unsigned long long* over_n, unsigned long long* n_moved)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t f = t.k - 16u;          // X's offsets in a k-mer: 0 .. f (11; k = 19 .. 26: 3 .. 10; k = 28: 12, of which an entry holds 1 .. 11)
    if (g >= n * (f + 1u)) return;
    const uint64_t p = g / (f + 1u);
    const uint32_t o = (uint32_t)(g - p * (f + 1u));
    const unsigned long long ok = okmer[p];
    if (ok >> 63) return;                  // a place no k-mer has (chains start at multiples of 16)
    const bool first = (ok >> CT_OK_FIRST) & 1ULL;
    if (o != f && !first) return;
    const uint32_t rem = (uint32_t)(ok >> (CT_OK_FIRST + 1u)) & 15u;
    const uint32_t n_win = (o < rem ? o : rem) + 1u;
    const uint64_t kf = ok & CT_MASK54, kl = okmer[p + n_win - 1] & CT_MASK54;
    CtEntry e[2];
    uint32_t keep = 0x1FFFu;
    if (!(t.k & 1u))      // even k: the k-mer j places on has X at offset o - j
        for (uint32_t j = 0; j < n_win; ++j) {
            const uint64_t kw = okmer[p + j] & CT_MASK54;
            if (kw == vg_revcomp(kw, t.k)) keep &= ~(1u << (o - j));
        }
    const int ne = ct_make_from_unitig(kf, kl, o, n_win, (uint32_t)p, e, t.k, keep);
    uint32_t* const cb = reinterpret_cast<uint32_t*>(const_cast<uint4*>(t.cb));
    for (int q = 0; q < ne; ++q) {
        const uint64_t b = ((uint64_t)ct_hash(e[q].d0) * t.n_buckets) >> 32;
        bool placed = false;
        for (uint32_t hop = 0; hop <= CT_HOPS && !placed; ++hop) {
            uint32_t* B = cb + ((b + hop) << 4);        // CtBucket: x[4], then (d1, d2, d3) per slot
            for (uint32_t s = 0; s < 4 && !placed; ++s)
                if (atomicCAS(&B[s], 0xFFFFFFFFu, e[q].d0) == 0xFFFFFFFFu) {
                    B[4 + 3 * s] = e[q].d1;
                    atomicOr(&B[5 + 3 * s], e[q].d2);      // the mark of slot 0 may already be there
                    B[6 + 3 * s] = e[q].d3;
                    placed = true;
                    if (hop) atomicAdd(n_moved, 1ULL);
                }
            if (!placed) atomicOr(&B[5], ct_mark(e[q].d0));
        }
        if (!placed) {
            const unsigned long long pos = atomicAdd(over_n, (unsigned long long)n_win);
            for (uint32_t j = 0; j < n_win; ++j)
                if (pos + j < over_cap) over_list[pos + j] = (uint32_t)(p + j);
        }
    }
}

// the overflow table: open addressing on the canonical k-mer (one cell per k-mer however many of its entries overflowed)
__global__ void ct_over_kernel(ulonglong2* over, uint32_t over_mask, const unsigned long long* okmer, const uint32_t* over_list, uint64_t n_over, uint32_t k)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_over) return;
    const uint32_t id = over_list[g];
    const uint64_t kw = okmer[id] & CT_MASK54, rc = vg_revcomp(kw, k);
    if (kw == rc) return;      // (even k: never emitted, never counted)
    const unsigned long long canon = kw < rc ? kw : rc;
    uint32_t s = xt_over_hash(canon) & over_mask;
    for (;;) {
        unsigned long long* cell = reinterpret_cast<unsigned long long*>(&over[s]);
        const unsigned long long was = atomicCAS(cell, XT_EMPTY, canon);
        if (was == XT_EMPTY) {
            cell[1] = id;
            return;
        }
        if (was == canon) return;
        s = (s + 1) & over_mask;
    }
}

__global__ void ct_over_clear_kernel(ulonglong2* over, uint32_t over_mask)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= over_mask) over[i] = make_ulonglong2(XT_EMPTY, 0ULL);
}

hipError_t launch_ctable_okmer(const TableView& t, const uint32_t* key_slot, uint32_t* pos_of_key, const uint32_t* link2, uint64_t n, bool identity,
                               unsigned long long* okmer, uint32_t* id_of_key, unsigned long long* n_unitigs, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint32_t g1 = (uint32_t)((n + 255) / 256);
    if (identity) hipLaunchKernelGGL(ct_identity_kernel, dim3(g1), dim3(256), 0, st, pos_of_key, n);
    hipLaunchKernelGGL(ct_okmer_kernel, dim3(g1), dim3(256), 0, st, t, key_slot, pos_of_key, link2, n, okmer, id_of_key, n_unitigs);
    return hipGetLastError();
}

hipError_t launch_ctable_build(const XTableView& t, const unsigned long long* okmer, uint64_t n, uint32_t* over_list, uint32_t over_cap,
                               unsigned long long* over_n, unsigned long long* n_moved, hipStream_t st)
{
    hipLaunchKernelGGL(ct_clear_kernel, dim3(4096), dim3(256), 0, st, const_cast<uint4*>(t.cb), (uint64_t)t.n_buckets + CT_HOPS);
    if (n) {
        const uint64_t m = n * (t.k - 15u);
        hipLaunchKernelGGL(ct_insert_kernel, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, st, t, okmer, n, over_list, over_cap, over_n, n_moved);
    }
    return hipGetLastError();
}

hipError_t launch_ctable_over(ulonglong2* over, uint32_t over_mask, const unsigned long long* okmer, const uint32_t* over_list, uint64_t n_over,
                              uint32_t k, hipStream_t st)
{
    hipLaunchKernelGGL(ct_over_clear_kernel, dim3((over_mask + 256) / 256), dim3(256), 0, st, over, over_mask);
    if (n_over)
        hipLaunchKernelGGL(ct_over_kernel, dim3((uint32_t)((n_over + 255) / 256)), dim3(256), 0, st, over, over_mask, okmer, over_list, n_over, k);
    return hipGetLastError();
}

// ---- counting ---------------------------------------------------------------------------------------------------------
// A grid position whose bucket is marked (some entry with this home went on to the next bucket) and whose windows are not all
// answered yet does NOT chase the trail on the spot -- that made every row wait for two or three dependent memory round trips
// on behalf of a handful of lanes (2.5 of 10.6 ms at chr20 class, VGMI_DBG=8 ablation).  It queues a 16-byte item {X, flanks,
// windows still open, hop, next bucket} instead, and the wavefront looks 64 queued items up at a time, all lanes busy, one round
// trip per batch; an item that meets another marked bucket is queued again, one that has seen CT_HOPS + 1 of them asks the exact
// overflow table.
//
// K = 19 .. 25 (round 5): the same rows, the same twelve bytes per lane, the same queues; a lane looks a 16-mer up every G = 6 (K = 19, 20: 4)
// of its bases instead of once -- the 16 bases that end G j bases into the lane's stretch, j = 0 .. 12 / G - 1 -- and asks each for the G
// windows of K bases that end in the G bases behind it.  Flanks of F = K - 16 bases; the bases behind a later X that belong to the next
// lane are zeros (no window asked for reaches them).  The ends covered are those of K = 27: stream positions 12 L - 1 .. 12 L + 10
// (K = 23 .. 26: by the pair of lanes together, three lookups per pair -- see the loop).
// One lookup position of a lane for K < 27: the 16-mer that ends D bases into the lane's twelve (D < 0: that many bases in front of them)
// and the G windows of K bases that end in the G bases behind it.  W2:W1:W0 = the lane's 48-base window (base e from the end of its
// stretch at bits [2e, 2e + 2)), U = the non-base flags of those bases, oldest first (bit 47 - e).  Every shift is a compile-time constant.
template <uint32_t K, int D, uint32_t G>
__device__ __forceinline__ void ct_position(uint32_t W0, uint32_t W1, uint32_t W2, uint64_t U, uint32_t& x, uint32_t& l, uint32_t& r, uint32_t& vm)
{
    constexpr uint32_t F = K - 16u > 11u ? 11u : K - 16u, MF = (1u << (2u * F)) - 1u, sh = (uint32_t)(24 - 2 * D), ub = (uint32_t)(36 + D - (int)K);
    static_assert(sh >= 2u && sh <= 32u && G >= 1u && G <= F + 1u && 36 + D - (int)K >= 0, "position outside the lane's window");
    const uint64_t Uj = U >> ub;                    // the window that ends w bases behind X spans bits w .. w + K - 1
    const uint32_t A = (uint32_t)Uj & ((1u << K) - 1u), a = ((uint32_t)(Uj >> K) & ((1u << (G - 1u)) - 1u)) << 1;
    const uint32_t bad_b = A ? (0xFFFFFFFFu >> __builtin_clz(A)) : 0u;
    vm = ~(a | (0u - a) | bad_b) & ((1u << G) - 1u);
    if constexpr (sh < 32u) {
        x = __builtin_amdgcn_alignbit(W1, W0, sh);
        l = __builtin_amdgcn_alignbit(W2, W1, sh) & MF;
    } else {
        x = W1;
        l = W2 & MF;
    }
    if constexpr (sh >= 2u * F) r = (W0 >> (sh - 2u * F)) & MF;
    else r = (W0 << (2u * F - sh)) & MF;             // the bases behind X that belong to the next lane: zeros (no window asked for reaches them)
}

template <uint32_t K, bool DEFER>
__device__ __forceinline__ void countc_body(const RowParams& p, const XTableView& xt, const CtDefer& df)
{
    constexpr uint32_t F = K - 16u > 11u ? 11u : K - 16u, EX = K - 16u - F;      // ct_flank(K), ct_excess(K): k = 28 keeps flanks of 11 and answers 11 windows an entry
    constexpr uint32_t G = K == 27u ? 12u : (K <= 20u ? 4u : 6u), NP = 12u / G;
    static_assert(K >= 19u && K <= 28u, "context table: k = 19 .. 28");
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[2048];     // position LUT of count27_kernel (stage_lut27 layout)
    __shared__ __attribute__((aligned(16))) uint2 s_runs[4][CT_RUNQ];
    __shared__ __attribute__((aligned(16))) uint4 s_pend[4][CT_PENDQ];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    for (uint32_t i = tid; i < 2048; i += blockDim.x) {
        const uint32_t set = i >> 10, b = (i >> 8) & 3u, c = vg_nt4(i & 255u);
        s_lut[i] = (uint16_t)(((c & 3u) << (2 * (3 - b))) | ((c >> 2) << ((set ? 12 : 8) + b)));
    }
    __syncthreads();
    uint2* const runs = s_runs[wave];
    uint4* const pend = s_pend[wave];

    const uint64_t n_bytes = p.n_bytes_dev ? *p.n_bytes_dev : p.n_bytes;
    const uint64_t total_rows = n_bytes / 768;            // complete rows; the ragged tail goes to rows_kernel (launch_count)
    const uint64_t total_waves = (uint64_t)gridDim.x * 4;
    const uint64_t rpw = (total_rows + total_waves - 1) / total_waves;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    const uint64_t r0 = gw * rpw;
    const uint64_t r1 = r0 + rpw < total_rows ? r0 + rpw : total_rows;
    if (r0 >= r1) return;

    const uint32_t my_run = lane / 12u, my_win = lane % 12u;
    uint32_t run_head = 0, run_n = 0, pend_head = 0, pend_n = 0;
    auto ring = [](uint32_t pos) -> uint32_t { return pos >= CT_RUNQ ? pos - CT_RUNQ : pos; };
    auto below = [](uint64_t m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };

    // 12 ASCII bytes -> 24 bits of bases (first base most significant) + 12 non-base flags
    auto encode12 = [&](uint32_t w0, uint32_t w1, uint32_t w2, uint32_t& be, uint32_t& inv) {
        auto enc4 = [&](uint32_t w, uint32_t set) -> uint32_t {
            return (uint32_t)s_lut[set * 1024u + (w & 0xFFu)] | s_lut[set * 1024u + 256u + ((w >> 8) & 0xFFu)] |
                   s_lut[set * 1024u + 512u + ((w >> 16) & 0xFFu)] | s_lut[set * 1024u + 768u + (w >> 24)];
        };
        const uint32_t g0 = enc4(w0, 0), g1 = enc4(w1, 1), g2 = enc4(w2, 0);
        be = (g0 & 0xFFu) << 16 | (g1 & 0xFFu) << 8 | (g2 & 0xFFu);
        inv = ((g0 | g1) >> 8) | (g2 & 0xF00u);       // g0: bits 8..11 -> 0..3, g1: 12..15 -> 4..7, g2: 8..11
    };
    auto ror1 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xF, 0xF, false); };

    // A drain step turns up to 5 queued runs {id0, hit windows | dir << 12} into counter updates, a lane per (run, window): the
    // hits of a run -- and of the runs queued next to it, which continue the same unitig -- are neighbours in the counter array
    // and leave as one or two atomic requests.  No return value: nothing waits for them.  (Built, measured on the same box and
    // dropped -- the plain form is the fastest: chains of counters aligned to 64-byte sectors 8.62 against 8.57 ms at chr20 class
    // (gpurun_out/r4d); a lane per aligned PAIR of counters and one 64-bit add for both 9.19 = 9.19 (r4g); counters as a difference
    // array over the places -- +1 / -1 at the two ends of a stretch of hits, prefix sums at read-out -- 8.82 against 8.44 (r4i).)
    // DEFER (round 6; VERDICT r5 #1b): the queued runs do not become atomics here.  They leave the wavefront's ring 64 at a time as ONE
    // coalesced 512-byte store into a chunk of CTD_CHUNK records the wavefront reserved in `df.rec` (one returning atomic per chunk);
    // ctd_scatter_kernel / ctd_accumulate_kernel (vgmi_ctdefer.hip) add them up behind this kernel, by counter region, in LDS.
    // c = min(255, sum) does not depend on the order (src/fastq_kmer.cpp:128-139), so the counters are the same.  A chunk that does not
    // fit the buffer any more is not written: from then on this wavefront's runs leave as atomics, as in the plain kernel.
    uint32_t chunk_at = 0, chunk_used = CTD_CHUNK;
    bool nofit = false;
    auto drain = [&]() {
        const uint32_t take = run_n < 5u ? run_n : 5u;
        const bool have = my_run < take;
        uint2 q = make_uint2(0, 0);
        if (have) q = runs[ring(run_head + my_run)];
        run_head = ring(run_head + take);
        run_n -= take;
        if (have && ((q.y >> my_win) & 1u) && !(VG_DBG(p.dbg) & 2u)) {
            const uint32_t id = (q.y & 0x1000u) ? q.x + my_win : q.x - my_win;
            __hip_atomic_fetch_add(xt.counts + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // up to 64 queued runs -> the wavefront's chunk (lanes without a run write a null record: no window set)
    auto flush = [&]() {
        if (chunk_used == CTD_CHUNK && !nofit) {
            uint32_t at = 0;
            if (lane == 0) at = __hip_atomic_fetch_add(df.cursor, CTD_CHUNK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (at > df.cap - CTD_CHUNK) nofit = true;      // (cap is a multiple of CTD_CHUNK and >= CTD_CHUNK; a cursor that wrapped 2^32 cannot happen: cap + waves * CTD_CHUNK < 2^32, ctd_scratch_bytes)
            else {
                chunk_at = at;
                chunk_used = 0;
            }
        }
        if (nofit) {
            drain();
            return;
        }
        const uint32_t take = run_n < 64u ? run_n : 64u;
        uint2 q = make_uint2(0, 0);
        if (lane < take) q = runs[ring(run_head + lane)];
        df.rec[chunk_at + chunk_used + lane] = q;
        run_head = ring(run_head + take);
        run_n -= take;
        chunk_used += 64u;
    };
    // the three places the plain kernel drains at: with the loads just issued (keep the ring short), for room in the ring, at the end
    auto relieve_early = [&]() {
        if constexpr (DEFER) { while (run_n >= 64u) flush(); }
        else { while (run_n >= 5u) drain(); }
    };
    auto relieve_for = [&](uint32_t n) {
        if constexpr (DEFER) { while (run_n + n > CT_RUNQ) flush(); }
        else { while (run_n + n > CT_RUNQ) drain(); }
    };

    // One bucket against one context per lane: the matching windows of each entry become a run; returns the windows answered and
    // whether the bucket is marked.  `early`: called with the loads just issued -- the queued runs leave while they are in flight.
    auto look = [&](bool act, uint64_t bucket, uint32_t cx, uint32_t cl, uint32_t cr, uint32_t vs, bool early, uint32_t& found, bool& marked) {
        // the four X of the bucket: most positions (59 % at chr20 class, 94 % at whole-genome class) end here, after ONE load
        const uint32_t* const Bk = reinterpret_cast<const uint32_t*>(xt.cb + (bucket << 2));
        uint4 xs = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (act) xs = *reinterpret_cast<const uint4*>(Bk);
        if (early) relieve_early();
        // ... the rest of the entries whose X is this one (the line is in the vector cache now), and slot 0's of a full bucket for its mark
        const bool q0 = xs.x == cx, q1 = xs.y == cx, q2 = xs.z == cx, q3 = xs.w == cx, full = xs.w != 0xFFFFFFFFu;
        CtEntry c0 = {0xFFFFFFFFu, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        if (q0 || full) c0 = CtEntry{xs.x, Bk[4], Bk[5], Bk[6]};
        if (q1) c1 = CtEntry{xs.y, Bk[7], Bk[8], Bk[9]};
        if (q2) c2 = CtEntry{xs.z, Bk[10], Bk[11], Bk[12]};
        if (q3) c3 = CtEntry{xs.w, Bk[13], Bk[14], Bk[15]};
        const uint32_t h0 = ct_match(c0, cx, cl, cr, F, EX) & vs, h1 = ct_match(c1, cx, cl, cr, F, EX) & vs;
        const uint32_t h2 = ct_match(c2, cx, cl, cr, F, EX) & vs, h3 = ct_match(c3, cx, cl, cr, F, EX) & vs;
        const uint64_t m0 = __ballot(h0 != 0), m1 = __ballot(h1 != 0), m2 = __ballot(h2 != 0), m3 = __ballot(h3 != 0);
        const uint32_t n = (uint32_t)(__builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
        if (n) {
            relieve_for(n);
            // lane order: the entries of neighbouring grid positions (the same unitig, 12 counters on) stay neighbours in the ring
            uint32_t pos = run_head + run_n + below(m0) + below(m1) + below(m2) + below(m3);
            if (h0) runs[ring(pos++)] = make_uint2(c0.d3, h0 | ((c0.d2 >> 12) & 0x1000u));
            if (h1) runs[ring(pos++)] = make_uint2(c1.d3, h1 | ((c1.d2 >> 12) & 0x1000u));
            if (h2) runs[ring(pos++)] = make_uint2(c2.d3, h2 | ((c2.d2 >> 12) & 0x1000u));
            if (h3) runs[ring(pos++)] = make_uint2(c3.d3, h3 | ((c3.d2 >> 12) & 0x1000u));
            run_n += n;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
        }
        found = h0 | h1 | h2 | h3;
        marked = full && (c0.d2 & ct_mark(cx));
    };
    // queue the contexts that go on: {X, L | open windows 0..9 << 22, R | open windows 10..11 << 22 | hop << 24, bucket}
    auto push = [&](bool on, uint32_t cx, uint32_t cl, uint32_t cr, uint32_t open, uint32_t hop, uint32_t bucket) {
        const uint64_t m = __ballot(on);
        if (m == 0) return;
        if (on) {
            uint32_t pos = pend_head + pend_n + below(m);
            pos = pos >= CT_PENDQ ? pos - CT_PENDQ : pos;
            pend[pos] = make_uint4(cx, cl | (open & 0x3FFu) << 22, cr | (open >> 10) << 22 | hop << 24, bucket);
        }
        pend_n += (uint32_t)__builtin_popcountll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    };
    auto pending_batch = [&]() {
        const uint32_t take = pend_n < 64u ? pend_n : 64u;
        const bool act = lane < take;
        uint4 it = make_uint4(0, 0, 0, 0);
        if (act) {
            const uint32_t pos = pend_head + lane;
            it = pend[pos >= CT_PENDQ ? pos - CT_PENDQ : pos];
        }
        pend_head = pend_head + take >= CT_PENDQ ? pend_head + take - CT_PENDQ : pend_head + take;
        pend_n -= take;
        const uint32_t cx = it.x, cl = it.y & CT_M22, cr = it.z & CT_M22, hop = (it.z >> 24) & 15u;
        const uint32_t vs = (it.y >> 22) | ((it.z >> 12) & 0xC00u);
        uint32_t found;
        bool marked;
        look(act, it.w, cx, cl, cr, vs, false, found, marked);
        const uint32_t open = vs & ~found;
        const bool on = marked && open != 0 && !(VG_DBG(p.dbg) & 8u);
        if (on && hop == CT_HOPS && !(VG_DBG(p.dbg) & 4u)) {       // CT_HOPS + 1 marked buckets: what is left may sit in the exact overflow table
            uint32_t rest = open;
            while (rest) {
                const uint32_t s = (uint32_t)__builtin_ctz(rest);
                rest &= rest - 1u;
                const uint32_t id = xt_over_find(xt, ct_window_kmer(cx, cl, cr, s, K));
                if (id != CT_NONE) __hip_atomic_fetch_add(xt.counts + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        push(on && hop < CT_HOPS, cx, cl, cr, open, hop + 1u, it.w + 1u);
    };

    // halo: the row in front of the range
    uint32_t pr1_be = 0, pr2_be = 0, pr3_be = 0, pr1_inv = 0xFFFu, pr2_inv = 0xFFFu, pr3_inv = 0xFFFu;
    const uint8_t* const bases = p.bases;
    const uint64_t rs = r0 > 0 ? r0 - 1 : r0;
    auto load_row = [&](uint64_t r, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
        // wave-uniform row base in scalar registers + a 32-bit lane offset: no 64-bit address held in vector registers
        const uint64_t ro = r * 768;
        const uint64_t rb = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ro) |
                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ro >> 32)) << 32;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(bases + rb + lane * 12u);
        w0 = __builtin_nontemporal_load(src);
        w1 = __builtin_nontemporal_load(src + 1);
        w2 = __builtin_nontemporal_load(src + 2);
    };
    uint32_t n0, n1, n2;
    load_row(rs, n0, n1, n2);
    for (uint64_t r = rs; r < r1; ++r) {
        const uint32_t w0 = n0, w1 = n1, w2 = n2;
        uint32_t be, inv;
        encode12(w0, w1, w2, be, inv);
        const uint32_t a1_be = ror1(be), a1_inv = ror1(inv);
        const uint32_t a2_be = ror1(a1_be), a2_inv = ror1(a1_inv);
        const uint32_t a3_be = ror1(a2_be), a3_inv = ror1(a2_inv);
        const uint32_t be1 = lane >= 1 ? a1_be : pr1_be, be2 = lane >= 2 ? a2_be : pr2_be, be3 = lane >= 3 ? a3_be : pr3_be;
        const uint32_t i1 = lane >= 1 ? a1_inv : pr1_inv, i2 = lane >= 2 ? a2_inv : pr2_inv, i3 = lane >= 3 ? a3_inv : pr3_inv;
        pr1_be = a1_be; pr2_be = a2_be; pr3_be = a3_be; pr1_inv = a1_inv; pr2_inv = a2_inv; pr3_inv = a3_inv;
        if (r + 1 < r1) load_row(r + 1, n0, n1, n2);     // the next row is in flight while this one is worked on
        if (r < r0) continue;      // warm-up row: halo only

        {   // empty-read check (reference: assert(len > 0), src/kmer.cpp:124): two adjacent non-bases are necessary
            const uint32_t adj = inv & ((inv << 1) | (i1 >> 11));
            if (__builtin_expect(__ballot(adj != 0) != 0, 0) && adj) {
                const uint64_t base_off = r * 768 + lane * 12u;
                for (uint32_t t = 0; t < 12; ++t) {
                    if (!((adj >> t) & 1u)) continue;
                    const uint64_t o = base_off + t;
                    if (bases[o] == '\n' && (o == 0 || bases[o - 1] == '\n')) atomicOr(p.status, 1u);
                }
            }
        }
        // 48-base window (see count27_kernel::scan_probe): base e from the end of the own chunk at bits [2e, 2e + 2) of W2:W1:W0
        // -- 0..11 own chunk, 12..27 the grid 16-mer X, 28..38 the 11 bases in front of it
        const uint32_t W0 = (be1 << 24) | be, W1 = (be2 << 16) | (be1 >> 8), W2 = (be3 << 8) | (be2 >> 16);
        if constexpr (K == 27u) {
            const uint32_t B = (i3 >> 9) | (i2 << 3) | (i1 << 15);
            const uint32_t a = (inv << 1) & 0xFFFu;
            const uint32_t bad_b = B ? (0xFFFFFFFFu >> __builtin_clz(B)) : 0u;
            const uint32_t vm = ~(a | (0u - a) | bad_b) & 0xFFFu;       // bit w: the window that ends w bases behind X is made of bases
            const bool act = B < 2048u && vm != 0;                       // X itself is 16 bases
            uint32_t cx, cl, cr, vs;
            ct_orient(__builtin_amdgcn_alignbit(W1, W0, 24), __builtin_amdgcn_alignbit(W2, W1, 24) & CT_M22, (W0 >> 2) & CT_M22, vm, cx, cl, cr, vs);
            const uint32_t b0 = (uint32_t)(((uint64_t)ct_hash(cx) * xt.n_buckets) >> 32);
            uint32_t found;
            bool marked;
            look(act, b0, cx, cl, cr, vs, true, found, marked);
            push(marked && (vs & ~found) != 0 && !(VG_DBG(p.dbg) & 8u), cx, cl, cr, vs & ~found, 1u, b0 + 1u);
            while (pend_n >= 64u) pending_batch();
        } else {
            // the non-base flags of the 48 bases, oldest first (bit u = 47 - e)
            const uint64_t U = (uint64_t)(i3 | i2 << 12 | (i1 & 0xFFu) << 24) | (uint64_t)((i1 >> 8) | inv << 4) << 32;
            auto probe = [&](uint32_t xr, uint32_t lr, uint32_t rr, uint32_t vm) {
                uint32_t cx, cl, cr, vs;
                ct_orient(xr, lr, rr, vm, cx, cl, cr, vs, F, EX);
                const uint32_t b0 = (uint32_t)(((uint64_t)ct_hash(cx) * xt.n_buckets) >> 32);
                uint32_t found;
                bool marked;
                look(vm != 0, b0, cx, cl, cr, vs, true, found, marked);
                push(marked && (vs & ~found) != 0 && !(VG_DBG(p.dbg) & 8u), cx, cl, cr, vs & ~found, 1u, b0 + 1u);
                while (pend_n >= 64u) pending_batch();
            };
            if constexpr (K == 28u) {
                // K = 28: an entry answers the eleven windows that end 1 .. 11 bases behind its X, so the schedule of K = 26 (eleven windows too,
                // ending 0 .. 10 behind) with every X one base earlier: the even lane's X ends 2 bases in front of its twelve (ends 12 L - 1 ..
                // 12 L + 9), the odd lane's first one 3 bases in front of its own (the even lane's last two ends and its own first nine), its
                // second one at its own base 8 (its last two ends)
                const bool odd_lane = (lane & 1u) != 0;
                uint32_t xe, le, re, ve, xo, lo_, ro, vo;
                ct_position<K, -1, 12u>(W0, W1, W2, U, xe, le, re, ve);
                ct_position<K, -2, 12u>(W0, W1, W2, U, xo, lo_, ro, vo);
                probe(odd_lane ? xo : xe, odd_lane ? lo_ : le, odd_lane ? ro : re, (odd_lane ? vo : ve) & ~1u);
                ct_position<K, 9, 3u>(W0, W1, W2, U, xo, lo_, ro, vo);
                probe(xo, lo_, ro, odd_lane ? vo & ~1u : 0u);
            } else if constexpr (K >= 23u) {
                // K = 23 .. 26: an entry answers NW = K - 15 >= 8 windows, so a PAIR of lanes (24 ends) needs three lookups, not four: the
                // even lane asks for the NW ends from the top of its twelve, the odd lane first for the even lane's 12 - NW last ends
                // together with its own first ones (the 16-mer that ends 12 - NW bases in front of its stretch: all of it is in the odd
                // lane's window), then for what is left of its own twelve
                constexpr uint32_t NW = F + 1u, G1 = 24u - 2u * NW;
                constexpr int BACK = 12 - (int)NW, D1 = 2 * (int)NW - 12;
                const bool odd_lane = (lane & 1u) != 0;
                uint32_t xe, le, re, ve, xo, lo_, ro, vo;
                ct_position<K, 0, NW>(W0, W1, W2, U, xe, le, re, ve);
                ct_position<K, -BACK, NW>(W0, W1, W2, U, xo, lo_, ro, vo);
                probe(odd_lane ? xo : xe, odd_lane ? lo_ : le, odd_lane ? ro : re, odd_lane ? vo : ve);
                ct_position<K, D1, G1>(W0, W1, W2, U, xo, lo_, ro, vo);
                probe(xo, lo_, ro, odd_lane ? vo : 0u);
            } else {
                // K = 19 .. 22: a 16-mer every G = 6 (K <= 20: 4) of the lane's bases
                uint32_t xr, lr, rr, vm;
                ct_position<K, 0, G>(W0, W1, W2, U, xr, lr, rr, vm);
                probe(xr, lr, rr, vm);
                ct_position<K, (int)G, G>(W0, W1, W2, U, xr, lr, rr, vm);
                probe(xr, lr, rr, vm);
                if constexpr (NP == 3u) {
                    ct_position<K, 2 * (int)G, G>(W0, W1, W2, U, xr, lr, rr, vm);
                    probe(xr, lr, rr, vm);
                }
            }
        }
    }
    while (pend_n) pending_batch();
    if constexpr (DEFER) {
        while (run_n) flush();
        if (!nofit)      // the rest of the wavefront's last chunk: null records
            for (; chunk_used < CTD_CHUNK; chunk_used += 64u) df.rec[chunk_at + chunk_used + lane] = make_uint2(0, 0);
    } else {
        while (run_n) drain();
    }
}

__global__ __launch_bounds__(256, 8) void count27c_kernel(RowParams p, XTableView xt) { countc_body<27u, false>(p, xt, CtDefer{}); }
template <uint32_t K>
__global__ __launch_bounds__(256, 8) void countkc_kernel(RowParams p, XTableView xt) { countc_body<K, false>(p, xt, CtDefer{}); }
// the same kernels with their runs of hits written out instead of counted (the counting: vgmi_ctdefer.hip)
template <uint32_t K>
__global__ __launch_bounds__(256, 8) void countkc_defer_kernel(RowParams p, XTableView xt, CtDefer df) { countc_body<K, true>(p, xt, df); }

template <uint32_t K>
static void launch_defer_k(const RowParams& p, const XTableView& t, uint32_t grid, hipStream_t st, const CtDefer& d)
{
    hipLaunchKernelGGL(countkc_defer_kernel<K>, dim3(grid), dim3(256), 0, st, p, t, d);
}

hipError_t launch_count27c(const RowParams& p, const XTableView& t, uint32_t n_cu, hipStream_t st, const CtDefer* defer)
{
    static const int wgs_env = [] {      // workgroups per CU (20 KB of LDS each: 8 fit); VGMI_CT_WGS for A/B
        const char* e = getenv("VGMI_CT_WGS");
        const int v = e ? atoi(e) : 0;
        return v < 1 ? 0 : v > 8 ? 8 : v;
    }();
    // measured, kernel ms: chr20 class (2 GB of buckets) 5 8.96, 6 8.31, 7 8.50, 8 8.62 (gpurun_out/r4d) -- a table in HBM is bound by requests
    // in flight, not by wavefronts; a table of a few megabytes (k = 26 / 28 on a small graph: it lives in the L2s) is bound by instruction
    // issue and wants every slot: k = 28, 2e7 reads 4 4.59, 5 4.23, 6 4.05, 7 3.93, 8 3.87 (round 6)
    const uint32_t wgs = wgs_env ? (uint32_t)wgs_env : ((uint64_t)t.n_buckets * 64u <= (32u << 20) ? 8u : 6u);
    if (defer && defer->rec) {
        switch (t.k) {
            case 27: launch_defer_k<27u>(p, t, n_cu * wgs, st, *defer); break;
            case 28: launch_defer_k<28u>(p, t, n_cu * wgs, st, *defer); break;
            case 26: launch_defer_k<26u>(p, t, n_cu * wgs, st, *defer); break;
            case 25: launch_defer_k<25u>(p, t, n_cu * wgs, st, *defer); break;
            case 24: launch_defer_k<24u>(p, t, n_cu * wgs, st, *defer); break;
            case 23: launch_defer_k<23u>(p, t, n_cu * wgs, st, *defer); break;
            case 22: launch_defer_k<22u>(p, t, n_cu * wgs, st, *defer); break;
            case 21: launch_defer_k<21u>(p, t, n_cu * wgs, st, *defer); break;
            case 20: launch_defer_k<20u>(p, t, n_cu * wgs, st, *defer); break;
            case 19: launch_defer_k<19u>(p, t, n_cu * wgs, st, *defer); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (t.k) {
        case 27: hipLaunchKernelGGL(count27c_kernel, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 25: hipLaunchKernelGGL(countkc_kernel<25u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 23: hipLaunchKernelGGL(countkc_kernel<23u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 21: hipLaunchKernelGGL(countkc_kernel<21u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 19: hipLaunchKernelGGL(countkc_kernel<19u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        // even k: the windows-of-bases rule here, the reference's run counter in the pass ahead of this launch (even_debit_kernel, vgmi_kernels.hip)
        case 28: hipLaunchKernelGGL(countkc_kernel<28u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 26: hipLaunchKernelGGL(countkc_kernel<26u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 24: hipLaunchKernelGGL(countkc_kernel<24u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 22: hipLaunchKernelGGL(countkc_kernel<22u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        case 20: hipLaunchKernelGGL(countkc_kernel<20u>, dim3(n_cu * wgs), dim3(256), 0, st, p, t); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vgk
