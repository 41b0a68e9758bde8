// vgmi_xtable.hip -- the large-graph read-counting kernel over a table keyed by the read's GRID 16-mer (k = 27).
//
// Same reference behaviour as count27_kernel (vgmi_kernels.hip): src/kmer.cpp:110-149 (emitter, odd k: a window counts
// iff its 27 bases are bases), :140-142 (membership), src/fastq_kmer.cpp:128-139 (saturating count).
//
// Why another table.  The HBM-resident kernel runs AT the memory system's request rate (58 G memory-side requests/s,
// DESIGN.md section 6.2): 37.8 requests per read, of which 13.6 are table lines -- the twelve candidate k-mers of a run
// hash to two or three minimiser buckets of two lines each -- and 10.5 are atomic requests on per-slot counters
// scattered the same way.  Here every graph k-mer is stored under EACH of its twelve canonical 16-mers:
//     line   = (h(X) * n_lines) >> 32, h a bijection of the 32-bit canonical 16-mer X (so line + tag identify X exactly)
//     entry  = { j' : 4   offset of X from the k-mer's end, counted in X's canonical orientation
//                f  : 22  the 11 bases of the k-mer outside X, in that orientation
//                tag: the low bits of h(X) that tell the 16-mers of one line apart
//                id : the k-mer's counter id }                                  8 bytes, 16 per 128-byte line
//     slot   = j' (0..11); a second k-mer with the same (line, j') -- the other allele, another X of the line -- takes one of
//              the spill slots 12..15, then slot j' / the spill slots of the next line, XT_HOPS lines at most (a lookup goes
//              on to the next line only when all four spill slots are taken; tag covers the h-values of XT_HOPS + 1
//              consecutive lines, so a match there is still exact).  16-mers of repeats, with more k-mers than that, send
//              the k-mer to a small exact overflow table keyed by the k-mer itself.
// A read position's grid 16-mer is one of those twelve for every k-mer that contains it, so a candidate run's twelve
// lookups are ONE line (plus its spill slots) by construction, whatever strand the read is on: (tag, j', f) of a window
// and of its reverse complement are the same triple.  Counters are dense by id; with ids numbered along the graph's
// paths (xtable_number_*) a run's hits are neighbours and leave as one or two atomic requests.
//
// The kernel is compiler-scheduled on purpose: this regime is bound by memory requests (VALU 12 % busy in count27_kernel),
// so it runs many small workgroups and lets the hardware overlap them instead of hand-counting vmcnt.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_xtable.h"

namespace vgk {

#define XT_RUNQ 96u          // run ring per wavefront (entries of 16 bytes): a row adds <= 64, a drain step takes 5

// ---- build ----------------------------------------------------------------------------------------------------------
__global__ void xtable_clear_kernel(XTableView t)
{
    const uint64_t n = 16ULL * ((uint64_t)t.n_lines + XT_HOPS);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) t.lines[i] = XT_EMPTY;
}

// one thread per (key, offset): canonical k-mers come from the compact table image (slots8[key_slot[i]]).  A pair that finds
// its home line and the XT_HOPS lines behind it full (slot j' and the four spill slots of each) puts the key's index on
// over_list: those keys go into the exact overflow table (xtable_over_kernel).  Slots never empty again, so a lookup that
// follows the same rule sees the same full lines and ends in that table too.
__global__ void xtable_insert_kernel(XTableView t, const unsigned long long* slots8, const uint32_t* key_slot, const uint32_t* id_of_key,
                                     uint64_t n_keys, uint32_t* over_list, uint32_t over_cap, unsigned long long* over_n)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_keys * 12) return;
    const uint64_t i = g / 12;
    const uint32_t w = (uint32_t)(g - i * 12);
    const uint64_t kmer = slots8[key_slot[i]] & ((1ULL << 54) - 1);
    uint64_t line, want;
    xt_key(t, kmer, w, line, want);
    const uint64_t id = id_of_key ? id_of_key[i] : i;
    auto insert = [&](uint64_t wnt) {
        const unsigned long long e = (wnt & ((1ULL << t.id_shift) - 1)) | id << t.id_shift;
        const uint32_t j = (uint32_t)wnt & 15u;
        for (uint32_t hop = 0; hop <= XT_HOPS; ++hop) {
            unsigned long long* L = t.lines + ((line + hop) << 4);
            if (atomicCAS(&L[j], XT_EMPTY, e) == XT_EMPTY) return;
            for (uint32_t s = 12; s < 16; ++s)
                if (atomicCAS(&L[s], XT_EMPTY, e) == XT_EMPTY) return;
        }
        const unsigned long long pos = atomicAdd(over_n, 1ULL);
        if (pos < over_cap) over_list[pos] = (uint32_t)i;
    };
    insert(want);
    // a 16-mer that is its own reverse complement reads the same on both strands, but the offset and the flank do not:
    // the other strand's window finds the k-mer under the mirrored pair
    const uint32_t x = (uint32_t)(kmer >> (2 * w));
    if (x == vg_revcomp16(x)) {
        const uint32_t j = (uint32_t)want & 15u, f = (uint32_t)(want >> 4) & 0x3FFFFFu;
        insert((want & ~0x3FFFFFFULL) | (uint64_t)(11u - j) | (uint64_t)((uint32_t)vg_revcomp(f, 11)) << 4);
    }
}

// the overflow table: open addressing on the canonical k-mer, one entry per key however many of its pairs overflowed
__global__ void xtable_over_clear_kernel(ulonglong2* over, uint32_t over_mask)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= over_mask) over[i] = make_ulonglong2(XT_EMPTY, 0ULL);
}

__global__ void xtable_over_kernel(ulonglong2* over, uint32_t over_mask, const unsigned long long* slots8, const uint32_t* key_slot,
                                   const uint32_t* id_of_key, const uint32_t* over_list, uint64_t n_over)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_over) return;
    const uint32_t i = over_list[g];
    const unsigned long long kmer = slots8[key_slot[i]] & ((1ULL << 54) - 1);
    uint32_t s = xt_over_hash(kmer) & over_mask;
    for (;;) {
        unsigned long long* cell = reinterpret_cast<unsigned long long*>(&over[s]);
        const unsigned long long was = atomicCAS(cell, XT_EMPTY, kmer);
        if (was == XT_EMPTY) {
            cell[1] = id_of_key ? id_of_key[i] : i;
            return;
        }
        if (was == kmer) return;      // another pair of the same key
        s = (s + 1) & over_mask;
    }
}

// ---- counter ids in path order --------------------------------------------------------------------------------------
// Any numbering of the keys is correct; this one makes the k-mers a read meets one after the other neighbours in the
// counter array, so that a candidate run's hits leave as one or two atomic requests instead of one each (DESIGN.md
// section 6.2).  The key set is taken as a bidirected de Bruijn graph: k-mer K has a unique neighbour on a side if exactly
// one of the four one-base extensions on that side is a key, and the link counts when the neighbour sees K the same way
// (unitigs).  Every chain end walks its chain; the end with the smaller key index numbers it.  Keys on cycles get what
// is left.  Side 0 = left (predecessors of the canonical k-mer), 1 = right.
#define XN_NONE 0xFFFFFFFFu

__global__ void xnum_links_kernel(XTableView t, const unsigned long long* slots8, const uint32_t* key_slot, uint64_t n, uint32_t* link)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    const uint64_t i = g >> 1;
    const uint32_t side = (uint32_t)g & 1u;
    const uint64_t mask = (1ULL << 54) - 1;
    const uint64_t K = slots8[key_slot[i]] & mask;
    uint32_t found = XN_NONE, cnt = 0, enter = 0;
    for (uint64_t b = 0; b < 4; ++b) {
        const uint64_t N = side ? ((K << 2) | b) & mask : (K >> 2) | (b << 52);
        const uint32_t id = xt_find(t, N);
        if (id == XN_NONE) continue;
        ++cnt;
        found = id;
        const bool flipped = N > vg_revcomp(N, 27);        // the neighbour's canonical form is the other strand
        enter = side ? (flipped ? 1u : 0u) : (flipped ? 0u : 1u);
    }
    // a k-mer that follows itself (homopolymers, K next to its own reverse complement) stays a chain end
    link[g] = (cnt == 1 && found != (uint32_t)i && found < 0x7FFFFFFFu) ? (found | enter << 31) : XN_NONE;
}

__global__ void xnum_mutual_kernel(const uint32_t* link, uint32_t* link2, uint64_t n)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    const uint32_t l = link[g];
    uint32_t out = XN_NONE;
    if (l != XN_NONE) {
        const uint64_t nb = l & 0x7FFFFFFFu, es = l >> 31;
        const uint32_t back = link[2 * nb + es];
        if (back != XN_NONE && (back & 0x7FFFFFFFu) == (uint32_t)(g >> 1) && (back >> 31) == ((uint32_t)g & 1u)) out = l;
    }
    link2[g] = out;
}

// one thread per (key, side) that is a chain end on that side
__global__ void xnum_walk_kernel(const uint32_t* link2, uint64_t n, uint32_t* id_of_key, unsigned long long* cursor)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * n) return;
    if (link2[g] != XN_NONE) return;               // not an end on this side
    const uint32_t start = (uint32_t)(g >> 1), s0 = (uint32_t)g & 1u;
    uint32_t cur = start, out = s0 ^ 1u;
    uint64_t len = 1;
    for (;;) {
        const uint32_t l = link2[2ull * cur + out];
        if (l == XN_NONE || len > n) break;
        cur = l & 0x7FFFFFFFu;
        out = (l >> 31) ^ 1u;
        ++len;
    }
    // the far end is (cur, side `out`).  One of the two ends owns the chain.
    const bool own = start < cur || (start == cur && (s0 == 0u || link2[2ull * start] != XN_NONE));
    if (!own || len > n) return;
    const uint64_t base = atomicAdd(cursor, (unsigned long long)len);
    cur = start;
    out = s0 ^ 1u;
    for (uint64_t pos = 0; pos < len; ++pos) {
        id_of_key[cur] = (uint32_t)(base + pos);
        const uint32_t l = link2[2ull * cur + out];
        if (l == XN_NONE) break;
        cur = l & 0x7FFFFFFFu;
        out = (l >> 31) ^ 1u;
    }
}

__global__ void xnum_rest_kernel(uint64_t n, uint32_t* id_of_key, unsigned long long* cursor)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (id_of_key[i] == XN_NONE) id_of_key[i] = (uint32_t)atomicAdd(cursor, 1ULL);    // keys on cycles
}

// every id in [0, n) exactly once?  (mark = n zeroed words; status bit 16 otherwise: the caller then keeps id = key index)
__global__ void xnum_check_kernel(const uint32_t* id_of_key, uint64_t n, uint32_t* mark, uint32_t* status)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = id_of_key[i];
    if (id >= n || atomicAdd(&mark[id], 1u) != 0u) atomicOr(status, 16u);
}

hipError_t launch_xtable_number(const XTableView& t, const unsigned long long* slots8, const uint32_t* key_slot, uint64_t n, uint32_t* link,
                                uint32_t* link2, uint32_t* id_of_key, unsigned long long* cursor, uint32_t* mark, uint32_t* status,
                                hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint32_t g2 = (uint32_t)((2 * n + 255) / 256), g1 = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(xnum_links_kernel, dim3(g2), dim3(256), 0, st, t, slots8, key_slot, n, link);
    hipLaunchKernelGGL(xnum_mutual_kernel, dim3(g2), dim3(256), 0, st, link, link2, n);
    hipLaunchKernelGGL(xnum_walk_kernel, dim3(g2), dim3(256), 0, st, link2, n, id_of_key, cursor);
    hipLaunchKernelGGL(xnum_rest_kernel, dim3(g1), dim3(256), 0, st, n, id_of_key, cursor);
    hipLaunchKernelGGL(xnum_check_kernel, dim3(g1), dim3(256), 0, st, id_of_key, n, mark, status);
    return hipGetLastError();
}

// ---- counting ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) const uint16_t xlds_u16;

__global__ __launch_bounds__(256, 8) void count27x_kernel(RowParams p, XTableView xt)
{
    constexpr uint32_t MASK_HI = (1u << (2 * 27 - 32)) - 1;
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[2048];     // position LUT of count27_kernel (stage_lut27 layout)
    __shared__ __attribute__((aligned(16))) uint4 s_runs[4][XT_RUNQ];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    for (uint32_t i = tid; i < 2048; i += blockDim.x) {
        const uint32_t set = i >> 10, b = (i >> 8) & 3u, c = vg_nt4(i & 255u);
        s_lut[i] = (uint16_t)(((c & 3u) << (2 * (3 - b))) | ((c >> 2) << ((set ? 12 : 8) + b)));
    }
    __syncthreads();
    uint4* const runs = s_runs[wave];

    const uint64_t n_bytes = p.n_bytes_dev ? *p.n_bytes_dev : p.n_bytes;
    const uint64_t total_rows = n_bytes / 768;            // complete rows; the ragged tail goes to rows_kernel (launch_count)
    const uint64_t total_waves = (uint64_t)gridDim.x * 4;
    const uint64_t rpw = (total_rows + total_waves - 1) / total_waves;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    const uint64_t r0 = gw * rpw;
    const uint64_t r1 = r0 + rpw < total_rows ? r0 + rpw : total_rows;
    if (r0 >= r1) return;

    const uint32_t* const grid = p.table.grid;
    const uint32_t gwl = p.table.grid_words_log2;
    const uint32_t my_run = lane / 12u, my_win = lane % 12u;
    uint32_t run_head = 0, run_n = 0;
    auto ring = [](uint32_t pos) -> uint32_t { return pos >= XT_RUNQ ? pos - XT_RUNQ : pos; };

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

    // A drain step ISSUES the first-slot loads of up to 5 runs x 12 windows and FINISHES the batch the previous step
    // issued (compare, spill slots, atomic): the loads of a batch have a whole step to arrive (the kernel is bound by
    // memory round trips, not instructions).  finish-only when nothing is queued.
    const uint64_t key_mask = (1ULL << xt.id_shift) - 1;
    uint64_t p_line = 0, p_want = 0;
    unsigned long long p_e = XT_EMPTY;
    bool p_act = false;
    auto drain = [&]() {
        // ---- issue
        const uint32_t take = run_n < 5u ? run_n : 5u;
        const bool have = my_run < take;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (have) q = runs[ring(run_head + my_run)];
        run_head = ring(run_head + take);
        run_n -= take;
        const bool act = have && ((q.z >> (12 + my_win)) & 1u);
        const uint32_t sh = 2 * (11 - my_win);
        const uint32_t lo = __builtin_amdgcn_alignbit(q.y, q.x, sh);
        const uint32_t hi = __builtin_amdgcn_alignbit(q.z, q.y, sh) & MASK_HI;
        uint64_t line, want;
        xt_key(xt, (uint64_t)hi << 32 | lo, my_win, line, want);
        unsigned long long e = XT_EMPTY;
        if (act) e = xt.lines[(line << 4) + ((uint32_t)want & 15u)];
        // ---- finish the previous batch
        if (p_act && p_e != XT_EMPTY) {
            uint32_t id;
            if (((p_e ^ p_want) & key_mask) == 0) id = (uint32_t)(p_e >> xt.id_shift);
            else {          // slot j' holds another k-mer: spill slots, the lines behind, the overflow table
                const unsigned long long r = xt_lookup_rest(xt, p_line, p_want, id);
                if (r != XT_EMPTY) id = (uint32_t)(r >> xt.id_shift);
            }
            if (id != 0xFFFFFFFFu) __hip_atomic_fetch_add(xt.counts + id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        p_act = act;
        p_e = e;
        p_want = want;
        p_line = line;
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
        // 48-base window (see count27_kernel::scan_probe): q 36..47 own chunk, 0..35 the three chunks before
        const uint32_t W0 = (be1 << 24) | be, W1 = (be2 << 16) | (be1 >> 8), W2 = (be3 << 8) | (be2 >> 16);
        const uint32_t B = (i3 >> 9) | (i2 << 3) | (i1 << 15);
        const uint32_t a = (inv << 1) & 0xFFFu;
        const uint32_t bad_b = B ? (0xFFFFFFFFu >> __builtin_clz(B)) : 0u;
        uint32_t vm = ~(a | (0u - a) | bad_b) & 0xFFFu;
        const bool ok16 = B < 2048u;
        const uint32_t mer = __builtin_amdgcn_alignbit(W1, W0, 24);
        uint64_t gx;
        uint32_t gm, rot;
        bool as_is;
        vg_grid_probe(mer, gwl, gx, gm, rot, as_is);
        const uint2 g = reinterpret_cast<const uint2*>(grid)[gx];     // issued here ...
        while (run_n >= 5u) drain();                                    // ... and in flight while the queued runs are looked up
        const uint32_t rr = __builtin_amdgcn_alignbit(g.y, g.y, rot);
        vm &= as_is ? rr : (__builtin_bitreverse32(rr) >> 20);
        const bool cand = ok16 && (g.x & gm) == gm && vm != 0;
        const uint64_t ball = __ballot(cand);
        const uint32_t n = (uint32_t)__builtin_popcountll(ball);
        while (run_n + n > XT_RUNQ) drain();
        if (cand) {
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
            runs[ring(run_head + run_n + pos)] =
                make_uint4(__builtin_amdgcn_alignbit(W1, W0, 2), __builtin_amdgcn_alignbit(W2, W1, 2), ((W2 >> 2) & 0xFFFu) | (vm << 12), 0u);
        }
        run_n += n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
    }
    while (run_n) drain();
    drain();     // finishes the last batch (issues nothing)
}

// K5 part 1 + K6 over dense counters: cov[i] = min(255, counts[id(i)]); hist[c] += 1 for flagged keys with c != 0
__global__ void xcov_kernel(const uint32_t* counts, const uint32_t* id_of_key, uint64_t n, const uint8_t* flag, uint8_t* cov,
                            unsigned long long* hist)
{
    __shared__ unsigned int s_hist[256];
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t c32 = counts[id_of_key ? id_of_key[i] : i];
        const uint32_t c = c32 < 255u ? c32 : 255u;
        cov[i] = (uint8_t)c;
        if (hist && c != 0 && flag && flag[i]) atomicAdd(&s_hist[c], 1u);
    }
    __syncthreads();
    if (hist)
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}

// raw counters <-> key-ordered array (read-sharded all-reduce)
__global__ void xcounts_xfer_kernel(uint32_t* counts, const uint32_t* id_of_key, uint32_t* ext, uint64_t n, bool import)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t* cell = counts + (id_of_key ? id_of_key[i] : i);
        if (import) *cell = ext[i];
        else ext[i] = *cell;
    }
}

// counters far above the 255 clamp are pulled back (the read-out clamps anyway): no counter can wrap, however deep a sample
__global__ void xclamp_kernel(uint32_t* counts, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (counts[i] > 0x40000000u && counts[i] < 0x80000000u) counts[i] = 0x40000000u;      // (above: an even k's debit whose increment is still to come on another stream)
}

hipError_t launch_xclamp(const XTableView& t, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(xclamp_kernel, dim3((uint32_t)(g < 2048 ? g : 2048)), dim3(256), 0, st, t.counts, n);
    return hipGetLastError();
}

hipError_t launch_xtable_build(const XTableView& t, const unsigned long long* slots8, const uint32_t* key_slot, const uint32_t* id_of_key,
                               uint64_t n_keys, uint32_t* over_list, uint32_t over_cap, unsigned long long* over_n, hipStream_t st)
{
    hipLaunchKernelGGL(xtable_clear_kernel, dim3(4096), dim3(256), 0, st, t);
    if (n_keys) {
        const uint64_t n = n_keys * 12;
        hipLaunchKernelGGL(xtable_insert_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, t, slots8, key_slot, id_of_key, n_keys,
                           over_list, over_cap, over_n);
    }
    return hipGetLastError();
}

hipError_t launch_xtable_over(ulonglong2* over, uint32_t over_mask, const unsigned long long* slots8, const uint32_t* key_slot,
                              const uint32_t* id_of_key, const uint32_t* over_list, uint64_t n_over, hipStream_t st)
{
    hipLaunchKernelGGL(xtable_over_clear_kernel, dim3((over_mask + 256) / 256), dim3(256), 0, st, over, over_mask);
    if (n_over)
        hipLaunchKernelGGL(xtable_over_kernel, dim3((uint32_t)((n_over + 255) / 256)), dim3(256), 0, st, over, over_mask, slots8, key_slot, id_of_key,
                           over_list, n_over);
    return hipGetLastError();
}

hipError_t launch_count27x(const RowParams& p, const XTableView& t, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL(count27x_kernel, dim3(grid), dim3(256), 0, st, p, t);
    return hipGetLastError();
}

hipError_t launch_xcov(const XTableView& t, const uint32_t* id_of_key, uint64_t n, const uint8_t* flag, uint8_t* cov, unsigned long long* hist,
                       hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(xcov_kernel, dim3((uint32_t)(g < 2048 ? g : 2048)), dim3(256), 0, st, t.counts, id_of_key, n, flag, cov, hist);
    return hipGetLastError();
}

hipError_t launch_xcounts_xfer(const XTableView& t, const uint32_t* id_of_key, uint32_t* ext, uint64_t n, bool import, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(xcounts_xfer_kernel, dim3((uint32_t)(g < 2048 ? g : 2048)), dim3(256), 0, st, t.counts, id_of_key, ext, n, import);
    return hipGetLastError();
}

}  // namespace vgk
