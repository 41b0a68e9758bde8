// vgmi_hmm.hip -- the forward / backward recursion of the genotyping HMM on the device, in the reference's arithmetic.
//
// replaces (inner loops of): GenotypeNameSpace::forward / backward (src/genotype.cpp:1170-1380): for every genotype g of the
// window,  r_g = sum over the previous node's entries p, IN THEIR ORDER, of  ((prev_p * no_recomb^keep) * recomb^change) * obs_g,
// keep = haplotypes g and p share, change = ploidy - keep;  then  total = sum of r_g in genotype order,  out_g = r_g / total
// (1 / n when total is zero).  The first node of a chain has  r_g = obs_g.  All of it is `long double` on the host; here it is
// vg_x80.h: the x87 unit's results bit for bit (one rounding per operation, gradual underflow), in integer instructions.
//
// One workgroup per chain (a window in one direction), one lane per genotype (<= 128).  The chain is serial node by node and
// term by term -- that is the reference's order of additions -- so the parallelism is genotypes x chains: 120 lanes x 2 x
// the windows of a sample.  The libm values (exp, pow) stay on the host and arrive as tables per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vg_x80.h"
#include "vgmi_kernels.h"

namespace vgk {

// what a step reads from memory, fetched one step ahead of its use (a chain is latency from end to end: a load that is
// waited for is time nothing else fills)
template <uint32_t STRIDE>
struct HmmStepIn {
    VgX80 keep_pow[STRIDE], change_pow[STRIDE];     // no_recomb^k, recomb^(ploidy - k)
    VgX80 obs;
    uint32_t restart;
};

// WAVES wavefronts share the genotypes evenly (2: a throughput launch; 4: a launch of few chains that has a CU per workgroup
// anyway -- the fewer genotypes a wavefront holds, the more often all of them agree that a term is negligible, and the idle
// SIMDs cost nothing: 392 instead of 405 ms on a chr20-scale sample; 8, two per SIMD, 455 ms)
template <uint32_t STRIDE, uint32_t WAVES>
__global__ __launch_bounds__(64 * WAVES) void hmm_recursion_kernel(HmmParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hmm_smem[];
    constexpr uint32_t stride = STRIDE;
    const uint32_t n = P.n_gt, per_wave = (n + WAVES - 1) / WAVES, lane = threadIdx.x & 63u;
    const bool active = lane < per_wave && (threadIdx.x >> 6) * per_wave + lane < n;
    const uint32_t g = active ? (threadIdx.x >> 6) * per_wave + lane : 0u;      // an idle lane reads genotype 0's inputs and writes nothing
    uint8_t* const s_keep = hmm_smem;                                              // n * n
    uint64_t* const s_step_m = reinterpret_cast<uint64_t*>(hmm_smem + ((n * n + 15u) & ~15u));   // 128 * stride
    int32_t* const s_step_e = reinterpret_cast<int32_t*>(s_step_m + 128u * stride);
    uint64_t* const s_r_m = reinterpret_cast<uint64_t*>(s_step_e + 128u * stride);      // 512 * stride bytes on: 8-byte aligned
    int32_t* const s_r_e = reinterpret_cast<int32_t*>(s_r_m + 128);

    const HmmChain ch = P.chains[blockIdx.x];
    const uint8_t* keep_g = P.keep + (size_t)ch.keep_index * n * n;
    for (uint32_t i = threadIdx.x; i < n * n; i += blockDim.x) s_keep[i] = keep_g[i];
    __syncthreads();
    const uint8_t* const my_keep = s_keep + (size_t)g * n;
    const VgX80 uniform = x80_load(P.uniform);
    if (ch.n_steps == 0) return;

    // a zero the compiler cannot see through: the per-step loads go through the vector memory path, whose counter the
    // waits on LDS do not share (a scalar load in flight would make every LDS wait a wait for memory)
    uint32_t lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    auto fetch = [&](uint64_t s, uint32_t row_s, HmmStepIn<STRIDE>& in) {
        const uint8_t* pw = P.pow + s * (size_t)(2 * stride) * 16 + lane_zero;
        for (uint32_t k = 0; k < stride; ++k) {
            in.keep_pow[k] = x80_load(pw + (size_t)k * 16);
            in.change_pow[k] = x80_load(pw + (size_t)(stride + (stride - 1 - k)) * 16);
        }
        in.obs = x80_load(P.obs + ((size_t)row_s * n + g) * 16);
        in.restart = P.restart[s + lane_zero];
    };
    const uint64_t s_end = ch.first_step + ch.n_steps;
    HmmStepIn<STRIDE> cur;
    fetch(ch.first_step, P.row[ch.first_step], cur);
    uint32_t row_next = ch.n_steps > 1 ? P.row[ch.first_step + 1] : 0u;

    // values stay normalised (vg_x80.h: VgN80) from the load of a score to the store of a row: the hundred products and sums
    // of a node then take the one-rounding path, and only a result that is not a normal number takes the general one
    VgN80 prev = {0, 0};
    for (uint64_t s = ch.first_step; s < s_end; ++s) {
        HmmStepIn<STRIDE> nx = cur;
        uint32_t row_after = 0;
        if (s + 1 < s_end) fetch(s + 1, row_next, nx);
        if (s + 2 < s_end) row_after = P.row[s + 2 + lane_zero];

        const bool restart = __builtin_amdgcn_readfirstlane(cur.restart) != 0;
        VgN80 o = {0, 0};
        if (active) o = n80_from(cur.obs);
        if (!restart) {
            // (prev * no_recomb^keep) * recomb^change for this lane's previous entry and every keep
            for (uint32_t k = 0; k < stride; ++k) {
                const VgN80 st = n80_mul(n80_mul(prev, n80_from(cur.keep_pow[k])), n80_from(cur.change_pow[k]));
                if (active) {
                    s_step_m[g * stride + k] = st.m;
                    s_step_e[g * stride + k] = st.e;
                }
            }
        }
        __syncthreads();
        VgN80 r = {0, 0};
        if (active) {
            if (restart || (VG_DBG(P.dbg) & 2u)) {
                r = o;
            } else {
                // the table entry of term p + 1 and the keep byte of term p + 2 are fetched while term p is computed: the
                // chain r -> r is the only dependency the loop has to wait for (one wavefront per SIMD: nothing else hides LDS)
                uint32_t at = my_keep[0];
                VgN80 nxt;
                nxt.m = s_step_m[at];
                nxt.e = s_step_e[at];
                uint32_t k2 = n > 1 ? my_keep[1] : 0;
                for (uint32_t p = 0; p < n; ++p) {
                    const VgN80 st = nxt;
                    if (p + 1 < n) {
                        at = (p + 1) * stride + k2;
                        nxt.m = s_step_m[at];
                        nxt.e = s_step_e[at];
                        k2 = my_keep[p + 2 < n ? p + 2 : p + 1];
                    }
                    // a term more than 64 binades below the sum so far leaves it as it is (its exponent is at most the
                    // operands' sum + 2); when that holds for every genotype of the wavefront -- past the entries that carry
                    // the previous node's weight it mostly does -- the term is not computed
                    const bool nothing = st.m == 0 || o.m == 0 || (r.m != 0 && r.e - (st.e + o.e - VG_X80_BIAS + 2) > 64);
                    if (__builtin_amdgcn_ballot_w64(!nothing) == 0 && !(VG_DBG(P.dbg) & 8u)) continue;
                    r = n80_muladd(r, st, o);
                }
            }
        }
        if (active) {
            s_r_m[g] = r.m;
            s_r_e[g] = r.e;
        }
        __syncthreads();
        VgN80 total = {0, 0};
        VgN80 tn;
        // every lane adds the same values in the same order; the addresses carry lane_zero so that this runs on the vector
        // unit, branch-free (as scalar code it is a jump per case and twice the time)
        tn.m = s_r_m[lane_zero];
        tn.e = s_r_e[lane_zero];
        for (uint32_t p = 0; p < ((VG_DBG(P.dbg) & 1u) ? 1u : n); ++p) {
            const VgN80 t = tn;
            if (p + 1 < n) {
                tn.m = s_r_m[p + 1 + lane_zero];
                tn.e = s_r_e[p + 1 + lane_zero];
            }
            // an entry more than 64 binades below the running sum (or zero) leaves it as it is: every lane sees the same two
            // values, so the jump is taken by the whole wavefront (most entries of an informative node are that small)
            if (t.m == 0 || (total.m != 0 && total.e - t.e > 64)) continue;
            total = n80_sum(total, t);
        }
        VgX80 out = uniform;
        if (total.m != 0) {
            prev = (VG_DBG(P.dbg) & 4u) ? r : n80_div(r, total);
            out = n80_to(prev);
        } else {
            prev = n80_from(uniform);
        }
        if (active) x80_store(P.out + (s * n + g) * 16, out);
        __syncthreads();
        cur = nx;
        row_next = row_after;
    }
}

// ---- posterior of a node (src/genotype.cpp:1387-1522) from the alpha / beta rows the recursion left on the device --------
//   denominator = sum of a_g * b_g in entry order;  post_g = (a_g * b_g) / denominator;  per genotype STRING (gid, made by the
//   host: alleles as decimal strings, sorted as strings) the sum of its entries' posts in entry order;  the first maximum in
//   string order (order[]) wins, its sum is the call's probability;  the call's genotype is the first entry of that string
//   with the largest post.  A zero denominator makes every post NaN on the host, which no comparison accepts: no call.
__device__ __forceinline__ bool x80_gt(VgX80 a, VgX80 b) { return a.e > b.e || (a.e == b.e && a.m > b.m); }

__global__ __launch_bounds__(128) void hmm_posterior_kernel(HmmPostParams P)
{
    __shared__ uint64_t s_m[128];
    __shared__ uint32_t s_e[128];
    __shared__ uint64_t s_sum_m[128];
    __shared__ uint32_t s_sum_e[128];
    __shared__ uint8_t s_gid[128];
    const uint32_t n = P.n_gt, g = threadIdx.x;
    const uint64_t rowi = P.row0 + blockIdx.x;
    const bool active = g < n;
    VgX80 p = {0, 0};
    if (active) {
        const VgX80 a = x80_load(P.ab + (P.fwd_step[rowi] * n + g) * 16), b = x80_load(P.ab + (P.bwd_step[rowi] * n + g) * 16);
        p = x80_mul(a, b);
        s_gid[g] = P.gid[rowi * n + g];
    }
    s_m[g] = p.m;
    s_e[g] = p.e;
    __syncthreads();
    VgX80 den = {0, 0};
    for (uint32_t q = 0; q < n; ++q) {
        VgX80 t;
        t.m = s_m[q];
        t.e = s_e[q];
        den = x80_add(den, t);
    }
    if (den.m == 0) {
        if (g == 0) P.winner[rowi] = 0xFFFFFFFFu;
        return;
    }
    __syncthreads();
    const VgX80 post = x80_div(p, den);
    s_m[g] = post.m;
    s_e[g] = post.e;
    __syncthreads();
    // lane k sums string k's entries in entry order
    VgX80 sum = {0, 0};
    for (uint32_t q = 0; q < n; ++q)
        if (s_gid[q] == g) {
            VgX80 t;
            t.m = s_m[q];
            t.e = s_e[q];
            sum = x80_add(sum, t);
        }
    s_sum_m[g] = sum.m;
    s_sum_e[g] = sum.e;
    __syncthreads();
    if (g == 0) {
        const uint8_t* ord = P.order + rowi * n;
        VgX80 best = {0, 0};
        uint32_t best_id = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < n && ord[k] != 0xFF; ++k) {
            VgX80 sk;
            sk.m = s_sum_m[ord[k]];
            sk.e = s_sum_e[ord[k]];
            if (best_id == 0xFFFFFFFFu || x80_gt(sk, best)) {     // the host starts from -1: the first string always enters
                best = sk;
                best_id = ord[k];
            }
        }
        VgX80 max_post = {0, 0};
        uint32_t win = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < n; ++q) {
            if (s_gid[q] != best_id) continue;
            VgX80 t;
            t.m = s_m[q];
            t.e = s_e[q];
            if (x80_gt(t, max_post)) {
                max_post = t;
                win = q;
            }
        }
        x80_store(P.prob + rowi * 16, best);
        P.winner[rowi] = best_id == 0xFFFFFFFFu ? 0xFFFFFFFEu : win;   // 0xFFFFFFFE: no string at all
    }
}

// ---- more than 128 genotypes per window (`-n` > 15 on a diploid sample: n (n + 1) / 2 pairs; any list the C ABI is given) -------
// The same recursion, the same order of additions: 256 lanes per chain, lane t holds genotypes t, t + 256, ... (GPL of them, in
// registers), the step table of ALL previous entries lives in LDS (12 bytes x n x (ploidy + 1)), and the keep matrix -- n x n
// bytes, too large for LDS from 363 genotypes on -- is read from global memory ROW p for term p: keep is symmetric (what two
// genotypes share does not depend on who asks), so keep[p][g] = keep[g][p] and the 256 lanes read consecutive bytes; the
// matrix of a window is L2-resident (n = 2 048: 4 MiB).  Time per node is n x GPL dependent terms per lane: the reference's
// O(n^2) per node, 256 genotypes at a time.
template <uint32_t STRIDE, uint32_t GPL>
__global__ __launch_bounds__(256) void hmm_recursion_big_kernel(HmmParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hmm_smem[];
    constexpr uint32_t stride = STRIDE, T = 256;
    const uint32_t n = P.n_gt, tid = threadIdx.x;
    uint64_t* const s_step_m = reinterpret_cast<uint64_t*>(hmm_smem);                 // n * stride
    uint64_t* const s_r_m = s_step_m + (size_t)n * stride;                             // n
    int32_t* const s_step_e = reinterpret_cast<int32_t*>(s_r_m + n);                   // n * stride
    int32_t* const s_r_e = s_step_e + (size_t)n * stride;                              // n
    const HmmChain ch = P.chains[blockIdx.x];
    const uint8_t* const keep_g = P.keep + (size_t)ch.keep_index * n * n;
    const VgX80 uniform = x80_load(P.uniform);
    if (ch.n_steps == 0) return;
    uint32_t lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    bool act[GPL];
    uint32_t gi[GPL];
#pragma unroll
    for (uint32_t j = 0; j < GPL; ++j) {
        act[j] = tid + j * T < n;
        gi[j] = act[j] ? tid + j * T : 0u;     // an idle slot reads genotype 0's inputs and writes nothing
    }
    VgN80 prev[GPL];
#pragma unroll
    for (uint32_t j = 0; j < GPL; ++j) prev[j] = VgN80{0, 0};
    const uint64_t s_end = ch.first_step + ch.n_steps;
    for (uint64_t s = ch.first_step; s < s_end; ++s) {
        const uint32_t row_s = P.row[s];
        const bool restart = P.restart[s] != 0;
        const uint8_t* pw = P.pow + s * (size_t)(2 * stride) * 16;
        VgN80 o[GPL], r[GPL];
#pragma unroll
        for (uint32_t j = 0; j < GPL; ++j) {
            o[j] = VgN80{0, 0};
            if (act[j]) o[j] = n80_from(x80_load(P.obs + ((size_t)row_s * n + gi[j]) * 16));
            r[j] = VgN80{0, 0};
        }
        if (!restart) {
            for (uint32_t k = 0; k < stride; ++k) {
                const VgN80 kp = n80_from(x80_load(pw + (size_t)k * 16 + lane_zero));
                const VgN80 cp = n80_from(x80_load(pw + (size_t)(stride + (stride - 1 - k)) * 16 + lane_zero));
#pragma unroll
                for (uint32_t j = 0; j < GPL; ++j) {
                    const VgN80 st = n80_mul(n80_mul(prev[j], kp), cp);
                    if (act[j]) {
                        s_step_m[(size_t)gi[j] * stride + k] = st.m;
                        s_step_e[(size_t)gi[j] * stride + k] = st.e;
                    }
                }
            }
        }
        __syncthreads();
        if (restart) {
#pragma unroll
            for (uint32_t j = 0; j < GPL; ++j) r[j] = o[j];
        } else {
            uint8_t kb[GPL], kb_next[GPL];
#pragma unroll
            for (uint32_t j = 0; j < GPL; ++j) kb_next[j] = keep_g[gi[j]];
            for (uint32_t p = 0; p < n; ++p) {
#pragma unroll
                for (uint32_t j = 0; j < GPL; ++j) kb[j] = kb_next[j];
                if (p + 1 < n) {
#pragma unroll
                    for (uint32_t j = 0; j < GPL; ++j) kb_next[j] = keep_g[(size_t)(p + 1) * n + gi[j]];
                }
#pragma unroll
                for (uint32_t j = 0; j < GPL; ++j) {
                    VgN80 st;
                    st.m = s_step_m[(size_t)p * stride + kb[j]];
                    st.e = s_step_e[(size_t)p * stride + kb[j]];
                    // a term more than 64 binades below the sum so far leaves it as it is (vgmi_hmm.hip, the small kernel)
                    const bool nothing = !act[j] || st.m == 0 || o[j].m == 0 || (r[j].m != 0 && r[j].e - (st.e + o[j].e - VG_X80_BIAS + 2) > 64);
                    if (__builtin_amdgcn_ballot_w64(!nothing) == 0) continue;
                    r[j] = n80_muladd(r[j], st, o[j]);
                }
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < GPL; ++j)
            if (act[j]) {
                s_r_m[gi[j]] = r[j].m;
                s_r_e[gi[j]] = r[j].e;
            }
        __syncthreads();
        VgN80 total = {0, 0};
        for (uint32_t p = 0; p < n; ++p) {
            VgN80 t;
            t.m = s_r_m[p + lane_zero];
            t.e = s_r_e[p + lane_zero];
            if (t.m == 0 || (total.m != 0 && total.e - t.e > 64)) continue;
            total = n80_sum(total, t);
        }
#pragma unroll
        for (uint32_t j = 0; j < GPL; ++j) {
            VgX80 out = uniform;
            if (total.m != 0) {
                prev[j] = n80_div(r[j], total);
                out = n80_to(prev[j]);
            } else {
                prev[j] = n80_from(uniform);
            }
            if (act[j]) x80_store(P.out + (s * n + gi[j]) * 16, out);
        }
        __syncthreads();
    }
}

// the posterior for any number of genotypes: 256 lanes, lane t takes entries t, t + 256, ...; at most 255 distinct genotype
// strings per node (gid is a byte; the host keeps a node with more on its own path)
__global__ __launch_bounds__(256) void hmm_posterior_big_kernel(HmmPostParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t post_smem[];
    const uint32_t n = P.n_gt, tid = threadIdx.x;
    uint64_t* const s_m = reinterpret_cast<uint64_t*>(post_smem);       // n
    uint64_t* const s_sum_m = s_m + n;                                  // 256
    uint32_t* const s_e = reinterpret_cast<uint32_t*>(s_sum_m + 256);   // n
    uint32_t* const s_sum_e = s_e + n;                                  // 256
    uint8_t* const s_gid = reinterpret_cast<uint8_t*>(s_sum_e + 256);   // n
    const uint64_t rowi = P.row0 + blockIdx.x;
    const uint64_t fs = P.fwd_step[rowi], bs = P.bwd_step[rowi];
    for (uint32_t g = tid; g < n; g += 256) {
        const VgX80 a = x80_load(P.ab + (fs * n + g) * 16), b = x80_load(P.ab + (bs * n + g) * 16);
        const VgX80 p = x80_mul(a, b);
        s_m[g] = p.m;
        s_e[g] = p.e;
        s_gid[g] = P.gid[rowi * n + g];
    }
    __syncthreads();
    VgX80 den = {0, 0};
    for (uint32_t q = 0; q < n; ++q) {
        VgX80 t;
        t.m = s_m[q];
        t.e = s_e[q];
        den = x80_add(den, t);
    }
    if (den.m == 0) {
        if (tid == 0) P.winner[rowi] = 0xFFFFFFFFu;
        return;
    }
    __syncthreads();
    for (uint32_t g = tid; g < n; g += 256) {
        VgX80 p;
        p.m = s_m[g];
        p.e = s_e[g];
        const VgX80 post = x80_div(p, den);
        s_m[g] = post.m;
        s_e[g] = post.e;
    }
    __syncthreads();
    VgX80 sum = {0, 0};      // lane k sums string k's entries in entry order
    for (uint32_t q = 0; q < n; ++q)
        if (s_gid[q] == tid) {
            VgX80 t;
            t.m = s_m[q];
            t.e = s_e[q];
            sum = x80_add(sum, t);
        }
    s_sum_m[tid] = sum.m;
    s_sum_e[tid] = sum.e;
    __syncthreads();
    if (tid == 0) {
        const uint8_t* ord = P.order + rowi * n;
        VgX80 best = {0, 0};
        uint32_t best_id = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < n && ord[k] != 0xFF; ++k) {
            VgX80 sk;
            sk.m = s_sum_m[ord[k]];
            sk.e = s_sum_e[ord[k]];
            if (best_id == 0xFFFFFFFFu || x80_gt(sk, best)) {
                best = sk;
                best_id = ord[k];
            }
        }
        VgX80 max_post = {0, 0};
        uint32_t win = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < n; ++q) {
            if (s_gid[q] != best_id) continue;
            VgX80 t;
            t.m = s_m[q];
            t.e = s_e[q];
            if (x80_gt(t, max_post)) {
                max_post = t;
                win = q;
            }
        }
        x80_store(P.prob + rowi * 16, best);
        P.winner[rowi] = best_id == 0xFFFFFFFFu ? 0xFFFFFFFEu : win;
    }
}

// ---- emission scores of a node: hidden states (src/genotype.cpp:640-830) and observable states (:960-1000) --------------------
// One workgroup of 128 lanes per node, lane g = genotype g = a PAIR of haplotypes (used[pos_a[g]], used[pos_b[g]]) -- or three or four
// of them (a polyploid sample's genotypes are blocks of `ploidy` consecutive haplotypes, src/genotype.cpp:846-873; round 5).  Per k-mer of
// the node, in list order: coverage c, multiplicity f and the haplotype bits decide, the same for every lane, which of the
// <= 16 haplotypes count as carrying the k-mer; the lane's copy number h is the sum over its two; (h, c, f) go through
// most_likely_depth (:1118-1145, float and double arithmetic as the host's SSE code does it) and select the term -- geometric for
// h = 0, Poisson(ave * h) otherwise, libm values tabulated by the host per sample -- and the lane multiplies it onto its product
// in the reference's long double (vg_x80.h).  A k-mer that is under-covered, multi-copy and carried makes the reference check
// the haplotype's SEQUENCE (:760-800): such a node is flagged and scored by the host.  Every k-mer list must be whole (no
// selected-haplotype pruning: all haplotypes are selected), which the caller guarantees and bit 1 of the flags verifies.
__device__ __forceinline__ uint32_t hmm_most_likely_depth(uint32_t h, uint32_t c, uint32_t f, float ave, double upper)
{
    if (f == 1u) return c;
    if (h > 0u && (float)c > ave * (float)h) return (uint32_t)(int32_t)(ave * (float)h) & 0xFFu;
    if (h == 0u && (float)c > ave) return ((double)f > (double)((float)c) / upper) ? 0u : (uint32_t)(int32_t)((float)c / (float)f) & 0xFFu;
    if (h == 0u) return (uint32_t)(int32_t)((float)c / (float)f) & 0xFFu;
    return c;
}

__global__ __launch_bounds__(128) void hmm_emissions_kernel(HmmEmitParams P)
{
    __shared__ uint64_t s_tm[1280];      // (ploidy + 1) x 256 terms: up to four haplotypes per genotype
    __shared__ int32_t s_te[1280];
    const uint32_t g = threadIdx.x;
    for (uint32_t i = g; i < (P.ploidy + 1u) * 256u; i += 128u) {
        const VgN80 t = n80_from(x80_load(P.tables + (size_t)i * 16));
        s_tm[i] = t.m;
        s_te[i] = t.e;
    }
    __syncthreads();
    const uint64_t rowi = P.fix_rows ? P.fix_rows[blockIdx.x] : P.row_lo + blockIdx.x;
    uint32_t fp = P.fix_rows ? P.fix_off[blockIdx.x] : 0u;
    const uint32_t fe = P.fix_rows ? P.fix_off[blockIdx.x + 1] : 0u;
    const uint64_t e0 = P.entry_begin[rowi];
    const uint32_t cnt = P.entry_count[rowi], gt0 = P.gt0[rowi];
    const bool active = g < P.n_gt;
    const uint32_t pa = P.pos_a[active ? g : 0u], pb = P.pos_b[active ? g : 0u];
    const uint32_t pc = P.ploidy > 2u ? P.pos_more[0][active ? g : 0u] : 0u, pd = P.ploidy > 3u ? P.pos_more[1][active ? g : 0u] : 0u;
    VgN80 prod;
    prod.m = 1ULL << 63;      // 1.0L
    prod.e = VG_X80_BIAS;
    uint32_t kept = 0, flag = 0;
    for (uint32_t j = 0; j < cnt; ++j) {
        const unsigned long long w = P.packed[e0 + j];
        const uint32_t c = P.cov[e0 + j], f = (uint32_t)(w >> 8) & 0xFFu;
        const unsigned long long bits = w >> 16;
        const uint32_t lb = (uint32_t)(bits >> (P.bl8 - 1u)) & 1u;
        if ((bits & P.top_mask) == 0) {      // the host would drop it from the list: this path does not prune
            flag |= 2u;
            continue;
        }
        ++kept;
        const bool in_interval = lb == 1u && (double)c >= P.lower && (double)c <= P.upper;
        uint32_t om = 0;
        for (uint32_t p = 0; p < P.n_used; ++p) {
            const uint32_t one = (in_interval && ((gt0 >> p) & 1u)) ? 1u : (uint32_t)(bits >> P.used[p]) & 1u;
            om |= one << p;
        }
        if ((double)c < P.lower && f >= 2u && om != 0) flag |= 1u;
        if (fp < fe && P.fix_j[fp] == j) {      // (the second launch: haplotypes whose sequence does not hold this k-mer do not carry it)
            om &= ~(uint32_t)P.fix_mask[fp];
            ++fp;
        }
        const uint32_t fj = (lb == 1u && f == 1u) ? 2u : f;
        uint32_t h = ((om >> pa) & 1u) + ((om >> pb) & 1u);
        if (P.ploidy > 2u) h += (om >> pc) & 1u;
        if (P.ploidy > 3u) h += (om >> pd) & 1u;
        const uint32_t cc = hmm_most_likely_depth(h, c, fj, P.ave, P.upper);
        const uint32_t ti = h * 256u + cc;
        VgN80 t;
        t.m = s_tm[ti];
        t.e = s_te[ti];
        prod = n80_mul(prod, t);
    }
    if (active) x80_store(P.obs + (rowi * P.n_gt + g) * 16, n80_to(prod));
    if (g == 0 && !P.fix_rows) {
        P.n_kept[rowi] = kept;
        P.flags[rowi] = (uint8_t)flag;
    }
}

// rows the host scored itself, handed in as one block: row[i] of the part <- src[i]
__global__ __launch_bounds__(128) void hmm_scatter_rows_kernel(uint8_t* obs, const uint64_t* rows, const uint8_t* src, uint32_t n_gt)
{
    const uint32_t g = threadIdx.x;
    if (g < n_gt)
        *reinterpret_cast<uint4*>(obs + (rows[blockIdx.x] * n_gt + g) * 16) = *reinterpret_cast<const uint4*>(src + ((size_t)blockIdx.x * n_gt + g) * 16);
}

// ---- a call's k-mer tallies (src/genotype.cpp:1387-1414 as posterior() reads them for the called haplotypes) -------------------
// Per row (node) with a called genotype g = (hap_a[g], hap_b[g]): over the node's entries, how many k-mers each called haplotype
// carries and the sum of their coverages (the caller divides), and how many k-mers have multiplicity <= 1 (clamped at 255).  A
// haplotype outside the panel or the selection reads as (0, 0), as on the host.  One lane per row: the entries of a node are ~50
// consecutive words.
__global__ __launch_bounds__(256) void hmm_tally_kernel(const unsigned long long* __restrict__ packed, const uint8_t* __restrict__ cov,
                                                        const uint64_t* __restrict__ entry_begin, const uint32_t* __restrict__ entry_count,
                                                        const uint32_t* __restrict__ winner, const uint8_t* __restrict__ hap_ab, uint32_t n_gt, uint32_t n_hap,
                                                        unsigned long long sel_mask, uint64_t n_rows, uint32_t* __restrict__ out, uint8_t* __restrict__ uniq)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= n_rows) return;
    uint32_t num_a = 0, sum_a = 0, num_b = 0, sum_b = 0, u = 0;
    const uint32_t g = winner[r];
    if (g < n_gt) {
        const uint32_t ha = hap_ab[2u * g], hb = hap_ab[2u * g + 1u];
        const bool ok_a = ha < n_hap && ((sel_mask >> ha) & 1ull), ok_b = hb < n_hap && ((sel_mask >> hb) & 1ull);
        const uint64_t e0 = entry_begin[r];
        const uint32_t cnt = entry_count[r];
        for (uint32_t j = 0; j < cnt; ++j) {
            const unsigned long long w = packed[e0 + j];
            const uint32_t c = cov[e0 + j];
            const unsigned long long bits = w >> 16;
            if (((uint32_t)(w >> 8) & 0xFFu) <= 1u && u < 255u) ++u;
            if (ok_a && ((bits >> ha) & 1ull)) { ++num_a; sum_a += c; }
            if (ok_b && ((bits >> hb) & 1ull)) { ++num_b; sum_b += c; }
        }
    }
    out[4 * r] = num_a;
    out[4 * r + 1] = sum_a;
    out[4 * r + 2] = num_b;
    out[4 * r + 3] = sum_b;
    uniq[r] = (uint8_t)u;
}

hipError_t launch_hmm_tally(const unsigned long long* packed, const uint8_t* cov, const uint64_t* entry_begin, const uint32_t* entry_count, const uint32_t* winner,
                            const uint8_t* hap_ab, uint32_t n_gt, uint32_t n_hap, unsigned long long sel_mask, uint64_t n_rows, uint32_t* out, uint8_t* uniq,
                            hipStream_t st)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(hmm_tally_kernel, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, packed, cov, entry_begin, entry_count, winner, hap_ab, n_gt, n_hap,
                       sel_mask, n_rows, out, uniq);
    return hipGetLastError();
}

hipError_t launch_hmm_scatter_rows(uint8_t* obs, const uint64_t* rows, const uint8_t* src, uint32_t n_gt, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(hmm_scatter_rows_kernel, dim3((uint32_t)n), dim3(128), 0, st, obs, rows, src, n_gt);
    return hipGetLastError();
}

hipError_t launch_hmm_emissions(const HmmEmitParams& P, uint64_t n_rows, hipStream_t st)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(hmm_emissions_kernel, dim3((uint32_t)n_rows), dim3(128), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_hmm_posterior(const HmmPostParams& P, uint64_t n_rows, hipStream_t st)
{
    if (n_rows == 0) return hipSuccess;
    if (P.n_gt > 128) {
        const size_t lds = (size_t)P.n_gt * 13 + 256 * 12 + 64;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hmm_posterior_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(hmm_posterior_big_kernel, dim3((uint32_t)n_rows), dim3(256), lds, st, P);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(hmm_posterior_kernel, dim3((uint32_t)n_rows), dim3(128), 0, st, P);
    return hipGetLastError();
}

size_t hmm_lds_bytes(uint32_t n_gt, uint32_t ploidy)
{
    const uint32_t stride = ploidy + 1;
    return (((size_t)n_gt * n_gt + 15u) & ~(size_t)15u) + (size_t)128 * stride * 12 + (size_t)128 * 12 + 64;
}

namespace {
template <uint32_t STRIDE, uint32_t WAVES>
hipError_t launch_recursion_as(const HmmParams& Q, uint32_t n_chains, size_t lds, size_t plain_lds, hipStream_t st)
{
    if (lds > plain_lds &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(hmm_recursion_kernel<STRIDE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
        (void)hipGetLastError();    // not granted: the plain launch
        lds = plain_lds;
    }
    hipLaunchKernelGGL((hmm_recursion_kernel<STRIDE, WAVES>), dim3(n_chains), dim3(64 * WAVES), lds, st, Q);
    return hipGetLastError();
}
template <uint32_t STRIDE>
hipError_t launch_recursion_waves(uint32_t waves, const HmmParams& Q, uint32_t n_chains, size_t lds, size_t plain_lds, hipStream_t st)
{
    switch (waves) {
        case 4: return launch_recursion_as<STRIDE, 4>(Q, n_chains, lds, plain_lds, st);
        default: return launch_recursion_as<STRIDE, 2>(Q, n_chains, lds, plain_lds, st);
    }
}
}  // namespace

namespace {
template <uint32_t STRIDE, uint32_t GPL>
hipError_t launch_recursion_big_as(const HmmParams& Q, uint32_t n_chains, hipStream_t st)
{
    const size_t lds = (size_t)Q.n_gt * (STRIDE + 1) * 12 + 64;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hmm_recursion_big_kernel<STRIDE, GPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((hmm_recursion_big_kernel<STRIDE, GPL>), dim3(n_chains), dim3(256), lds, st, Q);
    return hipGetLastError();
}
template <uint32_t STRIDE>
hipError_t launch_recursion_big(const HmmParams& Q, uint32_t n_chains, hipStream_t st)
{
    if (Q.n_gt <= 512) return launch_recursion_big_as<STRIDE, 2>(Q, n_chains, st);
    if (Q.n_gt <= 1024) return launch_recursion_big_as<STRIDE, 4>(Q, n_chains, st);
    return launch_recursion_big_as<STRIDE, 8>(Q, n_chains, st);
}
}  // namespace

hipError_t launch_hmm_recursion(const HmmParams& P, uint32_t n_chains, hipStream_t st)
{
    if (n_chains == 0) return hipSuccess;
    if (P.n_gt > 128) {
        if (P.n_gt > VGMI_HMM_MAX_GT) return hipErrorInvalidValue;
        HmmParams Q = P;
        Q.dbg = 0;
        switch (P.ploidy) {
            case 1: return launch_recursion_big<2>(Q, n_chains, st);
            case 2: return launch_recursion_big<3>(Q, n_chains, st);
            case 3: return launch_recursion_big<4>(Q, n_chains, st);
            case 4: return launch_recursion_big<5>(Q, n_chains, st);
            default: return hipErrorInvalidValue;
        }
    }
    const size_t plain_lds = hmm_lds_bytes(P.n_gt, P.ploidy);
    // A small launch asks for more than half a CU's LDS: its workgroups then have a CU each.  The parts of a sample are
    // launches of a few dozen chains on streams of their own; the dispatcher starts each at the same CUs, and chains that
    // share a SIMD wait for each other's instructions (measured: 450 instead of 400 ms for the later parts).
    const size_t alone = 84 * 1024;
    const bool spread = n_chains <= 64 && plain_lds < alone;
    const size_t lds = spread ? alone : plain_lds;
    uint32_t waves = spread ? 4u : 2u;
    if (const char* w = getenv("VGMI_HMM_WAVES")) waves = atoi(w) == 4 ? 4u : 2u;     // A/B
    // (Several callers on one device -- the samples of a run side by side -- need nothing special: a SIMD runs two of these
    // wavefronts at little more than one's pace, a chain is latency; tools/gpu_hmm_pack.sh, 1 000 steps of 120 genotypes: 60 chains in
    // one launch 33.5 ms, 480 chains 39.6 ms, 960 chains 39.5 ms in the dense layout -- and 78.7 ms packed two workgroups to a CU.)
    HmmParams Q = P;
    Q.dbg = vgmi_dbg_env();
    switch (P.ploidy) {
        case 1: return launch_recursion_waves<2>(waves, Q, n_chains, lds, plain_lds, st);
        case 2: return launch_recursion_waves<3>(waves, Q, n_chains, lds, plain_lds, st);
        case 3: return launch_recursion_waves<4>(waves, Q, n_chains, lds, plain_lds, st);
        case 4: return launch_recursion_waves<5>(waves, Q, n_chains, lds, plain_lds, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vgk
