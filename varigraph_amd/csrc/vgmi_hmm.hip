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

#include "vg_x80.h"
#include "vgmi_kernels.h"

namespace vgk {

__global__ __launch_bounds__(128) void hmm_recursion_kernel(HmmParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t hmm_smem[];
    const uint32_t n = P.n_gt, stride = P.ploidy + 1, g = threadIdx.x;
    const bool active = g < n;
    uint8_t* const s_keep = hmm_smem;                                              // n * n
    uint64_t* const s_step_m = reinterpret_cast<uint64_t*>(hmm_smem + ((n * n + 15u) & ~15u));   // 128 * stride
    int32_t* const s_step_e = reinterpret_cast<int32_t*>(s_step_m + 128u * stride);
    uint64_t* const s_r_m = reinterpret_cast<uint64_t*>(s_step_e + 128u * stride);      // 512 * stride bytes on: 8-byte aligned
    int32_t* const s_r_e = reinterpret_cast<int32_t*>(s_r_m + 128);

    const HmmChain ch = P.chains[blockIdx.x];
    const uint8_t* keep_g = P.keep + (size_t)ch.keep_index * n * n;
    for (uint32_t i = g; i < n * n; i += blockDim.x) s_keep[i] = keep_g[i];
    __syncthreads();
    const uint8_t* const my_keep = s_keep + (size_t)g * n;
    const VgX80 uniform = x80_load(P.uniform);

    // values stay normalised (vg_x80.h: VgN80) from the load of a score to the store of a row: the hundred products and sums
    // of a node then take the one-rounding path, and only a result that is not a normal number takes the general one
    VgN80 prev = {0, 0};
    for (uint64_t s = ch.first_step; s < ch.first_step + ch.n_steps; ++s) {
        const bool restart = P.restart[s] != 0;
        VgN80 o = {0, 0};
        if (active) o = n80_from(x80_load(P.obs + ((size_t)P.row[s] * n + g) * 16));
        if (!restart) {
            // (prev * no_recomb^keep) * recomb^change for this lane's previous entry and every keep
            const uint8_t* pw = P.pow + s * (size_t)(2 * stride) * 16;
            for (uint32_t k = 0; k < stride; ++k) {
                const VgN80 pk = n80_from(x80_load(pw + (size_t)k * 16)),
                            pc = n80_from(x80_load(pw + (size_t)(stride + (P.ploidy - k)) * 16));
                const VgN80 st = n80_mul(n80_mul(prev, pk), pc);
                s_step_m[g * stride + k] = st.m;
                s_step_e[g * stride + k] = st.e;
            }
        }
        __syncthreads();
        VgN80 r = {0, 0};
        if (active) {
            if (restart) {
                r = o;
            } else {
                // the table entry of term p + 1 and the keep byte of term p + 2 are fetched while term p is computed: the
                // chain r -> r is the only dependency the loop has to wait for (one wavefront per SIMD: nothing else hides LDS)
                uint32_t at = my_keep[0];
                VgN80 nx;
                nx.m = s_step_m[at];
                nx.e = s_step_e[at];
                uint32_t k2 = n > 1 ? my_keep[1] : 0;
                for (uint32_t p = 0; p < n; ++p) {
                    const VgN80 st = nx;
                    if (p + 1 < n) {
                        at = (p + 1) * stride + k2;
                        nx.m = s_step_m[at];
                        nx.e = s_step_e[at];
                        k2 = my_keep[p + 2 < n ? p + 2 : p + 1];
                    }
                    r = n80_muladd(r, st, o);
                }
            }
        }
        s_r_m[g] = r.m;
        s_r_e[g] = r.e;
        __syncthreads();
        VgN80 total = {0, 0};
        VgN80 tn;
        tn.m = s_r_m[0];
        tn.e = s_r_e[0];
        for (uint32_t p = 0; p < n; ++p) {
            const VgN80 t = tn;
            if (p + 1 < n) {
                tn.m = s_r_m[p + 1];
                tn.e = s_r_e[p + 1];
            }
            total = n80_add(total, t);
        }
        VgX80 out = uniform;
        if (total.m != 0) {
            prev = n80_div(r, total);
            out = n80_to(prev);
        } else {
            prev = n80_from(uniform);
        }
        if (active) x80_store(P.out + (s * n + g) * 16, out);
        __syncthreads();
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
    const uint64_t rowi = blockIdx.x;
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

hipError_t launch_hmm_posterior(const HmmPostParams& P, uint64_t n_rows, hipStream_t st)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(hmm_posterior_kernel, dim3((uint32_t)n_rows), dim3(128), 0, st, P);
    return hipGetLastError();
}

size_t hmm_lds_bytes(uint32_t n_gt, uint32_t ploidy)
{
    const uint32_t stride = ploidy + 1;
    return (((size_t)n_gt * n_gt + 15u) & ~(size_t)15u) + (size_t)128 * stride * 12 + (size_t)128 * 12 + 64;
}

hipError_t launch_hmm_recursion(const HmmParams& P, uint32_t n_chains, hipStream_t st)
{
    if (n_chains == 0) return hipSuccess;
    hipLaunchKernelGGL(hmm_recursion_kernel, dim3(n_chains), dim3(128), hmm_lds_bytes(P.n_gt, P.ploidy), st, P);
    return hipGetLastError();
}

}  // namespace vgk
