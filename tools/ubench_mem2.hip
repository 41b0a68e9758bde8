// ubench_mem2.hip -- second round of memory-system probes for the large-graph count kernel (see ubench_mem.hip):
// where the ~54 G requests/s ceiling of random loads sits (L2-resident footprints), what adjacent lanes cost when they
// touch CONSECUTIVE words of one random line (the access pattern of a candidate run: 12 k-mers -> 12 neighbouring slots /
// counters), 32- vs 64-bit atomics, and whether loads and atomics share one ceiling.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// A group of G = 1 << glog consecutive lanes picks ONE random line-aligned region and lane j of the group touches element j
// (ESZ bytes apart, starting at the region's first byte).  OP 0: load ESZ bytes (4 / 8 / 16), 1: 32-bit atomic add (ret),
// 2: 64-bit atomic add (ret), 3: 32-bit atomic add (no return), 4: plain 4-byte store
template <int OP, int ESZ>
__global__ __launch_bounds__(256) void grp_kernel(uint8_t* base, uint64_t mask, uint64_t n_per_lane, uint32_t glog, uint32_t align_log2,
                                                  unsigned long long* sink, uint32_t active_mod, uint32_t active_lt)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = (uint32_t)(tid >> 6);
    if (active_mod && (wave % active_mod) >= active_lt) return;
    const uint64_t gid = tid >> glog, sub = tid & ((1u << glog) - 1);
    uint64_t acc = 0;
    for (uint64_t i = 0; i < n_per_lane; i += 4) {
        uint64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t h = mix((gid * n_per_lane + i + j) * 0x9E3779B97F4A7C15ULL + 777);
            uint8_t* p = base + (((h & mask) >> align_log2) << align_log2) + sub * ESZ;
            if (OP == 0) {
                if (ESZ == 4) v[j] = *reinterpret_cast<const volatile uint32_t*>(p);
                else if (ESZ == 8) v[j] = *reinterpret_cast<const volatile uint64_t*>(p);
                else {
                    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                    const v4 q = *reinterpret_cast<const volatile v4*>(p);
                    v[j] = q.x ^ q.w;
                }
            } else if (OP == 1) v[j] = atomicAdd(reinterpret_cast<unsigned int*>(p), 1u);
            else if (OP == 2) v[j] = atomicAdd(reinterpret_cast<unsigned long long*>(p), 1ULL);
            else if (OP == 3) { atomicAdd(reinterpret_cast<unsigned int*>(p), 1u); v[j] = 0; }
            else { *reinterpret_cast<volatile uint32_t*>(p) = (uint32_t)h; v[j] = 0; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
    if (acc == 0x9999) atomicAdd(sink, 1ULL);
}

static bool g_first = true;
template <int OP, int ESZ>
static void run(const char* name, uint8_t* buf, uint64_t fbytes, uint64_t total_lane_ops, uint32_t glog, uint32_t align_log2,
                unsigned long long* sink, uint32_t wgs_per_cu = 8)
{
    const uint32_t grid = 256 * wgs_per_cu, block = 256;
    const uint64_t lanes = (uint64_t)grid * block;
    uint64_t npl = ((total_lane_ops / lanes) + 3) & ~3ULL;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((grp_kernel<OP, ESZ>), dim3(grid), dim3(block), 0, 0, buf, fbytes - 1, npl / 4 ? (npl / 4 + 3) & ~3ULL : 4, glog, align_log2, sink, 0u, 0u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((grp_kernel<OP, ESZ>), dim3(grid), dim3(block), 0, 0, buf, fbytes - 1, npl, glog, align_log2, sink, 0u, 0u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_rate = (double)(npl * lanes) / (ms * 1e-3);
    printf("%s{\"op\": \"%s\", \"elem_bytes\": %d, \"footprint_mib\": %.3f, \"group_lanes\": %u, \"region_align\": %u, \"wgs_per_cu\": %u, "
           "\"glane_ops_per_s\": %.2f, \"ggroups_per_s\": %.2f}",
           g_first ? "" : ",\n", name, ESZ, fbytes / 1048576.0, 1u << glog, 1u << align_log2, wgs_per_cu, lane_rate / 1e9, lane_rate / (1u << glog) / 1e9);
    g_first = false;
    fflush(stdout);
}

int main()
{
    const uint64_t max_f = 4ULL << 30;
    uint8_t* buf = nullptr;
    CHK(hipMalloc(&buf, max_f + 4096));
    CHK(hipMemset(buf, 0, max_f + 4096));
    unsigned long long* sink = nullptr;
    CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(sink, 0, 8));
    const uint64_t T = 1ULL << 30;
    printf("{\"rows\": [\n");
    // (a) where the random-load ceiling sits: L2-resident to HBM-resident footprints, one lane per line
    for (uint64_t f : {256ULL << 10, 1ULL << 20, 2ULL << 20, 4ULL << 20, 8ULL << 20, 16ULL << 20, 64ULL << 20, 1ULL << 30, 4ULL << 30}) {
        run<0, 8>("load", buf, f, T, 0, 7, sink);
        run<0, 8>("load", buf, f, T, 0, 7, sink, 2);
    }
    // (b) a group of G adjacent lanes on consecutive elements of one random 128-byte-aligned line (4 GiB footprint)
    for (uint32_t g : {0u, 1u, 2u, 3u, 4u}) {
        run<0, 8>("load", buf, max_f, T, g, 7, sink);
        run<1, 4>("atomic32_ret", buf, max_f, T / 2, g, 7, sink);
        run<3, 4>("atomic32_noret", buf, max_f, T / 2, g, 7, sink);
        if (g <= 3) run<2, 8>("atomic64_ret", buf, max_f, T / 2, g, 7, sink);
        run<4, 4>("store32", buf, max_f, T / 2, g, 7, sink);
    }
    // 12 lanes of 16 on 8-byte slots = the run pattern (96 bytes of one line); 16 lanes on 4-byte counters = 64 bytes
    run<0, 8>("load", buf, 2ULL << 30, T, 4, 7, sink);
    run<0, 16>("load", buf, 2ULL << 30, T, 2, 7, sink);
    run<0, 16>("load", buf, 2ULL << 30, T, 3, 7, sink);
    // (c) same on a 128 MiB footprint (Infinity-Cache resident)
    for (uint32_t g : {0u, 3u}) {
        run<0, 8>("load", buf, 128ULL << 20, T, g, 7, sink);
        run<1, 4>("atomic32_ret", buf, 128ULL << 20, T / 2, g, 7, sink);
        run<2, 8>("atomic64_ret", buf, 128ULL << 20, T / 2, g, 7, sink);
    }
    // (d) loads and atomics together: even waves load, odd waves do atomics; do the two ceilings add or share?
    {
        const uint32_t grid = 256 * 8;
        const uint64_t lanes = (uint64_t)grid * 256;
        const uint64_t npl = ((T / lanes) + 3) & ~3ULL;
        hipStream_t s1, s2;
        CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t a0, a1, b0, b1;
        CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1)); CHK(hipEventCreate(&b0)); CHK(hipEventCreate(&b1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a0, s1));
        hipLaunchKernelGGL((grp_kernel<0, 8>), dim3(grid / 2), dim3(256), 0, s1, buf, (2ULL << 30) - 1, npl, 0u, 7u, sink, 0u, 0u);
        CHK(hipEventRecord(a1, s1));
        CHK(hipEventRecord(b0, s2));
        hipLaunchKernelGGL((grp_kernel<1, 4>), dim3(grid / 2), dim3(256), 0, s2, buf + (2ULL << 30), (2ULL << 30) - 1, npl / 3, 0u, 7u, sink, 0u, 0u);
        CHK(hipEventRecord(b1, s2));
        CHK(hipDeviceSynchronize());
        float msa = 0, msb = 0;
        CHK(hipEventElapsedTime(&msa, a0, a1));
        CHK(hipEventElapsedTime(&msb, b0, b1));
        printf(",\n{\"op\": \"concurrent: loads (half the grid) next to atomics (other half)\", \"load_g_per_s\": %.2f, \"atomic_g_per_s\": %.2f, \"load_ms\": %.2f, \"atomic_ms\": %.2f}",
               npl * (lanes / 2) / (msa * 1e-3) / 1e9, (npl / 3) * (lanes / 2) / (msb * 1e-3) / 1e9, msa, msb);
    }
    printf("\n]}\n");
    return 0;
}
