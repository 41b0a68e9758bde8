// ubench_mem.hip -- random-access ceilings of the MI355X memory system, the numbers the large-graph count kernel is
// designed against: independent random 8-byte loads / 4-byte atomics over a footprint F,
// optionally next to a streaming reader.  Standalone: hipcc --offload-arch=gfx950 -O3 tools/ubench_mem.hip -o tools/bin/ubench_mem
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// MODE 0: 8-byte loads, 1: 4-byte no-return atomics, 2: 4-byte returning atomics, 3: 8-byte nt loads,
// 4: load 8 B then atomic on the SAME 64-byte sector (probe + in-slot counter), 5: 16-byte loads, 6: 4-byte loads
// GROUP: lanes of a group of GROUP consecutive lanes share one random 64*?-byte region (locality of a run)
template <int MODE, int ILP>
__global__ __launch_bounds__(256) void rnd_kernel(uint8_t* base, uint64_t mask64, uint64_t n_per_lane, uint32_t group_log2,
                                                  uint32_t span_log2, unsigned long long* sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t gid = tid >> group_log2;
    const uint64_t sub = tid & ((1u << group_log2) - 1);
    uint64_t acc = 0;
    for (uint64_t i = 0; i < n_per_lane; i += ILP) {
        uint64_t v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            const uint64_t h = mix((gid * n_per_lane + i + j) * 0x9E3779B97F4A7C15ULL + 12345);
            // group base: aligned to 1 << span_log2 bytes; the lane's element inside it
            uint64_t off = (h & mask64) & ~((1ULL << span_log2) - 1);
            const uint64_t inner = (mix(h + sub) >> 20) & ((1ULL << span_log2) - 1);
            off += group_log2 ? (inner & ~15ULL) : 0;
            uint8_t* p = base + (off & ~15ULL);
            if (MODE == 0) v[j] = *reinterpret_cast<const volatile uint64_t*>(p);
            else if (MODE == 3) v[j] = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p));
            else if (MODE == 1) { atomicAdd(reinterpret_cast<unsigned int*>(p), 1u); v[j] = 0; }
            else if (MODE == 2) v[j] = atomicAdd(reinterpret_cast<unsigned int*>(p), 1u);
            else if (MODE == 4) {
                const uint64_t x = *reinterpret_cast<const volatile uint64_t*>(p);
                if (x != 0x1234567) atomicAdd(reinterpret_cast<unsigned int*>(p + 8), 1u);
                v[j] = x;
            } else if (MODE == 5) {
                typedef uint32_t v4 __attribute__((ext_vector_type(4)));
                const v4 q = *reinterpret_cast<const volatile v4*>(p);
                v[j] = q.x ^ q.w;
            } else {
                v[j] = *reinterpret_cast<const volatile uint32_t*>(p);
            }
        }
#pragma unroll
        for (int j = 0; j < ILP; ++j) acc ^= v[j];
    }
    if (acc == 0x9999) atomicAdd(sink, 1ULL);
}

// streaming reader that keeps running until *stop != 0 (runs on another stream next to rnd_kernel)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* src, uint64_t n16, unsigned long long* sink, int passes)
{
    u32x4 a = {0, 0, 0, 0};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            const u32x4 v = __builtin_nontemporal_load(&src[i]);
            a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
        }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x9999) atomicAdd(sink, 1ULL);
}

template <int MODE>
static double run(uint8_t* buf, uint64_t fbytes, uint64_t total, uint32_t group_log2, uint32_t span_log2, uint32_t wgs_per_cu,
                  unsigned long long* sink)
{
    const uint32_t grid = 256 * wgs_per_cu, block = 256;
    const uint64_t lanes = (uint64_t)grid * block;
    uint64_t npl = total / lanes;
    npl = (npl + 3) & ~3ULL;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rnd_kernel<MODE, 4>), dim3(grid), dim3(block), 0, 0, buf, fbytes - 1, npl / 8, group_log2, span_log2, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((rnd_kernel<MODE, 4>), dim3(grid), dim3(block), 0, 0, buf, fbytes - 1, npl, group_log2, span_log2, sink);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return (double)(npl * lanes) / (ms * 1e-3);
}

int main(int argc, char** argv)
{
    const uint64_t max_f = argc > 1 ? strtoull(argv[1], 0, 10) << 20 : (32ULL << 30);
    const uint64_t total = argc > 2 ? strtoull(argv[2], 0, 10) : (1ULL << 29);
    uint8_t* buf = nullptr;
    CHK(hipMalloc(&buf, max_f + 4096));
    CHK(hipMemset(buf, 0, max_f + 4096));
    unsigned long long* sink = nullptr;
    CHK(hipMalloc(&sink, 8));
    CHK(hipMemset(sink, 0, 8));
    const char* names[] = {"load8", "atomic4_noret", "atomic4_ret", "load8_nt", "load8+atomic_same_sector", "load16", "load4"};
    printf("{\"what\": \"random accesses per second, independent lanes (group 1) unless stated\", \"rows\": [\n");
    bool first = true;
    auto emit = [&](const char* name, uint64_t f, uint32_t g, uint32_t s, uint32_t w, double rate) {
        printf("%s{\"op\": \"%s\", \"footprint_mib\": %llu, \"group\": %u, \"span_bytes\": %u, \"wgs_per_cu\": %u, \"gacc_per_s\": %.3f, \"gb_per_s_64B_sectors\": %.1f}",
               first ? "" : ",\n", name, (unsigned long long)(f >> 20), 1u << g, g ? 1u << s : 0, w, rate / 1e9, rate * 64 / 1e9);
        first = false;
        fflush(stdout);
    };
    for (uint64_t f = 32ULL << 20; f <= max_f; f <<= 2) {
        emit(names[0], f, 0, 4, 8, run<0>(buf, f, total, 0, 4, 8, sink));
        emit(names[0], f, 0, 4, 4, run<0>(buf, f, total, 0, 4, 4, sink));
        emit(names[3], f, 0, 4, 8, run<3>(buf, f, total, 0, 4, 8, sink));
        emit(names[5], f, 0, 4, 8, run<5>(buf, f, total, 0, 4, 8, sink));
        emit(names[6], f, 0, 4, 8, run<6>(buf, f, total, 0, 4, 8, sink));
        emit(names[1], f, 0, 4, 8, run<1>(buf, f, total / 2, 0, 4, 8, sink));
        emit(names[2], f, 0, 4, 8, run<2>(buf, f, total / 2, 0, 4, 8, sink));
        emit(names[4], f, 0, 4, 8, run<4>(buf, f, total / 2, 0, 4, 8, sink));
        // locality: groups of 8 lanes inside one 64 B / 128 B / 256 B / 1 KiB region
        for (uint32_t s : {6u, 7u, 8u, 10u}) {
            emit(names[0], f, 3, s, 8, run<0>(buf, f, total, 3, s, 8, sink));
            emit(names[2], f, 3, s, 8, run<2>(buf, f, total / 2, 3, s, 8, sink));
        }
    }
    // random loads next to a streaming reader (separate streams)
    {
        const uint64_t f = max_f >= (2ULL << 30) ? (2ULL << 30) : max_f;
        u32x4* sbuf = nullptr;
        const uint64_t sbytes = 4ULL << 30;
        CHK(hipMalloc(&sbuf, sbytes));
        CHK(hipMemset(sbuf, 1, sbytes));
        hipStream_t s1, s2;
        CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t a0, a1, b0, b1;
        CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1)); CHK(hipEventCreate(&b0)); CHK(hipEventCreate(&b1));
        // streaming alone
        CHK(hipEventRecord(b0, s2));
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 4), dim3(256), 0, s2, sbuf, sbytes / 16, sink, 4);
        CHK(hipEventRecord(b1, s2));
        CHK(hipDeviceSynchronize());
        float msb = 0;
        CHK(hipEventElapsedTime(&msb, b0, b1));
        printf(",\n{\"op\": \"stream_nt_alone\", \"gb_per_s\": %.1f}", 4.0 * sbytes / (msb * 1e-3) / 1e9);
        const uint32_t grid = 256 * 4;
        const uint64_t lanes = (uint64_t)grid * 256;
        const uint64_t npl = ((total / lanes) + 3) & ~3ULL;
        CHK(hipEventRecord(a0, s1));
        hipLaunchKernelGGL((rnd_kernel<0, 4>), dim3(grid), dim3(256), 0, s1, buf, f - 1, npl, 0u, 4u, sink);
        CHK(hipEventRecord(a1, s1));
        CHK(hipEventRecord(b0, s2));
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 4), dim3(256), 0, s2, sbuf, sbytes / 16, sink, 4);
        CHK(hipEventRecord(b1, s2));
        CHK(hipDeviceSynchronize());
        float msa = 0;
        CHK(hipEventElapsedTime(&msa, a0, a1));
        CHK(hipEventElapsedTime(&msb, b0, b1));
        printf(",\n{\"op\": \"load8_2GiB_next_to_stream\", \"gacc_per_s\": %.3f, \"stream_gb_per_s\": %.1f}", npl * lanes / (msa * 1e-3) / 1e9,
               4.0 * sbytes / (msb * 1e-3) / 1e9);
    }
    printf("\n]}\n");
    return 0;
}
