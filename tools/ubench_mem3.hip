// ubench_mem3.hip -- probes for the partitioned pipeline: (1) random loads into an XCD-private slice that fits the XCD's
// L2 (every workgroup picks the slice of the XCD it actually runs on, HW_REG_XCC_ID) next to a streaming reader in the
// same wave; (2) workgroup-scope (L2-executed) atomics into XCD-private slices vs device-scope atomics.
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
__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

// OP 0: 8-byte load, 1: device-scope atomic add, 2: workgroup-scope atomic add (executed in the XCD's L2)
// every wave also streams `stream_per_op` 16-byte nt loads per random op from a big buffer (0 = none)
template <int OP>
__global__ __launch_bounds__(256) void slice_kernel(uint8_t* base, uint64_t slice_bytes, uint64_t n_per_lane, const uint4* stream,
                                                    uint64_t stream_n16, uint32_t stream_per_op, unsigned long long* sink,
                                                    unsigned int* xcd_census)
{
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcd_census[x], 1u);
    uint8_t* slice = base + (uint64_t)x * slice_bytes;
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    uint64_t sp = tid;
    const uint64_t sstride = (uint64_t)gridDim.x * blockDim.x;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    for (uint64_t i = 0; i < n_per_lane; i += 4) {
        uint64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t h = mix((tid * n_per_lane + i + j) * 0x9E3779B97F4A7C15ULL + 99);
            uint8_t* p = slice + ((h % slice_bytes) & ~7ULL);
            if (OP == 0) v[j] = *reinterpret_cast<const volatile uint64_t*>(p);
            else if (OP == 1) v[j] = atomicAdd(reinterpret_cast<unsigned int*>(p), 1u);
            else v[j] = __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (uint32_t s = 0; s < stream_per_op; ++s) {
                const v4 q = __builtin_nontemporal_load(reinterpret_cast<const v4*>(stream) + (sp % stream_n16));
                acc ^= q.x;
                sp += sstride;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
    if (acc == 0x9999) atomicAdd(sink, 1ULL);
}

static bool g_first = true;
template <int OP>
static void run(const char* name, uint8_t* buf, uint64_t slice_bytes, uint64_t total, const uint4* stream, uint64_t stream_n16,
                uint32_t stream_per_op, unsigned long long* sink, unsigned int* census)
{
    const uint32_t grid = 256 * 8, block = 256;
    const uint64_t lanes = (uint64_t)grid * block;
    const uint64_t npl = ((total / lanes) + 3) & ~3ULL;
    CHK(hipMemset(census, 0, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((slice_kernel<OP>), dim3(grid), dim3(block), 0, 0, buf, slice_bytes, (uint64_t)8, stream, stream_n16, stream_per_op, sink, census);
    CHK(hipDeviceSynchronize());
    CHK(hipMemset(census, 0, 64));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((slice_kernel<OP>), dim3(grid), dim3(block), 0, 0, buf, slice_bytes, npl, stream, stream_n16, stream_per_op, sink, census);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned int h[16];
    CHK(hipMemcpy(h, census, 64, hipMemcpyDeviceToHost));
    const double rate = (double)(npl * lanes) / (ms * 1e-3);
    printf("%s{\"op\": \"%s\", \"slice_mib_per_xcd\": %.3f, \"stream_16B_loads_per_op\": %u, \"gops_per_s\": %.2f, \"stream_gb_per_s\": %.1f, "
           "\"blocks_per_xcd\": [%u,%u,%u,%u,%u,%u,%u,%u]}",
           g_first ? "" : ",\n", name, slice_bytes / 1048576.0, stream_per_op, rate / 1e9, rate * stream_per_op * 16 / 1e9,
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    g_first = false;
    fflush(stdout);
}

int main()
{
    uint8_t* buf = nullptr;
    const uint64_t max_slice = 64ULL << 20;
    CHK(hipMalloc(&buf, 8 * max_slice));
    CHK(hipMemset(buf, 0, 8 * max_slice));
    uint4* stream = nullptr;
    const uint64_t sbytes = 4ULL << 30;
    CHK(hipMalloc(&stream, sbytes));
    CHK(hipMemset(stream, 1, sbytes));
    unsigned long long* sink = nullptr;
    unsigned int* census = nullptr;
    CHK(hipMalloc(&sink, 8)); CHK(hipMemset(sink, 0, 8));
    CHK(hipMalloc(&census, 64));
    const uint64_t T = 1ULL << 31;
    printf("{\"rows\": [\n");
    for (uint64_t s : {512ULL << 10, 1ULL << 20, 2ULL << 20, 3ULL << 20, 4ULL << 20, 16ULL << 20, 64ULL << 20}) {
        run<0>("load8, XCD-private slice", buf, s, T, stream, sbytes / 16, 0, sink, census);
        run<0>("load8, XCD-private slice", buf, s, T / 2, stream, sbytes / 16, 1, sink, census);
        run<0>("load8, XCD-private slice", buf, s, T / 4, stream, sbytes / 16, 4, sink, census);
        run<1>("atomic32 device scope, XCD-private slice", buf, s, T / 4, stream, sbytes / 16, 0, sink, census);
        run<2>("atomic32 workgroup scope (L2), XCD-private slice", buf, s, T / 2, stream, sbytes / 16, 0, sink, census);
        run<2>("atomic32 workgroup scope (L2), XCD-private slice", buf, s, T / 4, stream, sbytes / 16, 1, sink, census);
    }
    printf("\n]}\n");
    return 0;
}
