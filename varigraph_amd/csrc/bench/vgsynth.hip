// vgsynth.hip -- libvgsynth.so: the seeded synthetic workloads of bench.py, tools/ and the tests (vgsynth.h).  BENCH / TEST TOOLING: not part
// of the product library, which neither links nor loads it.
#include "vgsynth.h"

#include <hip/hip_runtime.h>

#include "../vgmi_device.h"
#include "vg_synth.h"

namespace {

#define VG_SYNTH_MAX_HAPS 8
struct SynthHaps {
    uint32_t n;
    uint64_t off[VG_SYNTH_MAX_HAPS];
    uint64_t len[VG_SYNTH_MAX_HAPS];
};


static uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap)
{
    uint64_t g = (n + block - 1) / block;
    if (g == 0) g = 1;
    return (uint32_t)(g < cap ? g : cap);
}

// bench/test tooling: seeded synthetic read block (vg_synth.h)
__global__ void synth_reads_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                   const char* hap_cat, SynthHaps haps, char* out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t total = n_reads * (read_len + 1);
    const char* hp[VG_SYNTH_MAX_HAPS];
    for (uint32_t h = 0; h < haps.n; ++h) hp[h] = hap_cat + haps.off[h];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t r = i / (read_len + 1);
        const uint32_t j = (uint32_t)(i - r * (read_len + 1));
        out[i] = j == read_len ? '\n' : vgs_read_base(seed, first_read + r, j, read_len, hp, haps.len, haps.n);
    }
}

hipError_t launch_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, const char* hap_cat,
                              const SynthHaps& haps, char* out, hipStream_t st)
{
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_reads_kernel, dim3(grid_for(n_reads * (read_len + 1), 256, 8192)), dim3(256), 0, st, seed,
                       first_read, n_reads, read_len, hap_cat, haps, out);
    return hipGetLastError();
}


}  // namespace

extern "C" {

int vgs_reads_device(int device, void* stream, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                     const char* dev_hap_cat, const uint64_t* hap_off, uint32_t n_hap, char* dev_out)
{
    if (!dev_hap_cat || !hap_off || !dev_out) return -1;
    if (n_hap < 1 || n_hap > VG_SYNTH_MAX_HAPS) return -1;      // 1 .. 8 haplotypes
    if (read_len < 1 || read_len > VGS_INSERT) return -1;      // read_len in 1 .. 350
    SynthHaps h{};
    h.n = n_hap;
    for (uint32_t i = 0; i < n_hap; ++i) {
        h.off[i] = hap_off[i];
        h.len[i] = hap_off[i + 1] - hap_off[i];
        if (h.len[i] < VGS_INSERT) return -1;      // a haplotype shorter than the insert size
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipSetDevice(device) != hipSuccess) return -2;
    if (launch_synth_reads(seed, first_read, n_reads, read_len, dev_hap_cat, h, dev_out, st) != hipSuccess) return -2;
    if (hipStreamSynchronize(st) != hipSuccess) return -2;
    return 0;
}

int vgs_reads_host(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, const char* hap_cat,
                   const uint64_t* hap_off, uint32_t n_hap, char* out)
{
    if (!hap_cat || !hap_off || !out || n_hap < 1 || n_hap > VG_SYNTH_MAX_HAPS) return -1;
    if (read_len < 1 || read_len > VGS_INSERT) return -1;
    const char* hp[VG_SYNTH_MAX_HAPS];
    uint64_t hl[VG_SYNTH_MAX_HAPS];
    for (uint32_t i = 0; i < n_hap; ++i) {
        hp[i] = hap_cat + hap_off[i];
        hl[i] = hap_off[i + 1] - hap_off[i];
        if (hl[i] < VGS_INSERT) return -1;
    }
    for (uint64_t r = 0; r < n_reads; ++r) {
        char* o = out + r * (read_len + 1);
        for (uint32_t j = 0; j < read_len; ++j) o[j] = vgs_read_base(seed, first_read + r, j, read_len, hp, hl, n_hap);
        o[read_len] = '\n';
    }
    return 0;
}

int vgs_snp_keys_host(const char* ref, uint64_t ref_len, const uint64_t* pos, const char* alts, uint64_t n_sites,
                      uint32_t k, uint64_t* keys_out)
{
    if (!ref || !pos || !alts || !keys_out || k < 1 || k > 28) return -1;
    const uint64_t mask = k == 32 ? ~0ULL : (1ULL << (2 * k)) - 1;
    for (uint64_t i = 0; i < n_sites; ++i) {
        const uint64_t p = pos[i];
        if (p < k - 1 || p + k > ref_len) return -1;
        for (uint32_t allele = 0; allele < 2; ++allele) {
            uint64_t fwd = 0, rc = 0;
            uint64_t* out = keys_out + (2 * i + allele) * k;
            for (uint64_t q = p - (k - 1); q <= p + (k - 1); ++q) {
                const uint32_t c = vg_nt4((unsigned char)(allele && q == p ? alts[i] : ref[q]));
                if (c > 3) return -1;
                fwd = (fwd << 2 | c) & mask;
                rc = (rc >> 2) | (uint64_t)(3u ^ c) << (2 * (k - 1));
                if (q >= p) out[q - p] = vg_hash64(fwd < rc ? fwd : rc, mask) << 8 | k;
            }
        }
    }
    return 0;
}

int vgs_reference_host(uint64_t seed, uint64_t len, char* out)
{
    if (!out) return -1;
    for (uint64_t i = 0; i < len; ++i) out[i] = vgs_ref_base(seed, i);
    return 0;
}

}  // extern "C"
