// vg_synth.h -- seeded synthetic-read generator shared by the host tool and the HIP kernel.
//
// BENCH / TEST TOOLING, not part of the reference seam: the reference ships no data and no
// generator, so the workloads of BASELINE.json are produced here.
// Everything is a pure function of (seed, pair index, base index) through a counter-based
// mixer, so the CPU and the GPU produce byte-identical read blocks in any order.
//
// Model: fragment start uniform on the chosen haplotype, insert 350,
// both mates `read_len` bases, strand random, substitution 0.5 %/base, N 0.01 %/base.
// A pair p yields reads 2p (mate 1) and 2p+1 (mate 2).  A read block is the '\n'-joined
// concatenation of reads: read r occupies bytes [r*(read_len+1), r*(read_len+1)+read_len)
// followed by '\n'.
#ifndef VG_SYNTH_H
#define VG_SYNTH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define VGS_FN __host__ __device__ static inline
#else
#define VGS_FN static inline
#endif

VGS_FN uint64_t vgs_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
VGS_FN uint64_t vgs_rand(uint64_t seed, uint64_t a, uint64_t b)
{
    return vgs_mix(vgs_mix(seed ^ (a * 0xD6E8FEB86659FD93ULL)) + b);
}

#define VGS_INSERT 350u
#define VGS_N_THRESH 1678u      /* 0.01 % of 2^24 */
#define VGS_SUB_THRESH 85564u   /* N_THRESH + 0.5 % of 2^24 */

// One base of read `r` (= 2*pair + mate).  hap[h] are ASCII haplotype sequences of length
// hap_len[h] (>= VGS_INSERT); n_hap >= 1.
VGS_FN char vgs_read_base(uint64_t seed, uint64_t r, uint32_t j, uint32_t read_len,
                          const char* const* hap, const uint64_t* hap_len, uint32_t n_hap)
{
    const uint64_t p = r >> 1;
    const uint32_t mate = (uint32_t)(r & 1);
    const uint32_t h = (uint32_t)(vgs_rand(seed, p, 0) % n_hap);
    const uint64_t start = vgs_rand(seed, p, 1) % (hap_len[h] - VGS_INSERT + 1);
    const uint32_t flip = (uint32_t)(vgs_rand(seed, p, 2) & 1);
    // mate^flip == 0: forward read at the fragment's left end; == 1: reverse-complement read
    // ending at the fragment's right end.
    char b;
    if ((mate ^ flip) == 0) {
        b = hap[h][start + j];
    } else {
        char c = hap[h][start + VGS_INSERT - 1 - j];
        b = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    }
    const uint64_t e = vgs_rand(seed, r, 16 + j);
    const uint32_t u = (uint32_t)(e & 0xFFFFFF);
    if (u < VGS_N_THRESH) return 'N';
    if (u < VGS_SUB_THRESH && b != 'N') {
        const char acgt[4] = {'A', 'C', 'G', 'T'};
        uint32_t code = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3;
        return acgt[(code + 1 + (uint32_t)((e >> 24) % 3)) & 3];
    }
    return b;
}

// iid uniform reference base i
VGS_FN char vgs_ref_base(uint64_t seed, uint64_t i)
{
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    return acgt[vgs_mix(seed + i) >> 62];
}
#endif
