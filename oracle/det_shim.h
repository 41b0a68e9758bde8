/* det_shim.h -- force-included (-include) when building the *deterministic* flavour of the
 * reference (oracle/_ref/varigraph_det).  TEST INFRASTRUCTURE ONLY.
 *
 * The reference seeds two mt19937 engines from std::random_device
 * (src/counting_bloom_filter.cpp:80-87 Bloom seeds, include/haplotype_select.hpp:25
 * Dirichlet sampler), so `construct` writes a different graph.bin every run.  Replacing
 * the entropy source by a constant makes both bit-reproducible without touching any
 * reference source line.  The stock flavour (oracle/_ref/varigraph_ref) is built without it.
 */
#ifndef VG_DET_SHIM_H
#define VG_DET_SHIM_H
#include <random>
namespace std {
struct vg_fixed_rd {
    typedef unsigned int result_type;
    vg_fixed_rd() {}
    explicit vg_fixed_rd(const std::string&) {}
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return 0xffffffffu; }
    double entropy() const noexcept { return 0.0; }
    result_type operator()() { return 20241022u; }
};
}  // namespace std
#define random_device vg_fixed_rd
#endif
