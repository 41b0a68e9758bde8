// make_mbf.hpp -- construct-side driver of the counting Bloom filter on the device.
//
// Restates ConstructIndex::build_fasta_index + make_mbf (src/construct_index.cpp:85-139,150-177):
// every FASTA record adds its length to the genome size, the FIRST record of a name provides that
// chromosome's sequence (unordered_map::emplace), n = genomeSize - k + 1, p = 0.01, and every
// chromosome goes through kmerBit::kmer_sketch_bf -> BloomFilter::add -- here vgmi_bloom_add_seq.
// The seeds are the caller's: the reference draws them from std::random_device
// (src/counting_bloom_filter.cpp:80-87); reference_bloom_seeds() reproduces that draw for a given
// random_device value.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct vgmi_ctx;

namespace vgh {

std::vector<uint64_t> reference_bloom_seeds(uint32_t random_device_value, uint32_t n_hash);

struct MbfResult {
    uint64_t genome_size = 0, m = 0;
    uint32_t n_hash = 0, n_chromosomes = 0;
};

// throws std::runtime_error; seeds.size() must equal the derived hash count (or be empty: then
// reference_bloom_seeds(rd_value, n_hash) is used)
MbfResult make_mbf(vgmi_ctx* ctx, const std::string& fasta_path, uint32_t k, std::vector<uint64_t> seeds,
                   uint32_t rd_value);

}  // namespace vgh
