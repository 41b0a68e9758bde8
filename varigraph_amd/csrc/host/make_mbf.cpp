#include "make_mbf.hpp"

#include <cstring>
#include <limits>
#include <random>
#include <stdexcept>
#include <unordered_map>

#include "fastx_reader.hpp"
#include "vgmi.h"

namespace vgh {

// BloomFilter::_init_seeds (src/counting_bloom_filter.cpp:80-87) with the entropy made explicit
std::vector<uint64_t> reference_bloom_seeds(uint32_t random_device_value, uint32_t n_hash)
{
    std::mt19937 gen(random_device_value);
    std::uniform_int_distribution<size_t> dis(1, std::numeric_limits<size_t>::max());
    std::vector<uint64_t> seeds;
    for (uint32_t i = 0; i < n_hash; ++i) seeds.push_back(dis(gen));
    return seeds;
}

MbfResult make_mbf(vgmi_ctx* ctx, const std::string& fasta_path, uint32_t k, std::vector<uint64_t> seeds,
                   uint32_t rd_value)
{
    MbfResult r;
    std::unordered_map<std::string, std::string> chr;  // mFastaSeqMap
    {
        FastxReader rd(fasta_path);
        while (rd.next() >= 0) {
            r.genome_size += rd.seq().size();           // mGenomeSize += ks->seq.l (:111)
            const std::string& s = rd.seq();
            chr.emplace(rd.name(), std::string(s.data(), strnlen(s.data(), s.size())));  // `sequence = ks->seq.s` (:107)
        }
    }
    if (r.genome_size < k) throw std::runtime_error("reference shorter than k");
    const uint64_t n = r.genome_size - k + 1;           // make_mbf :154
    vgmi_bloom_params(n, 0.01, &r.m, &r.n_hash);
    if (seeds.empty()) seeds = reference_bloom_seeds(rd_value, r.n_hash);
    if (seeds.size() != r.n_hash) throw std::runtime_error("wrong number of Bloom seeds");
    if (vgmi_bloom_create(ctx, r.m, r.n_hash, seeds.data()) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
    for (const auto& kv : chr) {
        if (kv.second.empty()) throw std::runtime_error("empty chromosome sequence (the reference aborts on assert(len > 0), kmer.cpp:27)");
        if (vgmi_bloom_add_seq(ctx, kv.second.data(), kv.second.size(), k) != VGMI_OK)
            throw std::runtime_error(vgmi_last_error(ctx));
        ++r.n_chromosomes;
    }
    return r;
}

}  // namespace vgh
