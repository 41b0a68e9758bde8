// constructor.hpp -- `construct`: reference FASTA + cohort VCF -> graph.bin, with the counting Bloom filter of the
// reference genome built and queried on the device.
//
// Restates (own flow, same container types where their iteration order reaches the file):
//   ConstructIndex::build_fasta_index   src/construct_index.cpp:85-139
//   ConstructIndex::make_mbf            src/construct_index.cpp:150-177      -> vgmi_bloom_create / vgmi_bloom_add_seq (K3)
//   ConstructIndex::construct           src/construct_index.cpp:188-480      VCF -> nodes, VCF lines kept for the output
//   ConstructIndex::vcf_construct       src/construct_index.cpp:507-590
//   ConstructIndex::index + index_run   src/construct_index.cpp:592-700, 1125-1248
//     kmerBit::kmer_sketch_construct    src/kmer.cpp:65-97                  BloomFilter::count / ::find -> vgmi_bloom_query (K4)
//   construct_index::find_node_up_down_seq                                  -> node_flanks.hpp
//   ConstructIndex::save_index          src/construct_index.cpp:760-902      byte layout: SURVEY.md Appendix A
// graph.bin's k-mer records come out in the iteration order of std::unordered_map<uint64_t, ...> and every node's
// k-mer list in the order of a per-node std::unordered_map: the same libstdc++ containers are filled in the same
// sequence here, which makes the file byte-identical to the reference's for the same Bloom seeds.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct vgmi_ctx;

namespace vgh {

struct ConstructConfig {                  // defaults: include/varigraph.hpp:49-68
    std::string reference, vcf, out = "graph.bin";
    uint32_t k = 27;                      // -k
    uint32_t vcf_ploidy = 2;              // --vcf-ploidy
    bool fast = false;                    // --fast
    bool use_unique_kmers = false;        // --use-unique-kmers
    uint32_t threads = 10;                // -t: host threads of the indexing phases
    std::vector<uint64_t> bloom_seeds;    // empty: drawn like BloomFilter::_init_seeds from random_device_value
    uint32_t random_device_value = 0;
    bool release_memory = true;           // false: the graph containers are left to process exit (the CLI: freeing 1e8 small
                                          // allocations costs seconds and nothing follows)
};

struct ConstructStats {
    uint64_t genome_size = 0, graph_base_num = 0, n_kmers = 0, n_haplotypes = 0, n_variant_nodes = 0;
    uint64_t bloom_bytes = 0, bloom_queries = 0;
    double seconds_bloom = 0, seconds_index = 0;
};

// throws std::runtime_error where the reference prints and exits
ConstructStats construct_graph(vgmi_ctx* ctx, const ConstructConfig& cfg);

}  // namespace vgh
