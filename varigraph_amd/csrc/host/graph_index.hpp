// graph_index.hpp -- the genome-graph index as flat arrays (host side of the MI355X build).
//
// Restates, for the needs of `genotype --load-graph`:
//   ConstructIndex::load_index   src/construct_index.cpp:911-1105  (byte layout: SURVEY.md App. A)
//   ConstructIndex::graph2node   src/construct_index.cpp:710-751, graph2node_run :1572-1603
//   the sample-independent half of Varigraph::get_hom_kmer  src/varigraph.cpp:263-287
// The reference keeps the k-mer table as unordered_map<uint64_t,kmerCovFreBitVec> with a heap
// vector per k-mer and reads BitVec one byte per read() call; here everything is bulk-read into
// contiguous arrays that can be handed to vgmi_table_upload / vgmi_nodes_upload as they are.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

struct vgmi_ctx;

namespace vgh {

struct GraphNode {  // one nodeSrt (include/construct_index.hpp:105-121) without per-sample state
    uint32_t start = 0;
    std::vector<std::string> seqs;      // seqVec
    std::vector<uint16_t> hap_gt;       // hapGtVec
    std::vector<uint64_t> kmer_hash;    // kmerHashVec (graph.bin order) -- of a graph built in memory (`construct`)
    // ... of a graph being loaded: the hashes are only ever read by graph2node, inside load(), so they stay where they are in the file
    // buffer (unaligned, 8 bytes each) instead of a million small vectors; cleared when load() returns
    const uint8_t* kmer_file = nullptr;
    uint32_t kmer_file_n = 0;
    size_t n_kmers() const { return kmer_file ? kmer_file_n : kmer_hash.size(); }
    uint64_t kmer(size_t j) const
    {
        if (!kmer_file) return kmer_hash[j];
        uint64_t v;
        __builtin_memcpy(&v, kmer_file + 8 * j, 8);
        return v;
    }
};

struct GraphIndex {
    uint64_t graph_base_num = 0;
    uint32_t k = 0, vcf_ploidy = 0;
    std::string vcf_head;
    std::map<std::string, uint32_t> chr_len;                                       // mFastaLenMap
    std::map<std::string, std::map<uint32_t, std::vector<std::string>>> vcf_info;  // mVcfInfoMap: per site its first kVcfFieldsKept columns
    static constexpr uint32_t kVcfFieldsKept = 8;                                  // CHROM POS ID REF ALT QUAL FILTER INFO
    uint64_t genome_size = 0;
    uint16_t hap_num = 0;
    std::map<uint16_t, std::string> hap_names;                                     // mHapMap
    std::map<std::string, std::map<uint32_t, GraphNode>> graph;                    // mGraphMap
    // the nodes of every chromosome in map order (what a walk over mGraphMap[chr] visits), as an array: graph2node and every
    // Genotyper read the million nodes through it instead of chasing the tree again.  load() fills it as it reads the file
    std::map<std::string, std::vector<const GraphNode*>> graph_seq;
    void index_nodes();                                                            // rebuilds graph_seq from graph
    // per entry of the node lists: multiplicity << 8 | haplotype bits << 16 (the graph's half of the word a Genotyper keeps per entry
    // and sample; bitlen <= 6).  A gather over the key arrays: made once here when several Genotypers share the graph, else by the
    // Genotyper itself
    std::vector<uint64_t> entry_words;
    void build_entry_words();

    // k-mer table, file record order
    std::vector<uint64_t> keys;
    std::vector<uint8_t> f;
    std::vector<int8_t> bitvec;  // keys.size() * bitlen
    uint64_t bitlen = 0;
    std::vector<uint8_t> hom_flag;

    // variant nodes in mGraphMap order (chromosome lexicographic, start ascending) and their
    // k-mer lists after graph2node, as CSR over key indices
    std::vector<std::string> chr_names;
    std::vector<uint32_t> node_chr, node_start;
    std::vector<uint64_t> node_off;
    std::vector<uint32_t> node_key_index;

    unsigned threads = 1;   // host threads of load(): key index build and node resolution (graph2node)

    // what the users of one graph share beyond the graph itself, by name (the Genotypers of a run: per part of the windows the row
    // lists and the device-resident recursion inputs, genotyper.cpp); lives and dies with the graph
    mutable std::mutex shared_mu;
    mutable std::map<std::string, std::shared_ptr<void>> shared_slots;

    // throws std::runtime_error
    void load(const std::string& path);
    void graph2node();
    void compute_hom_flags();
    // called by load() from its own thread as soon as keys / k are complete (the node lists and flags are not yet): lets the
    // caller start the device's table build while graph2node still runs on the host
    std::function<void()> on_keys;
    // graph2node's lookups as ONE batch (the device's table: vgmi_table_lookup): index_out[i] = index of keys[i] in `keys`, 0xFFFFFFFF when
    // absent; returns false when it cannot serve (then, and when unset, the host's own index does the lookups)
    std::function<bool(const uint64_t* keys, size_t n, uint32_t* index_out)> batched_find;
    // FastqKmer::save_index / load_index (src/fastq_kmer.cpp:200-298): ReadBase + the k-mer records with the sample's coverage
    void save_reads_index(const std::string& path, const uint8_t* cov, uint64_t read_base) const;
    void load_reads_index(const std::string& path, uint8_t* cov, uint64_t& read_base) const;
    int upload(vgmi_ctx* ctx) const;
    int upload_nodes(vgmi_ctx* ctx) const;
};

}  // namespace vgh
