// genotyper.hpp -- host HMM of `genotype`: from per-k-mer coverage to the output VCF text.
//
// Restates (own data layout, same arithmetic in the same order -- the numbers are x87 `long double` products and
// the VCF must come out byte-identical):
//   GENOTYPE::genotype / for_bac_post_run          src/genotype.cpp:41-160, 188-480   windows, forward / backward / posterior
//   haplotype_selection + HaplotypeSelect           src/genotype.cpp:500-610, src/haplotype_select.cpp
//   hidden_states / increment_vector                src/genotype.cpp:640-920
//   construct_index::find_node_up_down_seq          src/construct_index.cpp:1266-1549   (flanks of a haplotype's allele)
//   kmerBit::kmer_sketch_genotype                   src/kmer.cpp:150-190
//   transition_probabilities / observable_states / poisson / geometric / find_most_likely_depth
//                                                   src/genotype.cpp:930-1150
//   forward / backward / posterior / get_UK         src/genotype.cpp:1170-1540
//   cal_phred_scaled / save                         src/genotype.cpp:1559-1696, src/save.cpp:11-30
// Per-sample input is the coverage byte of every graph k-mer (vgmi_counts_finish); everything else comes from the
// graph index.  Like the reference, the forward pass prunes a node's k-mer list to the k-mers carried by a selected
// haplotype and the pruned list persists across samples (ConstructIndex::reset does not restore it).
#pragma once
#include "vgmi.h"
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "graph_index.hpp"

namespace vgh {

struct GenotypeConfig {           // defaults: include/varigraph.hpp:49-68
    std::string sample_type = "het";     // -g
    uint32_t sample_ploidy = 2;           // --sample-ploidy
    uint32_t haploid_num = 15;            // -n
    uint32_t chr_len_thread = 1000000;    // --granularity (bp)
    std::string transition = "rec";       // -m
    bool sv_only = false;                 // --sv
    uint32_t threads = 10;                // -t
    float min_gq = 0.0f;                  // --min-support
};

// The threads of a run share ONE budget of running compute threads (`-t`): several Genotypers work side by side (one per
// sample in flight), each with helper threads of its own, and a helper takes a token for the length of a work item -- a window, a
// block of text -- not while it waits for the device.  A Genotyper that works alone then has every thread of the run, instead of
// an n-th of them for the whole sample.  set(0) (the default): no limit.  Never hold a token while waiting for other threads.
class CpuBudget {
public:
    static void set(unsigned tokens);
    struct Hold {
        Hold();
        ~Hold();
        Hold(const Hold&) = delete;
        Hold& operator=(const Hold&) = delete;
    };
};

class Genotyper {
public:
    explicit Genotyper(const GraphIndex& g, unsigned threads = 1);   // threads: workers that copy the nodes' k-mer lists

    // cov: c of every key in graph.bin record order (g.keys); cov_node: the same counters gathered in node order (entry j = key
    // g.node_key_index[j]: vgmi_counts_finish's cov_node, the device's per-node depth lookup of src/genotype.cpp:546,660,1405-1408) --
    // when given, cov is not read at all; when nullptr, the gather is done here.  Returns the decompressed content of
    // <sample>.varigraph.vcf.gz.  Throws std::runtime_error where the reference prints and exits.
    std::string run(const uint8_t* cov, float hap_kmer_coverage, const std::string& sample_name,
                    const GenotypeConfig& cfg, const uint8_t* cov_node = nullptr);

    double last_hmm_seconds = 0, last_text_seconds = 0;   // of the last run(): windows on the pool / VCF text
    double last_device_seconds = 0;                       // ... of which the device recursion (0: the host ran it)
    size_t last_windows = 0, last_device_windows = 0;     // windows of the last run() / those whose recursion and posterior ran on the device

    // The forward / backward recursion of eligible windows (transition "rec", every genotype with `ploidy` haplotypes, at most
    // 2048 genotypes) runs on this context's device (vgmi_hmm_recursion: the reference's arithmetic bit for bit); nullptr: host.
    // max_parts: into how many device calls side by side a sample's windows may go (the process's four hardware queues,
    // shared by the consumers that genotype at the same time)
    void set_device(vgmi_ctx* ctx, unsigned max_parts = 4) { dev_ = ctx; dev_parts_ = max_parts ? max_parts : 1; }

    // `text` into `path` as block gzip (same content as SAVE's gzwrite), deflated by `threads` workers
    static void write_gz(const std::string& path, const std::string& text, unsigned threads = 1);

private:
    struct HmmScore {
        long double a = 0, b = 0;
        const std::vector<uint16_t>* haps = nullptr;   // the window's genotype this entry scores (alive during window())
    };
    struct SiteCall {
        long double probability = 0;
        std::vector<uint16_t> haps;
        std::vector<uint64_t> kmer_num;
        std::vector<float> kmer_ave_cov;
        uint8_t unique_kmers = 0;
    };
    // a node's k-mer list: a stretch of the Genotyper's one pool of places (half a million small vectors cost a fifth of a second
    // to allocate, per Genotyper); it only ever shrinks (the forward pass prunes it)
    struct KmerList {
        uint32_t* p = nullptr;
        uint32_t n = 0;
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        const uint32_t* data() const { return p; }
        const uint32_t* begin() const { return p; }
        const uint32_t* end() const { return p + n; }
        uint32_t operator[](size_t i) const { return p[i]; }
        uint32_t front() const { return p[0]; }
        uint32_t back() const { return p[n - 1]; }
        void keep(const std::vector<uint32_t>& subset)      // subset.size() <= size()
        {
            std::copy(subset.begin(), subset.end(), p);
            n = (uint32_t)subset.size();
        }
    };
    struct Node {
        uint32_t start = 0;
        const GraphNode* gn = nullptr;
        KmerList kmers;                // places in the node-ordered arrays (GraphIndex::node_key_index); pruned by the forward pass, persists across samples
        std::vector<HmmScore> hmm;     // per sample
        SiteCall call;                 // per sample
    };
    struct Chrom {
        std::string name;
        uint32_t len = 0;
        std::vector<Node> nodes;       // every node (ref-only spacers included), start ascending
    };
    // hidden states of one node, k-mer major: k-mer j of the node (after pruning) has coverage c[j], multiplicity
    // f[j] and, under genotype g of the window, h[j * n_genotypes + g] copies
    struct NodeStates {
        std::vector<uint8_t> c, f, h;
        size_t n_genotypes = 0;
        std::vector<uint32_t> kept;   // scratch reused from node to node
        std::vector<uint8_t> one;
        // Genotypes whose haplotypes carry the same k-mers of this node have the same column of h, hence the same
        // emission score: cls[g] = class of genotype g, rep[c] = its first member (both empty: every genotype for itself)
        std::vector<uint16_t> cls, rep;
    };
    // the window's genotypes as one flat list (the same for every node of the window)
    struct GenotypeList {
        std::vector<uint16_t> flat;
        std::vector<uint32_t> off;
        bool pairs = true;            // every genotype has two haplotypes
        // pairs over at most 16 haplotypes: position (in `used`) of each genotype's two haplotypes, for a byte shuffle
        std::vector<uint8_t> pos_a, pos_b;
    };
    struct Run;  // per-call constants

    struct WindowWork;   // what a window hands to the device recursion and gets back (genotyper.cpp)
    struct ScoreCtx;     // per-thread scratch and memoised libm values of score_states
    std::vector<long double> score_states(const NodeStates& ns, ScoreCtx& sc) const;
    // forced_top: the window's selected haplotypes as an earlier pass over the same window drew them (the host takes a window
    // back from the device path: same haplotypes, same pruned k-mer lists, nothing is drawn again)
    void window(Chrom& chr, uint32_t first, uint32_t last, const Run& r, WindowWork* work = nullptr,
                const std::vector<uint16_t>* forced_top = nullptr);
    void window_finish(WindowWork& w, const long double* prob, const uint32_t* winner, const Run& r, const uint32_t* tally = nullptr,
                       const uint8_t* uniq = nullptr);
    // false: more than 255 distinct genotype strings at this node (the device's ids are bytes)
    bool genotype_strings(const Node& n, const std::vector<std::vector<uint16_t>>& genotypes, uint8_t* gid, uint8_t* order) const;
    NodeStates hidden_states(Chrom& chr, uint32_t node_i, const std::vector<uint16_t>& top,
                             const std::vector<std::vector<uint16_t>>& genotypes, const std::vector<uint16_t>& used,
                             const GenotypeList& gl, double lower, double upper, bool filter, const Run& r, NodeStates&& recycled,
                             const Node* ahead);
    // what the haplotypes' sequences say about a node's under-covered multi-copy k-mers (src/genotype.cpp:760-800), as the entries
    // of the node's list that lose haplotypes (bits over `used`): for the device's second emission launch (vgmi_hmm_part_fix_rows)
    void sequence_fixes(const Chrom& chr, uint32_t node_i, const std::vector<uint16_t>& used, uint16_t gt0_mask, double lower, double upper,
                        const Run& r, std::vector<uint32_t>& fix_j, std::vector<uint16_t>& fix_mask) const;
    std::pair<std::string, std::string> flanks(const Chrom& chr, uint32_t node_i, uint16_t hap, uint16_t alt_gt,
                                               std::string& alt_seq, uint32_t want) const;
    void posterior(Node& n, const std::vector<uint16_t>& top, const Run& r) const;
    void prefetch_keys(const Node& n, const Run& r) const;

    const GraphIndex& g_;
    std::vector<Chrom> chroms_;   // mGraphMap order
    // What a part of the windows hands to the device and what does not depend on the sample: which nodes the HMM works on, their
    // entry ranges and reference-allele masks, and the genotype strings of every node that ever had a score.  Listed for the first
    // sample, kept for the next (one list per part; `key` names the options it was made under).
    struct EmitPartPlan {      // made for one pattern of scored rows; immutable once made; the device block goes with the last holder
        std::vector<uint8_t> scored;
        std::vector<std::vector<uint32_t>> win_nodes, win_rows;      // per window: the scored nodes and their rows
        // the part of a site's VCF line that every sample shares (CHROM .. INFO with FILTER forced PASS, FORMAT and the tab behind it:
        // src/genotype.cpp:1628-1640), per row of the part, end to end; a row without a site in the VCF has none
        std::string line_head;
        std::vector<uint64_t> line_head_off;                         // n_rows + 1
        vgmi_hmm_plan* plan = nullptr;
        size_t n_steps = 0;
        ~EmitPartPlan();
    };
    struct EmitPartCache {     // shared by the Genotypers of a graph (GraphIndex::shared_slots); `mu` guards the listing and the plan's making
        std::mutex mu;
        std::string key;
        std::vector<uint64_t> e_begin;
        std::vector<uint32_t> e_count, row_node;
        std::vector<uint16_t> gt0;
        std::vector<size_t> win_row0;
        std::shared_ptr<EmitPartPlan> plan;
    };
    std::vector<std::shared_ptr<EmitPartCache>> emit_cache_;
    std::unique_ptr<uint32_t[]> kmer_pool_;   // what the nodes' KmerLists point into: place j of the node-ordered arrays at [j]
    vgmi_ctx* dev_ = nullptr;
    unsigned dev_parts_ = 4;
    uint32_t n_hap_ = 0;
    std::vector<uint16_t> hap_ids_;   // the keys of g_.hap_names in their order
    std::atomic<bool> lists_whole_{true};   // no node's k-mer list has been pruned (the device's emission path needs whole, contiguous lists)
    bool entries_uploaded_ = false;         // the graph's per-entry multiplicity and haplotype bits are on dev_
    bool emit_device_off_ = false;          // the device's emission path turned out not to apply to this graph
    std::vector<uint64_t> packed_;    // per node-list entry: coverage (this sample) | multiplicity << 8 | haplotype bits << 16
};

}  // namespace vgh
