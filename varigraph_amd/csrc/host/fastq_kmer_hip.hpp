// fastq_kmer_hip.hpp -- the MI355X twin of the reference's read-counting class.
//
// Reference seam:
//   FastqKmer(unordered_map<uint64_t,kmerCovFreBitVec>& table, const vector<string>& fastqs,
//             const uint32_t& k, const uint32_t& threads)          include/fastq_kmer.hpp:57-62
//   void build_fastq_index();  uint64_t mReadBase;                  include/fastq_kmer.hpp:42,73
//   FastqKmerKernel(..., int buffer); build_fastq_index_kernel()    include/fastq_kmer.cuh:15-36
// Post-condition kept: after build_fastq_index() every key's c == min(255, occurrences over all
// reads of all files) and mReadBase == sum of record lengths.  Here the table lives on the device
// behind a vgmi context (uploaded once per run from the graph index) instead of being borrowed as
// a host unordered_map; the counters are fetched with fetch() when the caller wants them.
//
// Pipeline: one parser thread per input file (up to `threads`), each filling read blocks
// ('\n'-joined sequences) that the calling thread hands to vgmi_reads_submit, which stages them
// through pinned memory on alternating HIP streams.  The reference parses on its main thread and
// runs files one after another (src/fastq_kmer.cpp:47-49); the result is order-independent.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct vgmi_ctx;

namespace vgh {

class FastqKmerHip {
public:
    uint64_t mReadBase = 0;  // Sequencing file size (same name as the reference member)
    uint64_t mReadNum = 0;

    FastqKmerHip(vgmi_ctx* ctx, const std::vector<std::string>& fastqFileNameVec, uint32_t kmerLen,
                 uint32_t threads, size_t block_bytes = 32u << 20);

    // resets the device counters, streams every file through the device; throws std::runtime_error
    // on I/O errors, empty reads (reference: assert(len > 0)) and device errors
    void build_fastq_index();

    // c per key / per node entry / masked histogram (any may be null), see vgmi_counts_finish
    void fetch(uint8_t* cov, uint8_t* cov_node, uint64_t* hist256);

    double kernel_seconds() const { return kernel_s_; }

private:
    vgmi_ctx* ctx_;
    std::vector<std::string> files_;
    uint32_t k_, threads_;
    size_t block_bytes_;
    double kernel_s_ = 0;
};

// Coverage statistics that turn the counters into hapKmerCoverage_ (src/varigraph.cpp:198,220-243,
// 308-362).  Returns false where the reference exits with "Failed to retrieve depth information".
struct CoverageStats {
    float read_depth = 0, hap_kmer_coverage = 0;
    uint8_t max_coverage = 0, hom_coverage = 0;
};
bool coverage_stats(const uint64_t hist[256], uint64_t read_base, uint64_t genome_size, uint32_t sample_ploidy,
                    bool use_depth, CoverageStats& out);

}  // namespace vgh
