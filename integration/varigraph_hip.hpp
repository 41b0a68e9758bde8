// varigraph_hip.hpp -- the adapter INTEGRATION.md describes, as real code: the reference's own
// orchestrator (class Varigraph) with its read-counting step replaced by the MI355X path.
//
// This file is NOT reference code and contains none; it is compiled TOGETHER with the unmodified
// reference sources (oracle/Makefile target _ref/varigraph_hip) exactly like the reference's own
// CUDA twin `VarigraphKernel : Varigraph` (include/varigraph.cuh:72-119, src/varigraph.cu:62-117)
// is compiled next to its CPU classes.  Everything after the counters -- coverage peak, HMM, VCF
// writer -- is the reference's code, so a VCF written by this binary shows that the device
// counters are a drop-in for `FastqKmer::build_fastq_index`.
#pragma once
#include <stdexcept>

#include "include/varigraph.hpp"   // reference (-I/root/reference)
#include "fastq_kmer_hip.hpp"      // this repo: varigraph_amd/csrc/host
#include "vgmi.h"

class VarigraphHip : public Varigraph {
    vgmi_ctx* ctx_ = nullptr;
    std::vector<uint64_t> keys_;   // iteration order of mGraphKmerHashHapStrMap, fixed after load()
    bool uploaded_ = false;

    [[noreturn]] void die(const std::string& what) {
        std::cerr << "[VarigraphHip::" << getTime() << "] " << what << std::endl;
        exit(1);  // the reference's error model: message + exit (cuda_error_handling.hpp:10-16)
    }

public:
    VarigraphHip(const VarigraphConfig& config, int gpu, size_t buffer_mib) : Varigraph(config) {
        if (vgmi_create(gpu, buffer_mib, &ctx_) != VGMI_OK) die(vgmi_last_error(nullptr));
    }
    ~VarigraphHip() { vgmi_destroy(ctx_); }

    // Varigraph::fastq_genotype (src/varigraph.cpp:153-172) with kmer_read -> kmer_read_hip
    void fastq_genotype_hip() {
        ConstructIndexClassPtr_->graph2node();
        for (const auto& [sampleName, fastqFileNameVec] : sampleConfigTupleVec_) {
            std::cerr << "[" << __func__ << "::" << getTime() << "] " << "Processing sample: " << sampleName << "\n\n";
            kmer_read_hip(fastqFileNameVec);
            genotype(sampleName);
            ConstructIndexClassPtr_->reset();
        }
    }

    // Varigraph::kmer_read (src/varigraph.cpp:185-209): FastqKmer -> FastqKmerHip
    void kmer_read_hip(const std::vector<std::string>& fastqFileNameVec) {
        auto& table = ConstructIndexClassPtr_->mGraphKmerHashHapStrMap;
        if (!uploaded_) {
            keys_.reserve(table.size());
            for (const auto& kv : table) keys_.push_back(kv.first);
            if (vgmi_table_upload(ctx_, keys_.data(), keys_.size(), kmerLen_) != VGMI_OK) die(vgmi_last_error(ctx_));
            uploaded_ = true;
        }
        std::vector<uint8_t> cov(keys_.size());
        uint64_t readBase = 0;
        try {
            vgh::FastqKmerHip fk(ctx_, fastqFileNameVec, kmerLen_, threads_);
            fk.build_fastq_index();
            fk.fetch(cov.data(), nullptr, nullptr);
            readBase = fk.mReadBase;
        } catch (const std::exception& e) {
            die(e.what());
        }
        for (size_t i = 0; i < keys_.size(); ++i) table[keys_[i]].c = cov[i];   // src/fastq_kmer.cpp:132-138
        ReadDepth_ = readBase / (float)ConstructIndexClassPtr_->mGenomeSize;     // src/varigraph.cpp:198
        cal_ave_cov_kmer();                                                      // reference, unchanged
    }
};
