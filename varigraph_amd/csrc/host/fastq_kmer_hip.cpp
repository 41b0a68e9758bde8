#include "fastq_kmer_hip.hpp"

#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "fastx_reader.hpp"
#include "vgmi.h"

namespace vgh {

namespace {

struct Block {
    std::vector<char> bytes;
    size_t n_reads = 0;
    uint64_t read_base = 0;
};

struct Channel {  // parser threads -> submitting thread
    std::mutex mu;
    std::condition_variable cv_full, cv_free;
    std::deque<std::unique_ptr<Block>> full;
    std::vector<std::unique_ptr<Block>> free_list;
    size_t producers = 0;
    std::string error;
};

void parse_file(const std::string& path, Channel& ch, size_t block_bytes, unsigned decode_threads)
{
    auto grab = [&]() {
        std::unique_lock<std::mutex> lk(ch.mu);
        ch.cv_free.wait(lk, [&] { return !ch.free_list.empty(); });
        auto b = std::move(ch.free_list.back());
        ch.free_list.pop_back();
        b->bytes.clear();
        b->n_reads = 0;
        b->read_base = 0;
        return b;
    };
    auto push = [&](std::unique_ptr<Block> b) {
        std::lock_guard<std::mutex> lk(ch.mu);
        ch.full.push_back(std::move(b));
        ch.cv_full.notify_one();
    };
    try {
        FastxReader rd(path, decode_threads);
        std::unique_ptr<Block> cur = grab();
        while (rd.next() >= 0) {  // stops at EOF (-1) and at the first truncated record (-2)
            const std::string& s = rd.seq();
            // `string sequence = ks->seq.s` (src/fastq_kmer.cpp:101) ends at the first NUL, while
            // mReadBase adds the full ks->seq.l (:105)
            const size_t len = strnlen(s.data(), s.size());
            if (len == 0) throw std::runtime_error("'" + path + "': empty read sequence (the reference aborts on assert(len > 0), kmer.cpp:124)");
            if (cur->bytes.size() + len + 1 > block_bytes && cur->n_reads) {
                push(std::move(cur));
                cur = grab();
            }
            cur->bytes.insert(cur->bytes.end(), s.data(), s.data() + len);
            cur->bytes.push_back('\n');
            cur->n_reads++;
            cur->read_base += s.size();
        }
        if (cur->n_reads) push(std::move(cur));
        else {
            std::lock_guard<std::mutex> lk(ch.mu);
            ch.free_list.push_back(std::move(cur));
        }
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(ch.mu);
        if (ch.error.empty()) ch.error = e.what();
    }
    std::lock_guard<std::mutex> lk(ch.mu);
    ch.producers--;
    ch.cv_full.notify_all();
}

}  // namespace

FastqKmerHip::FastqKmerHip(vgmi_ctx* ctx, const std::vector<std::string>& fastqFileNameVec, uint32_t kmerLen,
                           uint32_t threads, size_t block_bytes)
    : ctx_(ctx), files_(fastqFileNameVec), k_(kmerLen), threads_(threads ? threads : 1), block_bytes_(block_bytes)
{
}

void FastqKmerHip::build_fastq_index()
{
    if (files_.empty()) throw std::runtime_error("Parameter error: -f");  // src/fastq_kmer.cpp:42-45
    uint32_t k_tab = 0;
    if (vgmi_table_info(ctx_, nullptr, &k_tab, nullptr, nullptr) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx_));
    if (k_tab != k_) throw std::runtime_error("k-mer length differs from the uploaded graph table");
    if (vgmi_counts_reset(ctx_) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx_));
    mReadBase = 0;
    mReadNum = 0;

    Channel ch;
    const size_t n_par = std::min<size_t>(threads_, files_.size());
    // block-gzip inputs: the threads not parsing inflate blocks (byte_source.hpp); at least 2 per file
    const unsigned decode_threads = std::max<unsigned>(2u, (unsigned)(threads_ / n_par));
    for (size_t i = 0; i < 2 * n_par + 1; ++i) {
        auto b = std::make_unique<Block>();
        b->bytes.reserve(block_bytes_ + 1024);
        ch.free_list.push_back(std::move(b));
    }
    size_t next_file = 0;
    std::vector<std::thread> workers;
    auto start_more = [&]() {  // called with ch.mu held
        while (ch.producers < n_par && next_file < files_.size()) {
            ch.producers++;
            workers.emplace_back(parse_file, files_[next_file++], std::ref(ch), block_bytes_, decode_threads);
        }
    };
    std::string err;
    {
        std::unique_lock<std::mutex> lk(ch.mu);
        start_more();
        for (;;) {
            ch.cv_full.wait(lk, [&] { return !ch.full.empty() || ch.producers == 0; });
            if (!ch.error.empty() && err.empty()) err = ch.error;
            if (ch.full.empty()) {
                if (next_file < files_.size() && err.empty()) { start_more(); continue; }
                if (ch.producers == 0) break;
                continue;
            }
            auto b = std::move(ch.full.front());
            ch.full.pop_front();
            lk.unlock();
            if (err.empty()) {
                if (vgmi_reads_submit(ctx_, b->bytes.data(), b->bytes.size(), nullptr, b->n_reads) != VGMI_OK)
                    err = vgmi_last_error(ctx_);
                mReadBase += b->read_base;
                mReadNum += b->n_reads;
            }
            lk.lock();
            ch.free_list.push_back(std::move(b));
            ch.cv_free.notify_one();
            start_more();
        }
    }
    for (auto& t : workers) t.join();
    if (!err.empty()) throw std::runtime_error(err);
}

void FastqKmerHip::fetch(uint8_t* cov, uint8_t* cov_node, uint64_t* hist256)
{
    if (vgmi_counts_finish(ctx_, cov, cov_node, hist256) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx_));
    float ms = 0;
    uint64_t n = 0;
    vgmi_count_kernel_ms(ctx_, &ms, &n);
    kernel_s_ = ms / 1e3;
}

// src/varigraph.cpp:308-348 (get_hom_kmer_c) over the histogram of get_hom_kmer (:253-296), then
// :230-232 (--use-depth) and :360-362 (cal_hap_kmer_cov)
bool coverage_stats(const uint64_t hist[256], uint64_t read_base, uint64_t genome_size, uint32_t sample_ploidy,
                    bool use_depth, CoverageStats& out)
{
    out.read_depth = read_base / (float)genome_size;  // :198
    // the reference iterates a std::map holding only the coverages that occur
    uint8_t cov[256];
    uint64_t fre[256];
    size_t nb = 0;
    int index = -1, max_index = -1;
    uint8_t max_cov = 0, hom_cov = 0;
    uint64_t max_fre = 0;
    for (int v = 0; v < 256; ++v) {
        if (!hist[v]) continue;
        cov[nb] = (uint8_t)v;
        fre[nb] = hist[v];
        ++nb;
        ++index;
        if (v > 1 && hist[v] >= max_fre && v < UINT8_MAX) {
            max_index = index;
            max_cov = (uint8_t)v;
            max_fre = hist[v];
            hom_cov = (uint8_t)v;
        }
    }
    if (max_index == -1) return false;
    for (size_t i = (size_t)max_index + 1; i < nb - 1; i++) {
        if (cov[i] > out.read_depth) break;
        if (fre[i] >= fre[i - 1] && fre[i] >= fre[i + 1]) hom_cov = cov[i];
    }
    if (use_depth) hom_cov = out.read_depth * 0.8;
    out.max_coverage = max_cov;
    out.hom_coverage = hom_cov;
    out.hap_kmer_coverage = (hom_cov > 0 && sample_ploidy > 0)
                                ? static_cast<float>(hom_cov) / static_cast<float>(sample_ploidy)
                                : out.read_depth / static_cast<float>(sample_ploidy);
    return true;
}

}  // namespace vgh
