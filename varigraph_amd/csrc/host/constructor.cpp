// constructor.cpp -- see constructor.hpp
#include "constructor.hpp"


#include <fcntl.h>
#include <unistd.h>
#include <memory>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iterator>
#include <map>
#include <mutex>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "../vgmi_device.h"
#include "fastx_reader.hpp"
#include "graph_index.hpp"
#include "make_mbf.hpp"
#include "node_flanks.hpp"
#include "stl_order_map.hpp"
#include "vgmi.h"

namespace vgh {

namespace {

// split() of src/strip_split_join.cpp:30-51: every delimiter separates, empty tokens are kept, "" gives nothing
std::vector<std::string> split(const std::string& str, const std::string& delim)
{
    std::vector<std::string> res;
    if (str.empty()) return res;
    size_t from = 0;
    for (;;) {
        const size_t pos = str.find(delim, from);
        if (pos == std::string::npos) {
            res.push_back(str.substr(from));
            return res;
        }
        res.push_back(str.substr(from, pos - from));
        from = pos + delim.size();
    }
}

std::string join(const std::vector<std::string>& v, const std::string& delim)
{
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) {
        s += v[i];
        if (i + 1 != v.size()) s += delim;
    }
    return s;
}

// construct_index::gt_split (src/construct_index.cpp:1616-1650)
std::vector<std::string> gt_split(const std::string& gt)
{
    std::vector<std::string> out;
    if (gt == ".") return out;
    if (gt.find("/") != std::string::npos) return split(gt, "/");
    if (gt.find("|") != std::string::npos) return split(gt, "|");
    try {
        (void)std::stoul(gt);
    } catch (const std::exception&) {
        throw std::runtime_error("Error: GT is not separated by '/' or '|' -> " + gt);
    }
    out.push_back(gt);
    return out;
}

// lines of a plain / gzip / block-gzip file without their '\n' (include/GzChunkReader.hpp reads through gzopen);
// cohort VCFs are usually bgzip'd, whose blocks inflate on `decode_threads` workers (byte_source.hpp)
struct LineReader {
    std::unique_ptr<ByteSource> src;
    const unsigned char* cur = nullptr;
    const unsigned char* end = nullptr;
    bool eof = false;
    LineReader(const std::string& path, unsigned decode_threads) : src(ByteSource::open(path, decode_threads)) {}
    bool next(std::string& line)
    {
        line.clear();
        for (;;) {
            if (cur == end) {
                size_t n = 0;
                if (eof || !src->next_chunk(cur, n) || n == 0) {
                    eof = true;
                    cur = end = nullptr;
                    return !line.empty();
                }
                end = cur + n;
            }
            const unsigned char* nl = static_cast<const unsigned char*>(memchr(cur, '\n', (size_t)(end - cur)));
            if (nl) {
                line.append(reinterpret_cast<const char*>(cur), (size_t)(nl - cur));
                cur = nl + 1;
                return true;
            }
            line.append(reinterpret_cast<const char*>(cur), (size_t)(end - cur));
            cur = end;
        }
    }
};

// k-mer table record (kmerCovFreBitVec, include/construct_index.hpp:45-72) as the payload of a StlOrderMap node:
// byte 0 = c, byte 1 = f, then the haplotype bitmap

struct NodeView {
    uint32_t start;
    const GraphNode* gn;
};

// every k-mer key a sequence emits, in order, duplicates kept (the emitter of kmerBit::kmer_sketch_construct)
void emitted_keys(const std::string& s, uint32_t k, std::vector<uint64_t>& out)
{
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t fwd = 0, rev = 0;
    int l = 0, span = 0;
    const unsigned int len = (unsigned int)s.size();
    for (unsigned int i = 0; i < len; ++i) {
        const uint32_t c = vg_nt4((uint8_t)s[i]);
        if (c < 4) {
            span = l + 1 < (int)k ? l + 1 : (int)k;
            fwd = (fwd << 2 | c) & mask;
            rev = (rev >> 2) | (3ULL ^ c) << shift1;
            if (fwd == rev) continue;
            const uint64_t canon = fwd < rev ? fwd : rev;
            ++l;
            if (l >= (int)k && span < 256) out.push_back(vg_hash64(canon, mask) << 8 | (uint64_t)span);
        } else {
            l = 0;
            span = 0;
        }
    }
}

// graph.bin goes out through one 64 MiB buffer and write(2): the file is 1e8 fields of 1..8 bytes
class OutFile {
public:
    explicit OutFile(const std::string& path) : fd_(::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644)), buf_(64u << 20)
    {
        if (fd_ < 0) throw std::runtime_error("'" + path + "': No such file or directory.");
    }
    ~OutFile() { if (fd_ >= 0) ::close(fd_); }
    void write(const void* p, size_t n)
    {
        if (n == 0) return;   // empty strings / lists: their data() may be null
        if (n > buf_.size() - used_) {
            flush();
            if (n > buf_.size()) { raw(p, n); return; }
        }
        std::memcpy(buf_.data() + used_, p, n);
        used_ += n;
    }
    bool close()
    {
        flush();
        const bool ok = ::close(fd_) == 0 && ok_;
        fd_ = -1;
        return ok;
    }

private:
    void raw(const void* p, size_t n)
    {
        const char* c = static_cast<const char*>(p);
        while (n) {
            const ssize_t w = ::write(fd_, c, n);
            if (w <= 0) { ok_ = false; return; }
            c += w;
            n -= (size_t)w;
        }
    }
    void flush()
    {
        raw(buf_.data(), used_);
        used_ = 0;
    }
    int fd_;
    std::vector<char> buf_;
    size_t used_ = 0;
    bool ok_ = true;
};

template <typename T>
void put(OutFile& o, const T& v)
{
    o.write(&v, sizeof(T));
}

}  // namespace

ConstructStats construct_graph(vgmi_ctx* ctx, const ConstructConfig& cfg)
{
    using clock = std::chrono::steady_clock;
    const bool timing = getenv("VGH_TIMING") != nullptr;
    auto t_lap = clock::now();
    auto lap = [&](const char* what) {
        if (timing) std::fprintf(stderr, "[construct] %-18s %.3f s\n", what, std::chrono::duration<double>(clock::now() - t_lap).count());
        t_lap = clock::now();
    };
    ConstructStats st;
    if (cfg.k < 1 || cfg.k > 28) throw std::runtime_error("k must be in 1..28");
    const uint32_t ploidy = cfg.vcf_ploidy;

    // ---- build_fasta_index: the first record of a name provides the sequence, every record counts for the size
    // the big containers live in one block that the CLI leaves to process exit (ConstructConfig::release_memory)
    struct Heavy {
        std::unordered_map<std::string, std::string> fasta_seq;
        std::map<std::string, std::map<uint32_t, GraphNode>> graph;
        std::map<std::string, std::map<uint32_t, std::vector<std::string>>> vcf_info;
        std::unique_ptr<StlOrderMap> table;   // iteration order = record order of graph.bin (stl_order_map.hpp)
    };
    Heavy* heavy = new Heavy;
    std::unique_ptr<Heavy> heavy_owner(cfg.release_memory ? heavy : nullptr);
    std::unordered_map<std::string, std::string>& fasta_seq = heavy->fasta_seq;
    std::unordered_map<std::string, uint32_t> fasta_len;
    uint64_t genome_size = 0;
    {
        FastxReader rd(cfg.reference);
        while (rd.next() >= 0) {
            const std::string& s = rd.seq();
            genome_size += s.size();
            fasta_seq.emplace(rd.name(), std::string(s.data(), strnlen(s.data(), s.size())));
            fasta_len.emplace(rd.name(), (uint32_t)s.size());
            if (s.size() > UINT32_MAX) throw std::runtime_error("'" + rd.name() + "' length is greater than 4,294,967,295.");
        }
    }
    st.genome_size = genome_size;

    lap("reference FASTA");
    // ---- make_mbf on the device
    const auto t_bloom = clock::now();
    {
        if (genome_size < cfg.k) throw std::runtime_error("reference shorter than k");
        uint64_t m = 0;
        uint32_t n_hash = 0;
        vgmi_bloom_params(genome_size - cfg.k + 1, 0.01, &m, &n_hash);
        std::vector<uint64_t> seeds = cfg.bloom_seeds;
        if (seeds.empty()) seeds = reference_bloom_seeds(cfg.random_device_value, n_hash);
        if (seeds.size() != n_hash) throw std::runtime_error("wrong number of Bloom seeds");
        if (vgmi_bloom_create(ctx, m, n_hash, seeds.data()) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
        for (const auto& kv : fasta_seq) {
            if (kv.second.empty()) throw std::runtime_error("empty chromosome sequence (the reference aborts on assert(len > 0), kmer.cpp:27)");
            if (vgmi_bloom_add_seq(ctx, kv.second.data(), kv.second.size(), cfg.k) != VGMI_OK)
                throw std::runtime_error(vgmi_last_error(ctx));
        }
        st.bloom_bytes = m;
    }
    st.seconds_bloom = std::chrono::duration<double>(clock::now() - t_bloom).count();

    lap("Bloom build");
    // ---- construct: VCF -> nodes
    std::map<std::string, std::map<uint32_t, GraphNode>>& graph = heavy->graph;
    std::map<std::string, std::map<uint32_t, std::vector<std::string>>>& vcf_info = heavy->vcf_info;
    std::map<uint16_t, std::string> hap_map;
    hap_map[0] = "reference";
    uint16_t hap_num = 0;
    std::string vcf_head;
    uint64_t graph_base_num = genome_size;
    {
        uint32_t prev_start = 0, prev_end = 0;
        std::string prev_chr;
        LineReader lr(cfg.vcf, std::max(2u, cfg.threads));
        std::string line;
        auto ref_only_node = [&](const std::string& chr, uint32_t start, const std::string& seq) {
            GraphNode& n = graph[chr][start];
            n.start = start;
            n.seqs.push_back(seq);
            n.hap_gt.push_back(0);
        };
        while (lr.next(line)) {
            if (line.empty()) continue;
            if (line.find("##FORMAT") != std::string::npos) continue;
            if (line.find("#") != std::string::npos && line.find("#CHROM") == std::string::npos) {
                vcf_head += line + "\n";
                continue;
            }
            std::istringstream iss(line);
            std::vector<std::string> col(std::istream_iterator<std::string>{iss}, std::istream_iterator<std::string>());
            if (col.size() < 10)
                throw std::runtime_error("Error in '" + cfg.vcf + "': Number of columns in the VCF file is less than 10. Current column count: " +
                                         std::to_string(col.size()));
            if (line.find("#CHROM") != std::string::npos) {
                vcf_head += "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
                vcf_head += "##FORMAT=<ID=GQ,Number=1,Type=Float,Description=\"Genotype quality (phred-scaled 1 - max(GPP))\">\n";
                vcf_head += "##FORMAT=<ID=GPP,Number=1,Type=String,Description=\"Genotype posterior probabilities\">\n";
                vcf_head += "##FORMAT=<ID=NAK,Number=.,Type=Float,Description=\"Number of allele k-mers\">\n";
                vcf_head += "##FORMAT=<ID=CAK,Number=.,Type=Float,Description=\"Coverage of allele k-mers\">\n";
                vcf_head += "##FORMAT=<ID=UK,Number=1,Type=Integer,Description=\"Total number of unique kmers, capped at 255\">\n";
                vcf_head += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
                uint16_t hap = 1;
                for (size_t i = 9; i < col.size(); i++)
                    for (size_t j = 0; j < ploidy; j++) {
                        hap_map[hap] = col[i];
                        if (hap < UINT16_MAX) hap++;
                        else throw std::runtime_error("Error: The number of haplotypes exceeds the maximum limit of 65535.");
                    }
                hap_num = (uint16_t)hap_map.size();
                continue;
            }
            const std::string chr = col[0];
            const uint32_t ref_start = (uint32_t)std::stoul(col[1]);
            std::string ref_seq = col[3];
            const uint32_t ref_end = ref_start + (uint32_t)ref_seq.size() - 1;
            const std::vector<std::string> alts = split(col[4], ",");
            const std::vector<std::string> format = split(col[8], ":");
            const size_t gt_index = (size_t)(std::find(format.begin(), format.end(), "GT") - format.begin());
            if (gt_index == format.size()) throw std::runtime_error("Error: Genotype (GT) information is missing in FORMAT: " + line);
            auto sample_gt = [&](size_t i) -> std::vector<std::string> {
                const std::vector<std::string> f = split(col[i], ":");
                if (gt_index >= f.size()) throw std::runtime_error("Error: Genotype (GT) information is missing in: " + line);
                return gt_split(f[gt_index]);
            };
            {   // vcf_construct: the site's columns as the output VCF will print them (appends on a repeated start)
                std::vector<std::string>& info = vcf_info[chr].emplace(ref_start, std::vector<std::string>()).first->second;
                for (size_t i = 0; i < col.size(); i++) {
                    if (i < 9) {
                        info.push_back(col[i]);
                        continue;
                    }
                    const std::vector<std::string> gt = sample_gt(i);
                    std::string txt;
                    if (gt.empty()) {
                        for (size_t j = 0; j < ploidy; j++) txt += j == 0 ? "0" : "|0";
                    } else if (gt.size() >= ploidy) {
                        for (size_t j = 0; j < ploidy; j++) txt += j == 0 ? gt[j] : "|" + gt[j];
                    } else {
                        txt += join(gt, "|");
                        for (size_t j = 0; j < (ploidy - gt.size()); j++) txt += "|0";
                    }
                    info.push_back(txt);
                }
            }
            auto fa = fasta_seq.find(chr);
            if (fa == fasta_seq.end()) throw std::runtime_error("Error: Chromosome '" + chr + "' not found in reference genome.");
            if (chr != prev_chr) prev_start = 0;
            if (prev_start == ref_start) {
                std::fprintf(stderr, "[construct] Warning: Multiple variants detected, skipping this site -> %s %u\n", chr.c_str(), ref_start);
                continue;
            } else if (prev_start > ref_start) {
                std::fprintf(stderr, "[construct] Warning: Variants are unsorted, skipping this site -> %s %u>%u\n", chr.c_str(), prev_start, ref_start);
                continue;
            }
            const std::string true_ref = fa->second.substr(ref_start - 1, ref_seq.length());
            if (true_ref != ref_seq) {
                std::fprintf(stderr, "[construct] Warning: Sequence discrepancy detected between reference genome and VCF. Replacing with sequence from reference genome -> %s\t%u\n",
                             chr.c_str(), ref_start);
                ref_seq = true_ref;
            }
            if (chr != prev_chr) {
                if (prev_end > 0 && prev_end < fasta_seq[prev_chr].length()) {   // tail of the previous chromosome
                    const uint32_t s0 = prev_end + 1, e0 = (uint32_t)fasta_seq[prev_chr].length();
                    ref_only_node(prev_chr, s0, fasta_seq[prev_chr].substr(s0 - 1, e0 - s0 + 1));
                }
                if (ref_start > 1) ref_only_node(chr, 1, fa->second.substr(0, ref_start - 1));
            } else {
                const uint32_t s0 = prev_end + 1, e0 = ref_start - 1;
                if (s0 <= e0) ref_only_node(chr, s0, fa->second.substr(s0 - 1, e0 - s0 + 1));
            }
            GraphNode& node = graph[chr][ref_start];
            node.start = ref_start;
            node.seqs.push_back(ref_seq);
            node.hap_gt.push_back(0);
            node.seqs.insert(node.seqs.end(), alts.begin(), alts.end());
            graph_base_num += std::accumulate(alts.begin(), alts.end(), 0, [](int sum, const std::string& s) { return sum + s.size(); });
            if (node.seqs.size() > UINT16_MAX) throw std::runtime_error("Error: The number of haplotypes exceeds the maximum limit of 65535.");
            uint16_t hap = 1;
            for (size_t i = 9; i < col.size(); i++) {
                std::vector<std::string> gt = sample_gt(i);
                if (gt.size() > ploidy) gt.resize(ploidy);
                while (gt.size() < ploidy) gt.push_back("0");
                for (size_t j = 0; j < gt.size(); j++) {
                    node.hap_gt.push_back(gt[j] == "." ? (uint16_t)0 : (uint16_t)std::stoul(gt[j]));
                    if (hap < UINT16_MAX) hap++;
                    else throw std::runtime_error("Error: The number of haplotypes exceeds the maximum limit of 65535.");
                }
            }
            prev_start = ref_start;
            prev_end = ref_end;
            prev_chr = chr;
        }
        if (prev_end < fasta_seq[prev_chr].length()) {   // tail of the last chromosome
            const uint32_t s0 = prev_end + 1, e0 = (uint32_t)fasta_seq[prev_chr].length();
            ref_only_node(prev_chr, s0, fasta_seq[prev_chr].substr(s0 - 1, e0 - s0 + 1));
        }
    }

    lap("VCF -> nodes");
    // ---- index: per variant node and haplotype, the k-mers of allele + flanks with their Bloom count / presence.
    // Chunks of nodes go through three phases: (A, threads) flank sequences and emitted keys per node; one batched
    // Bloom query on the device; (C, threads) index_run's bookkeeping per node; then ConstructIndex::index's merge into
    // the table, sequentially and in node order (it fixes the record order of graph.bin).
    const auto t_index = clock::now();
    std::unique_ptr<StlOrderMap>& table = heavy->table;
    size_t table_bitlen = 0;
    {
        struct HapWork {   // one (node, haplotype) sequence whose k-mers were emitted
            uint16_t hap, gt;
            size_t key_begin, key_end;   // into the node's keys, then (after concatenation) into the chunk's
        };
        struct NodeWork {
            uint32_t ni = 0;
            GraphNode* node = nullptr;
            std::vector<HapWork> haps;
            std::vector<uint64_t> keys;
            std::unordered_map<uint64_t, std::vector<int8_t>> kept;   // phase C only: its iteration order is the node's k-mer order
            std::vector<uint64_t> kept_keys;                          // ... flattened by the same thread
            std::vector<int8_t> kept_bits;                            // kept_keys.size() * bitlen
            std::map<uint64_t, uint8_t> multi;
        };
        const uint32_t n_threads = std::max(1u, cfg.threads);
        auto parallel_for = [&](size_t n, const std::function<void(size_t)>& fn) {
            std::atomic<size_t> next{0};
            std::string error;
            std::mutex emu;
            auto worker = [&]() {
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= n) return;
                    try {
                        fn(i);
                    } catch (const std::exception& e) {
                        std::lock_guard<std::mutex> lk(emu);
                        if (error.empty()) error = e.what();
                        next.store(n);
                        return;
                    }
                }
            };
            std::vector<std::thread> pool;
            for (uint32_t t = 1; t < n_threads && t < n; ++t) pool.emplace_back(worker);
            worker();
            for (auto& th : pool) th.join();
            if (!error.empty()) throw std::runtime_error(error);
        };
        const size_t chunk_nodes = 20000;
        double t_a = 0, t_q = 0, t_c = 0, t_m = 0;
        auto since = [](clock::time_point t) { return std::chrono::duration<double>(clock::now() - t).count(); };
        for (auto& [chr, nodes] : graph) {
            std::vector<NodeView> view;
            view.reserve(nodes.size());
            for (auto& kv : nodes) view.push_back(NodeView{kv.first, &kv.second});
            std::vector<uint32_t> variant;
            for (uint32_t ni = 0; ni < view.size(); ++ni)
                if (view[ni].gn->hap_gt.size() != 1) variant.push_back(ni);
            st.n_variant_nodes += variant.size();
            const std::string& chr_name = chr;
            for (size_t c0 = 0; c0 < variant.size(); c0 += chunk_nodes) {
                const size_t c1 = std::min(variant.size(), c0 + chunk_nodes);
                std::vector<NodeWork> work(c1 - c0);
                // ---- phase A
                auto t_ph = clock::now();
                parallel_for(work.size(), [&](size_t w) {
                    NodeWork& nw = work[w];
                    nw.ni = variant[c0 + w];
                    GraphNode& node = *const_cast<GraphNode*>(view[nw.ni].gn);
                    nw.node = &node;
                    uint16_t hap = 0;
                    for (const uint16_t gt : node.hap_gt) {
                        if (cfg.fast && hap > 0 && gt == 0) {   // --fast: skip the haplotypes of VCF samples that are all-reference here
                            const uint16_t group = (hap - 1) / ploidy;
                            const uint16_t l = group * ploidy + 1, r = (group + 1) * ploidy;
                            const uint16_t sum = std::accumulate(node.hap_gt.begin() + l, node.hap_gt.begin() + r + 1, 0);
                            if (sum == 0) {
                                ++hap;
                                continue;
                            }
                        }
                        if (gt >= node.seqs.size())
                            throw std::runtime_error("Error: The node '" + chr_name + "-" + std::to_string(node.start) +
                                                     "' lacks sequence information for haplotype " + std::to_string(gt) + ".");
                        std::string seq = node.seqs[gt];
                        const auto fl = node_flanks(view, nw.ni, hap, gt, seq, cfg.k - 1);
                        seq = fl.first + seq + fl.second;
                        if (seq.empty()) throw std::runtime_error("empty allele sequence (the reference aborts on assert(len > 0))");
                        HapWork hw;
                        hw.hap = hap;
                        hw.gt = gt;
                        hw.key_begin = nw.keys.size();
                        emitted_keys(seq, cfg.k, nw.keys);
                        hw.key_end = nw.keys.size();
                        nw.haps.push_back(hw);
                        ++hap;
                    }
                });
                // ---- one Bloom batch for the chunk
                std::vector<uint64_t> keys;
                size_t total = 0;
                for (const NodeWork& nw : work) total += nw.keys.size();
                keys.reserve(total);
                t_a += since(t_ph);
                t_ph = clock::now();
                for (NodeWork& nw : work) {
                    const size_t base = keys.size();
                    for (HapWork& hw : nw.haps) {
                        hw.key_begin += base;
                        hw.key_end += base;
                    }
                    keys.insert(keys.end(), nw.keys.begin(), nw.keys.end());
                    std::vector<uint64_t>().swap(nw.keys);
                }
                std::vector<uint8_t> cnt(keys.size()), fnd(keys.size());
                if (!keys.empty() && vgmi_bloom_query(ctx, keys.data(), keys.size(), cnt.data(), fnd.data()) != VGMI_OK)
                    throw std::runtime_error(vgmi_last_error(ctx));
                st.bloom_queries += keys.size();
                t_q += since(t_ph);
                t_ph = clock::now();
                // ---- phase C: index_run's bookkeeping per node
                parallel_for(work.size(), [&](size_t w) {
                    NodeWork& nw = work[w];
                    GraphNode& node = *nw.node;
                    const size_t bitlen = (node.hap_gt.size() >> 3) + 1;
                    uint8_t min_fre = UINT8_MAX;
                    std::unordered_map<uint64_t, std::pair<std::vector<int8_t>, uint8_t>> by_key;
                    for (const HapWork& hw : nw.haps) {
                        std::map<uint8_t, std::unordered_set<uint64_t>> by_fre;
                        std::unordered_map<uint64_t, uint8_t> present;
                        for (size_t j = hw.key_begin; j < hw.key_end; ++j) {
                            by_fre.emplace(cnt[j], std::unordered_set<uint64_t>()).first->second.insert(keys[j]);
                            present[keys[j]] = fnd[j];
                        }
                        const uint16_t q = hw.hap >> 3, r = hw.hap & 7;
                        for (const auto& [fre, set] : by_fre) {
                            for (const uint64_t key : set) {
                                auto it = by_key.emplace(key, std::make_pair(std::vector<int8_t>(bitlen, 0), fre)).first;
                                it->second.second = fre;
                                it->second.first[q] |= (int8_t)(1 << r);
                                // also in the reference genome while not on haplotype 0 of this node: flag bit
                                if (hw.gt != 0 && present[key] && (it->second.first[0] & 1) == 0)
                                    it->second.first.back() |= (int8_t)(1u << 7);
                            }
                            min_fre = std::min(min_fre, fre);
                        }
                    }
                    if (min_fre == 0 || cfg.use_unique_kmers) min_fre = 1;
                    for (auto it = by_key.begin(); it != by_key.end();) {
                        if (it->second.second <= min_fre) {
                            nw.kept.emplace(it->first, std::move(it->second.first));
                            if (it->second.second >= 2) nw.multi.emplace(it->first, it->second.second);
                            it = by_key.erase(it);
                        } else {
                            ++it;
                        }
                    }
                    // the sequential merge below reads flat arrays; the map is taken apart by the thread that built it
                    nw.kept_keys.reserve(nw.kept.size());
                    nw.kept_bits.reserve(nw.kept.size() * bitlen);
                    for (const auto& [key, bits] : nw.kept) {
                        nw.kept_keys.push_back(key);
                        nw.kept_bits.insert(nw.kept_bits.end(), bits.begin(), bits.end());
                    }
                    std::unordered_map<uint64_t, std::vector<int8_t>>().swap(nw.kept);
                });
                t_c += since(t_ph);
                t_ph = clock::now();
                // ---- merge, in node order
                // the chunk's insert sequence first (node order, each node's k-mers in its own map's order: both reach the
                // file), then the inserts with the bucket and chain head of the keys ahead already on their way
                struct Pending { uint64_t key; const int8_t* bits; };
                std::vector<Pending> seq;
                std::vector<size_t> node_end(work.size());
                for (size_t w = 0; w < work.size(); ++w) {
                    NodeWork& nw = work[w];
                    GraphNode& node = *nw.node;
                    if (!nw.kept_keys.empty()) {
                        const size_t bl = nw.kept_bits.size() / nw.kept_keys.size();
                        if (!table) {
                            table_bitlen = bl;
                            table = std::make_unique<StlOrderMap>(2 + table_bitlen);
                        }
                        if (bl != table_bitlen) throw std::runtime_error("haplotype bitmaps of different lengths");
                        node.kmer_hash.insert(node.kmer_hash.end(), nw.kept_keys.begin(), nw.kept_keys.end());
                        for (size_t j = 0; j < nw.kept_keys.size(); ++j) seq.push_back(Pending{nw.kept_keys[j], nw.kept_bits.data() + j * bl});
                    }
                    node_end[w] = seq.size();
                }
                size_t i = 0;
                for (size_t w = 0; w < work.size(); ++w) {
                    for (; i < node_end[w]; ++i) {
                        if (i + 12 < seq.size()) table->prefetch(seq[i + 12].key);
                        const int8_t* bits = seq[i].bits;
                        const auto ins = table->emplace(seq[i].key);
                        uint8_t* e = table->payload(ins.first);
                        if (ins.second) {
                            std::memcpy(e + 2, bits, table_bitlen);
                            e[1]++;
                        } else {
                            for (size_t b = 0; b < table_bitlen; b++) e[2 + b] |= (uint8_t)bits[b];
                            if (e[1] < UINT8_MAX) e[1]++;
                        }
                    }
                    for (const auto& [key, fre] : work[w].multi) {   // after the node's own inserts, before the next node's
                        const uint32_t id = table ? table->find(key) : StlOrderMap::kNil;
                        if (id == StlOrderMap::kNil) throw std::runtime_error("The k-mer hash '" + std::to_string(key) + "' is not found in the table.");
                        uint8_t* e = table->payload(id);
                        if (e[1] == 1) e[1] += fre - 1;
                    }
                }
                t_m += since(t_ph);
            }
        }
        if (timing)
            std::fprintf(stderr, "[construct]   flanks+keys %.3f s, Bloom batch %.3f s, per-node bookkeeping %.3f s, merge %.3f s\n", t_a, t_q, t_c, t_m);
    }
    st.seconds_index = std::chrono::duration<double>(clock::now() - t_index).count();

    lap("index");
    // ---- save_index (byte layout: SURVEY.md Appendix A)
    {
        OutFile o(cfg.out);
        put<uint64_t>(o, graph_base_num);
        put<uint32_t>(o, cfg.k);
        put<uint32_t>(o, ploidy);
        put<uint32_t>(o, (uint32_t)vcf_head.length());
        o.write(vcf_head.data(), (size_t)vcf_head.length());
        put<uint32_t>(o, (uint32_t)vcf_info.size());
        for (const auto& [chr, sites] : vcf_info) {
            put<uint32_t>(o, (uint32_t)chr.length());
            o.write(chr.data(), (size_t)chr.length());
            put<uint32_t>(o, fasta_len.at(chr));
            put<uint32_t>(o, (uint32_t)sites.size());
            for (const auto& [start, fields] : sites) {
                put<uint32_t>(o, start);
                put<uint32_t>(o, (uint32_t)fields.size());
                for (const auto& f : fields) {
                    put<uint32_t>(o, (uint32_t)f.length());
                    o.write(f.data(), (size_t)f.length());
                }
            }
        }
        put<uint16_t>(o, hap_num);
        for (const auto& [idx, name] : hap_map) {
            put<uint16_t>(o, idx);
            put<uint32_t>(o, (uint32_t)name.length());
            o.write(name.data(), (size_t)name.length());
        }
        put<uint32_t>(o, (uint32_t)graph.size());
        for (const auto& [chr, nodes] : graph) {
            put<uint32_t>(o, (uint32_t)chr.length());
            o.write(chr.data(), (size_t)chr.length());
            put<uint32_t>(o, (uint32_t)nodes.size());
            for (const auto& [start, n] : nodes) {
                put<uint32_t>(o, start);
                put<uint32_t>(o, (uint32_t)n.seqs.size());
                for (const auto& s : n.seqs) {
                    put<uint32_t>(o, (uint32_t)s.length());
                    o.write(s.data(), (size_t)s.length());
                }
                put<uint32_t>(o, (uint32_t)n.hap_gt.size());
                o.write(reinterpret_cast<const char*>(n.hap_gt.data()), (size_t)(sizeof(uint16_t) * n.hap_gt.size()));
                put<uint32_t>(o, (uint32_t)n.kmer_hash.size());
                o.write(reinterpret_cast<const char*>(n.kmer_hash.data()), (size_t)(sizeof(uint64_t) * n.kmer_hash.size()));
            }
        }
        put<uint64_t>(o, (uint64_t)0);   // ReadBase
        if (table) {
            const uint64_t bl = table_bitlen;
            const std::vector<uint32_t> order = table->order();
            for (size_t i = 0; i < order.size(); ++i) {
                if (i + 16 < order.size()) table->prefetch_record(order[i + 16]);
                const uint32_t id = order[i];
                const uint8_t* e = table->payload(id);
                put<uint64_t>(o, table->key(id));
                put<uint8_t>(o, e[0]);
                put<uint8_t>(o, e[1]);
                put<uint64_t>(o, bl);
                o.write(e + 2, bl);
            }
        }
        if (!o.close()) throw std::runtime_error("'" + cfg.out + "': write error");
    }
    lap("save");
    st.graph_base_num = graph_base_num;
    st.n_kmers = table ? table->size() : 0;
    st.n_haplotypes = hap_map.size();
    return st;
}

}  // namespace vgh
