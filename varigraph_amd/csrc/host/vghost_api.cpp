// vghost_api.cpp -- C API of include/vghost.h over the C++ host classes
#include "vghost.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>

#include "fastq_kmer_hip.hpp"
#include "fast_inflate.hpp"
#include "fastx_reader.hpp"
#include "genotyper.hpp"
#include "graph_index.hpp"
#include "make_mbf.hpp"

struct vgh_graph {
    vgh::GraphIndex g;
};
struct vgh_genotyper {
    vgh::Genotyper gt;
    explicit vgh_genotyper(const vgh::GraphIndex& g) : gt(g) {}
};

static thread_local std::string g_err;

extern "C" {

const char* vgh_last_error(void) { return g_err.c_str(); }

int vgh_graph_load(const char* path, vgh_graph** out)
{
    if (!path || !out) return VGMI_E_INVALID;
    *out = nullptr;
    try {
        auto* h = new vgh_graph();
        try {
            h->g.load(path);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

void vgh_graph_free(vgh_graph* g) { delete g; }

int vgh_reads_index_save(const vgh_graph* h, const uint8_t* cov, uint64_t read_base, const char* path)
{
    if (!h || !cov || !path) return VGMI_E_INVALID;
    try {
        h->g.save_reads_index(path, cov, read_base);
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

int vgh_reads_index_load(const vgh_graph* h, const char* path, uint8_t* cov, uint64_t* read_base)
{
    if (!h || !cov || !path || !read_base) return VGMI_E_INVALID;
    try {
        h->g.load_reads_index(path, cov, *read_base);
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

int vgh_graph_get_info(const vgh_graph* h, vgh_graph_info* info)
{
    if (!h || !info) return VGMI_E_INVALID;
    const vgh::GraphIndex& g = h->g;
    info->graph_base_num = g.graph_base_num;
    info->genome_size = g.genome_size;
    info->n_keys = g.keys.size();
    info->bitlen = g.bitlen;
    info->n_variant_nodes = g.node_off.size() - 1;
    info->n_node_entries = g.node_key_index.size();
    info->k = g.k;
    info->vcf_ploidy = g.vcf_ploidy;
    info->hap_num = g.hap_num;
    info->n_chromosomes = (uint32_t)g.chr_names.size();
    return VGMI_OK;
}

const uint64_t* vgh_graph_keys(const vgh_graph* h) { return h->g.keys.data(); }
const uint8_t* vgh_graph_f(const vgh_graph* h) { return h->g.f.data(); }
const int8_t* vgh_graph_bitvec(const vgh_graph* h) { return h->g.bitvec.data(); }
const uint8_t* vgh_graph_hom_flag(const vgh_graph* h) { return h->g.hom_flag.data(); }
const uint64_t* vgh_graph_node_off(const vgh_graph* h) { return h->g.node_off.data(); }
const uint32_t* vgh_graph_node_key_index(const vgh_graph* h) { return h->g.node_key_index.data(); }
const uint32_t* vgh_graph_node_start(const vgh_graph* h) { return h->g.node_start.data(); }
const uint32_t* vgh_graph_node_chr(const vgh_graph* h) { return h->g.node_chr.data(); }
const char* vgh_graph_chr_name(const vgh_graph* h, uint32_t chr)
{
    return chr < h->g.chr_names.size() ? h->g.chr_names[chr].c_str() : nullptr;
}

int vgh_graph_upload(const vgh_graph* h, vgmi_ctx* ctx)
{
    if (!h || !ctx) return VGMI_E_INVALID;
    int rc = h->g.upload(ctx);
    if (rc) g_err = vgmi_last_error(ctx);
    return rc;
}

int64_t vgh_fastx_read_all_mt(const char* path, uint32_t decode_threads, char** block_out, size_t* n_bytes_out,
                              uint64_t* read_base, char source_kind[8])
{
    if (!path || !block_out || !n_bytes_out) return VGMI_E_INVALID;
    try {
        vgh::FastxReader rd(path, decode_threads ? decode_threads : 1);
        if (source_kind) {
            strncpy(source_kind, rd.source_kind(), 7);
            source_kind[7] = 0;
        }
        std::string block;
        int64_t n = 0;
        uint64_t rb = 0;
        while (rd.next() >= 0) {
            const std::string& s = rd.seq();
            block.append(s.data(), strnlen(s.data(), s.size()));
            block.push_back('\n');
            rb += s.size();
            ++n;
        }
        char* p = static_cast<char*>(malloc(block.size() ? block.size() : 1));
        if (!p) return VGMI_E_NOMEM;
        memcpy(p, block.data(), block.size());
        *block_out = p;
        *n_bytes_out = block.size();
        if (read_base) *read_base = rb;
        return n;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

int64_t vgh_fastx_read_all(const char* path, char** block_out, size_t* n_bytes_out, uint64_t* read_base)
{
    return vgh_fastx_read_all_mt(path, 1, block_out, n_bytes_out, read_base, nullptr);
}

int vgh_write_vcf_gz(const char* path, const char* text, size_t n_bytes, uint32_t threads)
{
    if (!path || (!text && n_bytes)) return VGMI_E_INVALID;
    try {
        vgh::Genotyper::write_gz(path, std::string(text ? text : "", n_bytes), threads);
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

void vgh_free(void* p) { free(p); }

uint32_t vgh_crc32(uint32_t crc, const void* data, size_t n) { return vgh::crc32_fast(crc, static_cast<const unsigned char*>(data), n); }

int vgh_bloom_reference_seeds(uint32_t random_device_value, uint32_t n_hash, uint64_t* seeds_out)
{
    if (!seeds_out) return VGMI_E_INVALID;
    const auto s = vgh::reference_bloom_seeds(random_device_value, n_hash);
    memcpy(seeds_out, s.data(), s.size() * 8);
    return VGMI_OK;
}

int vgh_make_mbf(vgmi_ctx* ctx, const char* fasta_path, uint32_t k, const uint64_t* seeds, uint32_t n_seeds,
                 uint32_t random_device_value, uint64_t* genome_size, uint64_t* m, uint32_t* n_hash)
{
    if (!ctx || !fasta_path) return VGMI_E_INVALID;
    try {
        std::vector<uint64_t> sv(seeds, seeds + (seeds ? n_seeds : 0));
        const vgh::MbfResult r = vgh::make_mbf(ctx, fasta_path, k, sv, random_device_value);
        if (genome_size) *genome_size = r.genome_size;
        if (m) *m = r.m;
        if (n_hash) *n_hash = r.n_hash;
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

int vgh_sample_count(const vgh_graph* h, vgmi_ctx* ctx, const char* const* fastq_paths, size_t n_files, uint32_t threads,
                     uint32_t sample_ploidy, int use_depth, uint8_t* cov_out, uint8_t* cov_node_out, uint64_t* hist_out,
                     vgh_sample_stats* stats)
{
    if (!h || !ctx || (!fastq_paths && n_files)) return VGMI_E_INVALID;
    try {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::string> files(fastq_paths, fastq_paths + n_files);
        vgh::FastqKmerHip fk(ctx, files, h->g.k, threads);
        fk.build_fastq_index();
        uint64_t hist[256];
        fk.fetch(cov_out, cov_node_out, hist);
        if (hist_out) memcpy(hist_out, hist, sizeof hist);
        if (stats) {
            memset(stats, 0, sizeof *stats);
            stats->read_base = fk.mReadBase;
            stats->n_reads = fk.mReadNum;
            vgh::CoverageStats cs;
            const bool ok = vgh::coverage_stats(hist, fk.mReadBase, h->g.genome_size, sample_ploidy, use_depth != 0, cs);
            stats->read_depth = cs.read_depth;
            stats->hap_kmer_coverage = cs.hap_kmer_coverage;
            stats->max_coverage = cs.max_coverage;
            stats->hom_coverage = cs.hom_coverage;
            stats->seconds_kernel = fk.kernel_seconds();
            stats->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (!ok) {
                g_err = "Failed to retrieve depth information of k-mers from the sequencing data. Please verify your data.";
                return VGMI_E_STATE;
            }
        }
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        const std::string m = e.what();
        return m.find("empty read") != std::string::npos || m.find("zero-length") != std::string::npos ? VGMI_E_EMPTY_READ
                                                                                                       : VGMI_E_INVALID;
    }
}

int vgh_coverage_stats(const uint64_t hist[256], uint64_t read_base, uint64_t genome_size, uint32_t sample_ploidy,
                       int use_depth, vgh_sample_stats* stats)
{
    if (!hist || !stats) return VGMI_E_INVALID;
    vgh::CoverageStats cs;
    const bool ok = vgh::coverage_stats(hist, read_base, genome_size, sample_ploidy, use_depth != 0, cs);
    stats->read_base = read_base;
    stats->read_depth = cs.read_depth;
    stats->hap_kmer_coverage = cs.hap_kmer_coverage;
    stats->max_coverage = cs.max_coverage;
    stats->hom_coverage = cs.hom_coverage;
    if (!ok) {
        g_err = "Failed to retrieve depth information of k-mers from the sequencing data. Please verify your data.";
        return VGMI_E_STATE;
    }
    return VGMI_OK;
}

void vgh_genotype_config_default(vgh_genotype_config* cfg)
{
    if (!cfg) return;
    const vgh::GenotypeConfig d;
    cfg->sample_type = "het";
    cfg->sample_ploidy = d.sample_ploidy;
    cfg->haploid_num = d.haploid_num;
    cfg->chr_len_thread = d.chr_len_thread;
    cfg->transition = "rec";
    cfg->sv_only = d.sv_only;
    cfg->threads = d.threads;
    cfg->min_gq = d.min_gq;
}

int vgh_genotyper_create(const vgh_graph* g, vgh_genotyper** out)
{
    if (!g || !out) return VGMI_E_INVALID;
    *out = nullptr;
    try {
        *out = new vgh_genotyper(g->g);
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

void vgh_genotyper_free(vgh_genotyper* gt) { delete gt; }

int vgh_genotype(vgh_genotyper* gt, const uint8_t* cov, float hap_kmer_coverage, const char* sample_name,
                 const vgh_genotype_config* cfg, char** vcf_text_out, size_t* n_bytes_out)
{
    if (!gt || !cov || !sample_name || !cfg || !vcf_text_out || !n_bytes_out) return VGMI_E_INVALID;
    try {
        vgh::GenotypeConfig c;
        if (cfg->sample_type) c.sample_type = cfg->sample_type;
        c.sample_ploidy = cfg->sample_ploidy;
        c.haploid_num = cfg->haploid_num;
        c.chr_len_thread = cfg->chr_len_thread;
        if (cfg->transition) c.transition = cfg->transition;
        c.sv_only = cfg->sv_only != 0;
        c.threads = cfg->threads ? cfg->threads : 1;
        c.min_gq = cfg->min_gq;
        const std::string text = gt->gt.run(cov, hap_kmer_coverage, sample_name, c);
        char* buf = static_cast<char*>(malloc(text.size() + 1));
        if (!buf) return VGMI_E_NOMEM;
        memcpy(buf, text.data(), text.size());
        buf[text.size()] = 0;
        *vcf_text_out = buf;
        *n_bytes_out = text.size();
        return VGMI_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VGMI_E_INVALID;
    }
}

}  // extern "C"
