/*
 * vghost.h -- C API of the C++17 host side (libvghost.so): the callers and data formats either
 * side of the device hot path, restated from the reference for the MI355X build.
 *
 *   graph.bin reader + graph2node     ConstructIndex::load_index / graph2node
 *                                     (src/construct_index.cpp:911-1105, :710-751,1572-1603)
 *   FASTA/Q reader                    kseq_read semantics (include/kseq.h:192-232) as used by
 *                                     FastqKmer::fastq_file_open (src/fastq_kmer.cpp:65-187)
 *   sample counting driver            FastqKmer::build_fastq_index (src/fastq_kmer.cpp:41-52) over vgmi.h
 *   coverage statistics               Varigraph::kmer_read / cal_ave_cov_kmer / get_hom_kmer_c /
 *                                     cal_hap_kmer_cov (src/varigraph.cpp:185-243,308-362)
 *
 *   genotyping HMM + VCF text         GENOTYPE::genotype / for_bac_post_run / hidden_states / forward / backward /
 *                                     posterior / save (src/genotype.cpp), HaplotypeSelect (src/haplotype_select.cpp),
 *                                     find_node_up_down_seq (src/construct_index.cpp:1266-1549)
 *
 * The C++ classes behind it (GraphIndex, FastxReader, FastqKmerHip, Genotyper) are in
 * varigraph_amd/csrc/host/; INTEGRATION.md shows how they slot into the reference.
 */
#ifndef VGHOST_H
#define VGHOST_H
#include <stddef.h>
#include <stdint.h>

#include "vgmi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vgh_graph vgh_graph;

typedef struct vgh_graph_info {
    uint64_t graph_base_num;  /* mGraphBaseNum */
    uint64_t genome_size;     /* mGenomeSize after load(): sum of lengths of chromosomes in mVcfInfoMap (construct_index.cpp:959-960) */
    uint64_t n_keys;          /* mGraphKmerHashHapStrMap.size() */
    uint64_t bitlen;          /* BitVec bytes per key */
    uint64_t n_variant_nodes; /* nodes with hapGtVec.size() > 1 */
    uint64_t n_node_entries;  /* sum over variant nodes of kept k-mers (<= 128 each) */
    uint32_t k;               /* mKmerLen */
    uint32_t vcf_ploidy;      /* mVcfPloidy */
    uint32_t hap_num;         /* mHapNum */
    uint32_t n_chromosomes;
} vgh_graph_info;

const char *vgh_last_error(void);

/* graph.bin (plain or gzip) -> flat host arrays.  Key order = file record order. */
int vgh_graph_load(const char *path, vgh_graph **out);
void vgh_graph_free(vgh_graph *g);
int vgh_graph_get_info(const vgh_graph *g, vgh_graph_info *info);
const uint64_t *vgh_graph_keys(const vgh_graph *g);           /* n_keys */
const uint8_t *vgh_graph_f(const vgh_graph *g);               /* n_keys */
const int8_t *vgh_graph_bitvec(const vgh_graph *g);           /* n_keys * bitlen */
const uint8_t *vgh_graph_hom_flag(const vgh_graph *g);        /* n_keys; A8 predicate, see vgmi_flags_upload */
const uint64_t *vgh_graph_node_off(const vgh_graph *g);       /* n_variant_nodes + 1 */
const uint32_t *vgh_graph_node_key_index(const vgh_graph *g); /* n_node_entries, graph2node order */
const uint32_t *vgh_graph_node_start(const vgh_graph *g);     /* n_variant_nodes: nodeStart (1-based) */
const uint32_t *vgh_graph_node_chr(const vgh_graph *g);       /* n_variant_nodes: chromosome ordinal (lexicographic) */
const char *vgh_graph_chr_name(const vgh_graph *g, uint32_t chr);
/* table + node CSR + flags -> device context */
int vgh_graph_upload(const vgh_graph *g, vgmi_ctx *ctx);

/* The reference's per-sample dump of the k-mer table (FastqKmer::save_index / load_index, src/fastq_kmer.cpp:200-298; public but
 * never called by its CLI): u64 ReadBase, then every record of the graph's k-mer table as
 * u64 key | u8 c | u8 f | u64 bitLen | i8[bitLen] -- graph.bin's last section with the sample's coverage in c -- in the
 * iteration order of the reference's unordered_map after load_index (derived, csrc/host/stl_order_map.hpp).
 * save writes exactly the bytes the reference writes for the same counters; load accepts the records in any order (the
 * reference loads them into a map) and fails on a key the graph does not hold. cov: n_keys bytes in key order. */
int vgh_reads_index_save(const vgh_graph *g, const uint8_t *cov, uint64_t read_base, const char *path);
int vgh_reads_index_load(const vgh_graph *g, const char *path, uint8_t *cov, uint64_t *read_base);

/* FASTA/Q files -> '\n'-joined read block appended to a caller buffer (for tests of the parser).
 * Returns the number of reads, or <0.  *read_base gets sum(seq.l). */
int64_t vgh_fastx_read_all(const char *path, char **block_out, size_t *n_bytes_out, uint64_t *read_base);
/* same with `decode_threads` inflate workers for block-gzip (BGZF) input; source_kind (may be NULL) gets "plain",
 * "gzip" or "bgzf": how the bytes were decoded (csrc/host/byte_source.hpp) */
int64_t vgh_fastx_read_all_mt(const char *path, uint32_t decode_threads, char **block_out, size_t *n_bytes_out,
                              uint64_t *read_base, char source_kind[8]);
void vgh_free(void *p);
/* CRC-32 (gzip polynomial) of the ingest decoder, csrc/host/fast_inflate.hpp (for tests) */
uint32_t vgh_crc32(uint32_t crc, const void *data, size_t n);

/* construct side: ConstructIndex::build_fasta_index + make_mbf (src/construct_index.cpp:85-139,150-177)
 * with the Bloom filter on the device.  seeds == NULL: the seeds BloomFilter::_init_seeds
 * (src/counting_bloom_filter.cpp:80-87) would draw if std::random_device returned
 * random_device_value.  The filter stays in ctx (vgmi_bloom_fetch / vgmi_bloom_query). */
int vgh_bloom_reference_seeds(uint32_t random_device_value, uint32_t n_hash, uint64_t *seeds_out);
int vgh_make_mbf(vgmi_ctx *ctx, const char *fasta_path, uint32_t k, const uint64_t *seeds, uint32_t n_seeds,
                 uint32_t random_device_value, uint64_t *genome_size, uint64_t *m, uint32_t *n_hash);

typedef struct vgh_sample_stats {
    uint64_t read_base;       /* FastqKmer::mReadBase */
    uint64_t n_reads;
    float read_depth;         /* ReadDepth_ (varigraph.cpp:198) */
    float hap_kmer_coverage;  /* hapKmerCoverage_ (varigraph.cpp:360-362) */
    uint8_t max_coverage;     /* get_hom_kmer_c */
    uint8_t hom_coverage;     /* after the --use-depth override, if any */
    double seconds_total, seconds_kernel;
} vgh_sample_stats;

/* One sample: counts reset -> all fastq files -> finish + coverage statistics.
 * cov_out[n_keys], cov_node_out[n_node_entries], hist_out[256] may be NULL. */
int vgh_sample_count(const vgh_graph *g, vgmi_ctx *ctx, const char *const *fastq_paths, size_t n_files,
                     uint32_t threads, uint32_t sample_ploidy, int use_depth, uint8_t *cov_out,
                     uint8_t *cov_node_out, uint64_t *hist_out, vgh_sample_stats *stats);

/* Coverage statistics alone (Varigraph::kmer_read / cal_ave_cov_kmer): hist = masked coverage histogram
 * (vgmi_counts_finish).  Fills read_depth, hap_kmer_coverage, max_coverage, hom_coverage of *stats; returns
 * VGMI_E_STATE where the reference exits with "Failed to retrieve depth information". */
int vgh_coverage_stats(const uint64_t hist[256], uint64_t read_base, uint64_t genome_size, uint32_t sample_ploidy,
                       int use_depth, vgh_sample_stats *stats);

/* Genotyping HMM (host; x87 long double like the reference).  The object keeps the per-node k-mer lists, which the
 * forward pass prunes and which persist across samples exactly as in the reference (reset() does not restore them):
 * create one per graph, call vgh_genotype once per sample in `-s` order. */
typedef struct vgh_genotyper vgh_genotyper;
typedef struct vgh_genotype_config {   /* defaults of include/varigraph.hpp:49-68 via vgh_genotype_config_default */
    const char *sample_type;    /* -g: "het" | "hom" */
    uint32_t sample_ploidy;     /* --sample-ploidy */
    uint32_t haploid_num;       /* -n */
    uint32_t chr_len_thread;    /* --granularity, in bp */
    const char *transition;     /* -m: "rec" | "fre" */
    int sv_only;                /* --sv */
    uint32_t threads;           /* -t */
    float min_gq;               /* --min-support */
} vgh_genotype_config;
void vgh_genotype_config_default(vgh_genotype_config *cfg);
int vgh_genotyper_create(const vgh_graph *g, vgh_genotyper **out);
void vgh_genotyper_free(vgh_genotyper *gt);
/* cov: c of every key in graph.bin record order.  *vcf_text_out (vgh_free) = decompressed content of
 * <sample>.varigraph.vcf.gz. */
int vgh_genotype(vgh_genotyper *gt, const uint8_t *cov, float hap_kmer_coverage, const char *sample_name,
                 const vgh_genotype_config *cfg, char **vcf_text_out, size_t *n_bytes_out);
/* <sample>.varigraph.vcf.gz as `varigraph-mi genotype` writes it: block gzip (BGZF) of `text`, deflated by `threads`
 * workers; the decompressed bytes are what the reference's SAVE::save writes (src/save.cpp:11-30) */
int vgh_write_vcf_gz(const char *path, const char *text, size_t n_bytes, uint32_t threads);

#ifdef __cplusplus
}
#endif
#endif
