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
 * The C++ classes behind it (GraphIndex, FastxReader, FastqKmerHip) are in
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

/* FASTA/Q files -> '\n'-joined read block appended to a caller buffer (for tests of the parser).
 * Returns the number of reads, or <0.  *read_base gets sum(seq.l). */
int64_t vgh_fastx_read_all(const char *path, char **block_out, size_t *n_bytes_out, uint64_t *read_base);
void vgh_free(void *p);

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

#ifdef __cplusplus
}
#endif
#endif
