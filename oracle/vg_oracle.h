/*
 * vg_oracle.h -- CPU restatement of varigraph's per-sample genotyping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (varigraph_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Parity status: PINNED.  Every function below is checked against the real reference
 * (built from the unmodified sources under /root/reference by oracle/Makefile into
 * oracle/_ref/) and against the golden vectors in tests/golden/ that were dumped
 * from that build (tests/golden/make_golden.py).
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef VG_ORACLE_H
#define VG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A1: include/seq_nt4_table.hpp:5-22 */
extern const uint8_t vgo_nt4_table[256];

/* A2: include/hash64.hpp:5-14 */
uint64_t vgo_hash64(uint64_t key, uint64_t mask);

/* A3: the rolling canonical k-mer loop shared by src/kmer.cpp:20-53 (bf), :65-97
 * (construct), :110-149 (fastq), :161-200 (genotype), without the per-variant sink.
 * Writes every emitted key (hash64(min(fwd,rc))<<8 | k) in read order, duplicates
 * included, to out[] (capacity must be >= len).  Returns the number of keys.
 * Returns (size_t)-1 where the reference would abort (assert len>0 && 0<k<=28). */
size_t vgo_sketch(const char *s, size_t len, uint32_t k, uint64_t *out);

/* A6 + A3(find) + A5(increment): exact-membership table with saturating u8 counters.
 * src/kmer.cpp:140-142 (find) and src/fastq_kmer.cpp:128-139,167-178 (c++ if c<255). */
typedef struct vgo_table vgo_table;
vgo_table *vgo_table_new(const uint64_t *keys, size_t n); /* NULL on duplicate key / OOM */
void vgo_table_free(vgo_table *t);
size_t vgo_table_size(const vgo_table *t);
/* index of key in the keys[] given to vgo_table_new, or -1 */
int64_t vgo_table_find(const vgo_table *t, uint64_t key);
/* One read through kmer_sketch_fastq + the main-thread increment loop.  Returns the
 * number of hits (keys pushed, src/kmer.cpp:141), or -1 where the reference aborts. */
int64_t vgo_count_read(vgo_table *t, const char *s, size_t len, uint32_t k);
/* '\n'-joined block of reads (each read terminated by '\n'); adds sum(len) to *read_base
 * (src/fastq_kmer.cpp:105).  Empty reads return -1 (reference assert). */
int64_t vgo_count_block(vgo_table *t, const char *block, size_t n_bytes, uint32_t k,
                        uint64_t *read_base);
void vgo_table_counts(const vgo_table *t, uint8_t *c_out); /* c per key, input order */
void vgo_table_reset(vgo_table *t);                        /* include/construct_index.hpp:317-331 */

/* B1: src/counting_bloom_filter.cpp:70-77 */
uint64_t vgo_bloom_size(uint64_t n, double p);
uint32_t vgo_bloom_num_hashes(uint64_t n, uint64_t m);
/* B2: src/counting_bloom_filter.cpp:90-98 over src/MurmurHash3.cpp:255-332 with len=8;
 * the seed parameter is `unsigned int`, so only the low 32 bits of a stored seed count. */
uint64_t vgo_murmur_sum(uint64_t key, uint64_t seed);
/* B3: src/counting_bloom_filter.cpp:28-36 */
void vgo_bloom_add(uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh, uint64_t key);
/* B3 driver: src/kmer.cpp:20-53 (kmer_sketch_bf) for one sequence; returns #k-mers added or -1 */
int64_t vgo_bloom_add_seq(uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh,
                          const char *s, size_t len, uint32_t k);
/* B4: src/counting_bloom_filter.cpp:51-67 (count = min) and :40-47 (find = all non-zero) */
uint8_t vgo_bloom_count(const uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh,
                        uint64_t key);
int vgo_bloom_find(const uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh,
                   uint64_t key);

/* A8: src/varigraph.cpp:253-296 (get_hom_kmer).  bitvec is n rows of bitlen bytes
 * (bitlen = hap_num/8+1, graph.bin layout).  hist[c] = number of keys with c!=0, f<=1
 * that are carried by all vcf_ploidy haplotypes of at least one VCF sample. */
void vgo_hom_hist(const uint8_t *c, const uint8_t *f, const int8_t *bitvec, size_t bitlen,
                  size_t n, uint32_t hap_num, uint32_t vcf_ploidy, uint64_t hist[256]);
/* A8: src/varigraph.cpp:308-348 (get_hom_kmer_c).  Returns 0 and fills max_cov/hom_cov,
 * or -1 where the reference exits (no coverage >1). */
int vgo_hom_peak(const uint64_t hist[256], float read_depth, uint8_t *max_cov, uint8_t *hom_cov);
/* A7 + A8: src/varigraph.cpp:198 (ReadDepth_), :230-232 (--use-depth), :360-362 */
float vgo_read_depth(uint64_t read_base, uint64_t genome_size);
float vgo_hap_kmer_cov(uint8_t hom_cov, uint32_t sample_ploidy, float read_depth);
uint8_t vgo_use_depth_cov(float read_depth);

#ifdef __cplusplus
}
#endif
#endif
