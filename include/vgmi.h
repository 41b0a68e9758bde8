/*
 * vgmi.h -- C ABI of the MI355X (gfx950) implementation of varigraph's per-sample genotyping
 * hot path.  This is the drop-in boundary: plain pointers and sizes, no C++ types, no torch.
 *
 * The reference has no plugin/FFI layer; its own CPU<->GPU seam is a pair of C++ classes whose
 * CUDA twins override one method each.  Every entry point below names the reference interface
 * it replaces (file:line relative to the reference tree).  INTEGRATION.md shows the C++ adapter
 * (`FastqKmerHip : FastqKmer`, `BloomFilterHip : BloomFilter`) a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative vgmi_status; vgmi_last_error() has text.
 *   - one context per (host thread, device); a context is not thread-safe; contexts are
 *     independent.  Nothing throws, nothing exits (the reference prints and exit()s; the CLI
 *     adapter reproduces that on top of the codes).
 *   - "host" pointers are ordinary host memory owned by the caller; "dev" pointers are device
 *     memory on the context's device, owned by the caller (e.g. a torch tensor), 16-byte aligned.
 *   - a *read block* is the '\n'-joined concatenation of read sequences (ASCII, any case, as in
 *     a FASTQ sequence line): every read is followed by exactly one '\n'.  This is the layout
 *     the reference GPU path builds on the host with 'N' separators (src/fastq_kmer.cu:171-175).
 *   - keys are the reference's graph k-mer keys: hash64(min(fwd,rc), 2^(2k)-1) << 8 | k
 *     (src/kmer.cpp:135-138, include/hash64.hpp:5-14), exactly as stored in graph.bin.
 */
#ifndef VGMI_H
#define VGMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vgmi_ctx vgmi_ctx;

typedef enum vgmi_status {
    VGMI_OK = 0,
    VGMI_E_INVALID = -1,      /* bad argument (NULL, k out of 1..28, misaligned device pointer ...) */
    VGMI_E_NO_DEVICE = -2,    /* no HIP device / bad ordinal  (reference: cudaSetDevice failure, main.cu:221,444) */
    VGMI_E_HIP = -3,          /* a HIP runtime call failed    (reference: gpuErrchk, include/cuda_error_handling.hpp:10-16) */
    VGMI_E_STATE = -4,        /* call order violated (e.g. reads before a table) */
    VGMI_E_DUPLICATE_KEY = -5,/* table keys not unique (the reference map cannot hold duplicates) */
    VGMI_E_BAD_KEY = -6,      /* key low byte != k or payload >= 2^(2k) */
    VGMI_E_EMPTY_READ = -7,   /* a zero-length read: the reference aborts on assert(len>0), src/kmer.cpp:124 */
    VGMI_E_NOMEM = -8
} vgmi_status;

/* ---- device / context -------------------------------------------------------------------
 * replaces: cudaSetDevice(config.gpu) + VarigraphKernelConfig{gpu,buffer}
 *           (main.cu:221-229,444-452; include/varigraph.cuh:19-28; flags --gpu/--buffer main.cu:99-100) */
int vgmi_device_count(void);
int vgmi_create(int device, size_t buffer_mib, vgmi_ctx **out);
void vgmi_destroy(vgmi_ctx *ctx);
/* ctx may be NULL: error of the last failed vgmi_create.  A context serves several threads at once (FASTQ streams, HMM parts, a broadcast
 * thread): the text is the CALLING thread's own last failure on this context when it has one, else the context's last. */
const char *vgmi_last_error(const vgmi_ctx *ctx);
/* free / total bytes of the context's device right now (hipMemGetInfo; free includes the working blocks this context keeps
 * between HMM calls and is not using, which the next call reuses or releases): callers that size optional device work (the HMM
 * recursion's score arrays) decide from it instead of from a fixed bound.  No reference counterpart. */
int vgmi_device_memory(vgmi_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
/* the HIP stream every kernel of this context is launched on (hipStream_t), for event timing */
void *vgmi_stream(vgmi_ctx *ctx);
int vgmi_device_of(vgmi_ctx *ctx, int *device);      /* the ordinal the context was created on */

/* ---- graph k-mer table (immutable after upload) -----------------------------------------
 * replaces: the `unordered_map<uint64_t,kmerCovFreBitVec>& GraphKmerHashHapStrMap` argument of
 *           FastqKmer / FastqKmerKernel (include/fastq_kmer.hpp:57-62, include/fastq_kmer.cuh:15-36):
 *           membership test of src/kmer.cpp:140 and the `.c` counters of src/fastq_kmer.cpp:132-138.
 * keys[i] keeps its index i: every per-key output below is in this order. */
int vgmi_table_upload(vgmi_ctx *ctx, const uint64_t *host_keys, size_t n_keys, uint32_t k);
/* Device image of the table (for one RCCL broadcast from the rank that parsed graph.bin;
 * the reference is single-device).  export copies it into a caller-owned device
 * buffer of vgmi_table_image_bytes(); import adopts such a buffer's contents on another rank. */
int vgmi_table_image_bytes(vgmi_ctx *ctx, size_t *bytes);
int vgmi_table_export(vgmi_ctx *ctx, void *dev_dst, size_t bytes);
int vgmi_table_import(vgmi_ctx *ctx, const void *dev_src, size_t bytes);
/* Same hand-over inside ONE process that drives several devices (`varigraph-mi genotype --gpus a,b,...`): dst adopts a
 * copy of src's table image through one device-to-device transfer (xGMI peer copy between different devices); the table
 * is built once per run, not once per device. */
int vgmi_table_clone(vgmi_ctx *dst, vgmi_ctx *src);
/* ... and between the PROCESSES of one node, one per GPU: ONE RCCL broadcast of the image over xGMI (librccl is loaded on first use).
 * Rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the others by any means it likes (`varigraph-mi genotype --procs`
 * forks its ranks around a shared page); then every rank calls vgmi_table_broadcast with its own context: the root's holds the
 * table, the others' adopt what arrives.  world = 1 is legal (a communicator of one).  No reference counterpart (single device). */
int vgmi_rccl_unique_id(void *id128);
int vgmi_table_broadcast(vgmi_ctx *ctx, int rank, int world, const void *id128);
/* The same in two steps, so that the communicator -- seconds of ncclCommInitRank, which needs neither a context nor a table -- comes
 * up BESIDE a rank's graph load and table build (its own thread), not behind them: vgmi_comm_create on `device` (every rank, any
 * time after the id exists), then vgmi_table_broadcast_comm with the rank's context on that device (root = rank 0), then
 * vgmi_comm_destroy.  A rank that cannot take part (the root without a table, a receiver without room for the image) tells the
 * others inside the collectives: every rank returns an error, none waits.  A rank that fails on its own between the collectives (a HIP
 * call, a collective's own error) aborts the communicator on its way out (ncclCommAbort), which ends its peers' pending collectives with
 * an error; the communicator is then good for vgmi_comm_destroy only.  vgmi_table_broadcast is the three calls in one. */
typedef struct vgmi_comm vgmi_comm;
int vgmi_comm_create(int device, int rank, int world, const void *id128, vgmi_comm **out);
int vgmi_table_broadcast_comm(vgmi_ctx *ctx, vgmi_comm *comm);
/* The root may take a snapshot of its image first (a device-to-device copy; no sample counted yet): the broadcast then sends the
 * snapshot and frees it, on a stream of its own -- so the root starts counting (which sets per-sample bits inside the live image)
 * without waiting for the communicator, and the broadcast leaves from a thread of the caller's whenever the communicator is up. */
int vgmi_table_snapshot(vgmi_ctx *ctx);
void vgmi_comm_destroy(vgmi_comm *comm);
int vgmi_table_info(vgmi_ctx *ctx, size_t *n_keys, uint32_t *k, size_t *n_slots, size_t *filter_bits);
/* Batched exact lookup, the table's `find`: index_out[i] = the index keys[i] has in the uploaded key array, 0xFFFFFFFF when
 * the table does not hold it.  Replaces the per-node loop of Varigraph graph2node (src/construct_index.cpp:710-751:
 * mGraphKmerHashHapStrMap.find(kmerHash) for every k-mer of every variant node) with one call over the concatenated node
 * lists.  Host pointers.  Reads the table only and works on a stream of its own: may be called while another thread
 * streams reads into the same context. */
int vgmi_table_lookup(vgmi_ctx *ctx, const uint64_t *host_keys, size_t n, uint32_t *index_out);
/* The k = 27 table of large graphs is keyed by the read's grid 16-mer: its number of 128-byte lines (0: not in
 * use) and how many (k-mer, 16-mer) pairs found no room near their home line and are served by the exact overflow table
 * (16-mers of repeats). Diagnostic only; no reference counterpart. */
int vgmi_xtable_info(vgmi_ctx *ctx, size_t *n_lines, size_t *overflow_pairs);
/* Since round 4 the default table of those graphs is the CONTEXT TABLE (VGMI_CTABLE=0 keeps the grid-16-mer
 * table): one 16-byte entry per occurrence of a 16-mer in a unitig of the key set, four to a 64-byte bucket.  Its number of
 * buckets (0: not in use), entries, the unitigs they came from, entries that sit behind their home bucket, and k-mers served by
 * the exact overflow table (16-mers of repeats).  Since round 5 also the table of such graphs with k = 19 .. 25 (flanks of k - 16
 * bases; VGMI_CTABLE_K=0 keeps them on the generic / literal kernels).  Diagnostic only; no reference counterpart. */
int vgmi_ctable_info(vgmi_ctx *ctx, size_t *n_buckets, size_t *n_entries, size_t *n_unitigs, size_t *moved_entries,
                     size_t *overflow_kmers);

/* Per-node k-mer lists in CSR form.
 * replaces: nodeSrt::GraphKmerHashHapStrMapIterVec built by ConstructIndex::graph2node
 *           (src/construct_index.cpp:710-751,1572-1603): node v owns key_index[node_off[v]..node_off[v+1]). */
int vgmi_nodes_upload(vgmi_ctx *ctx, const uint64_t *host_node_off, const uint32_t *host_key_index,
                      size_t n_nodes);
/* flag[i] != 0 iff key i has f<=1 and is carried by all vcfPloidy haplotypes of at least one VCF
 * sample: the sample-independent part of Varigraph::get_hom_kmer (src/varigraph.cpp:263-287). */
int vgmi_flags_upload(vgmi_ctx *ctx, const uint8_t *host_hom_flag);

/* ---- per sample ---------------------------------------------------------------------------
 * replaces: FastqKmer::build_fastq_index / FastqKmerKernel::build_fastq_index_kernel
 *           (src/fastq_kmer.cpp:41-187, src/fastq_kmer.cu:20-274) and ConstructIndex::reset
 *           (include/construct_index.hpp:317-331). */
int vgmi_counts_reset(vgmi_ctx *ctx);
/* One read block from host memory (copied to pinned staging, sent asynchronously; returns
 * before the kernel has run).  read_off[n_reads+1] = byte offset of each read start
 * (read_off[n_reads] == n_bytes); may be NULL, it is then derived from the '\n's when needed
 * (even k only). */
int vgmi_reads_submit(vgmi_ctx *ctx, const char *host_bases, size_t n_bytes,
                      const uint64_t *host_read_off, size_t n_reads);
/* Same, block already resident in device memory (no copy).  dev_read_off is required for even k. */
int vgmi_reads_submit_device(vgmi_ctx *ctx, const char *dev_bases, size_t n_bytes,
                             const uint64_t *dev_read_off, size_t n_reads);
/* Sum of read lengths submitted since the last reset: FastqKmer::mReadBase
 * (include/fastq_kmer.hpp:42, src/fastq_kmer.cpp:105). */
int vgmi_read_base(vgmi_ctx *ctx, uint64_t *read_base);
/* Waits for all submitted blocks, then (any output may be NULL)
 *   cov_out[n_keys]        c of every key = min(255, occurrences)             (src/fastq_kmer.cpp:133-137)
 *   cov_node_out[node_off[n_nodes]]  c gathered in node order: the per-node depth lookup
 *                          `iter->second.c` of src/genotype.cpp:546,660,1405-1408
 *   hist256_out[256]       #flagged keys per non-zero c: Varigraph::get_hom_kmer (src/varigraph.cpp:253-296)
 * Returns VGMI_E_EMPTY_READ if any submitted block held a zero-length read. */
int vgmi_counts_finish(vgmi_ctx *ctx, uint8_t *host_cov_out, uint8_t *host_cov_node_out,
                       uint64_t *host_hist256_out);
/* Device-side variant: leaves the results in device buffers of the same shapes (no D2H). */
int vgmi_counts_finish_device(vgmi_ctx *ctx, uint8_t *dev_cov_out, uint8_t *dev_cov_node_out,
                              uint64_t *dev_hist256_out);
/* Strong-scaling mode for ONE huge sample (not in the reference): the reads are sharded
 * over ranks, every rank counts its shard, then the raw 32-bit counters (key order, not yet clamped)
 * are summed with one all-reduce (RCCL) and written back before vgmi_counts_finish clamps them:
 * min(255, sum over ranks) is exactly the single-GPU result. */
int vgmi_counts_export_device(vgmi_ctx *ctx, uint32_t *dev_counts_out /* n_keys */);
int vgmi_counts_import_device(vgmi_ctx *ctx, const uint32_t *dev_counts /* n_keys */);
/* Accumulated GPU time (HIP events on the context stream) of the read-counting kernel since
 * the last reset, and the number of launches; for roofline accounting. */
int vgmi_count_kernel_ms(vgmi_ctx *ctx, float *ms, uint64_t *launches);

/* ---- device-side FASTQ parsing -----------------------------------------------------------------
 * replaces: kseq_read + the per-record sequence copy of FastqKmer::fastq_file_open (include/kseq.h:192-232,
 *           src/fastq_kmer.cpp:97-105) for REGULAR four-line FASTQ records; the host only moves file text (plain, or
 *           what it inflated) into pinned buffers.  One stream per input file; streams of one context may be driven
 *           from different host threads side by side (everything else of a context stays single-threaded).
 *   open     after vgmi_counts_reset, odd k
 *   acquire  the next pinned staging buffer (waits until the device has taken the previous contents)
 *   commit   n_bytes of file text, continuing where the previous commit ended: copied, parsed and counted
 *            asynchronously
 *   close    waits; n_records / n_bases (= the records' contribution to mReadBase, also added to vgmi_read_base) /
 *            consumed_bytes = how much of the committed text the device took.  stopped == 0: the text ended inside a
 *            record (or cleanly): the unconsumed tail (< 1 MiB) is returned in tail_out for the host reader.
 *            stopped != 0: record number n_records is not a regular four-line record (FASTA, wrapped lines, length
 *            mismatch, empty sequence, '\r' / NUL bytes ...): the host reader must take the stream over at byte
 *            consumed_bytes -- a record boundary, where a fresh kseq state is exactly the reference's state. */
typedef struct vgmi_fastq vgmi_fastq;
int vgmi_fastq_open(vgmi_ctx *ctx, vgmi_fastq **out);
int vgmi_fastq_acquire(vgmi_fastq *fq, char **host_buf, size_t *capacity);
int vgmi_fastq_commit(vgmi_fastq *fq, size_t n_bytes);
/* Text one block-gzip commit may inflate to (>= the staging capacity: the device side is sized to keep every wavefront
 * busy with a member, whatever the staging buffers are). */
int vgmi_fastq_text_capacity(vgmi_fastq *fq, size_t *text_bytes);
/* Compressed bytes the next block-gzip commit would like to see (0: no preference yet): about a whole number of rounds of the
 * inflate kernel's wavefronts, a member each, by the member sizes of the last commit.  A commit takes whole rounds only when
 * it is given more than one; what it leaves comes again (`taken`). */
int vgmi_fastq_bgzf_want(vgmi_fastq *fq, size_t *comp_bytes);
/* Block-gzip (BGZF: bgzip, htslib) input: the staging buffer holds COMPRESSED file bytes, continuing where the previous
 * commit's `taken` ended.  The host walks the member headers; every whole member whose text fits the chunk is inflated
 * on the device (one wavefront per member, CRC-32 and ISIZE checked) into the text the FASTQ kernels parse.
 *   taken     compressed bytes consumed (whole members); the caller puts the rest in front of the next buffer
 *   n_text    text bytes these members inflate to
 *   not_bgzf  the bytes at `taken` are not a block-gzip member (plain gzip member, damage): the device path ends there
 * replaces: gzread's inflate under kseq (include/kseq.h:59-72, src/fastq_kmer.cpp:74-78). */
int vgmi_fastq_commit_bgzf(vgmi_fastq *fq, size_t n_bytes, size_t *taken, size_t *n_text, int *not_bgzf);
/* An ORDINARY gzip member (one DEFLATE stream: what `gzip` writes) inflated on the device, host memory to host memory: block starts are
 * guessed every 32 KiB of compressed bytes, the stretches between them decoded side by side with placeholders for the window each
 * cannot know, checked by having to end exactly where the next one starts, then resolved (vgmi_gunzip.hip; replaces zlib's inflate
 * behind gzread, include/kseq.h:59-72).  n_out = text bytes written, consumed = compressed bytes they came from, member_end = the
 * member's last block was reached; reason = why the device stopped earlier (0: it did not).  Whatever it does not take is the host
 * decoder's.  Test and bench entry of the primitive. */
int vgmi_gunzip_buffer(vgmi_ctx *ctx, const void *host_gz, size_t n, void *host_out, size_t cap, size_t *n_out, size_t *consumed,
                       int *member_end, uint32_t *reason);
/* The same inside a FASTQ stream (vgmi_fastq_open / _acquire, then this instead of _commit): the staged bytes [0, n_bytes) continue an
 * ordinary gzip stream -- at a member header when the stream stands at a member's start, else at the byte that holds the next block's
 * first bit, i.e. what the previous call left untaken.  taken = staged bytes used up (present the rest again in front of what
 * follows); n_text = text inflated, parsed and counted by this call.  stop: 0 go on; 1 the gzip data is over (a member ended and
 * what follows is no member header: gzread ignores it too); 2 the device cannot take these bytes (vgmi_fastq_gzip_status: why) -- the
 * host decoder carries on from the text the device parser has consumed (vgmi_fastq_close's `consumed`).  at_eof: the staged bytes
 * are the file's last. */
int vgmi_fastq_commit_gzip(vgmi_fastq *fq, size_t n_bytes, int at_eof, size_t *taken, size_t *n_text, int *stop);
int vgmi_fastq_gzip_status(vgmi_fastq *fq, uint64_t *device_text_bytes, uint32_t *reason);
/* After the last commit (waits): failed != 0 if a member did not inflate to its ISIZE / CRC-32; good_compressed_bytes =
 * compressed bytes in front of the first such member (all committed bytes if none failed): the text of those bytes went
 * through the parser, the host decoder takes the file over at that offset. */
int vgmi_fastq_bgzf_status(vgmi_fastq *fq, int *failed, uint64_t *good_compressed_bytes, uint32_t *reason);
int vgmi_fastq_close(vgmi_fastq *fq, uint64_t *n_records, uint64_t *n_bases, uint64_t *consumed_bytes, int *stopped,
                     char *tail_out, size_t tail_cap, size_t *tail_len);

/* K1 alone: the key every position of a read block emits, for emitter parity tests.
 * keys_out[i] = key of the k-mer ENDING at byte i, or UINT64_MAX when the reference emits nothing
 * there (kmerBit::kmer_sketch_* loop, src/kmer.cpp:126-146).  Host buffers. */
int vgmi_sketch_keys(vgmi_ctx *ctx, const char *host_bases, size_t n_bytes,
                     const uint64_t *host_read_off, size_t n_reads, uint32_t k,
                     uint64_t *host_keys_out);

/* ---- construct-side counting Bloom filter ------------------------------------------------
 * replaces: BloomFilter(n,p)/add/count/find (include/counting_bloom_filter.hpp:43-77,
 *           src/counting_bloom_filter.cpp:28-98) and BloomFilterKernel::add_kernel
 *           (include/counting_bloom_filter.cuh:56-82), driven by ConstructIndex::make_mbf via
 *           kmerBit::kmer_sketch_bf (src/construct_index.cpp:150-177, src/kmer.cpp:20-53). */
/* m and n_hash as BloomFilter::_calculate_size/_calculate_num_hashes give them */
int vgmi_bloom_params(uint64_t n, double p, uint64_t *m, uint32_t *n_hash);
/* seeds: the 64-bit values the reference stores; only their low 32 bits reach MurmurHash3
 * (`unsigned int seed`, src/counting_bloom_filter.cpp:90) */
int vgmi_bloom_create(vgmi_ctx *ctx, uint64_t m, uint32_t n_hash, const uint64_t *host_seeds);
/* one contiguous sequence (a chromosome); the emitter state spans the whole call */
int vgmi_bloom_add_seq(vgmi_ctx *ctx, const char *host_bases, uint64_t len, uint32_t k);
int vgmi_bloom_add_seq_device(vgmi_ctx *ctx, const char *dev_bases, uint64_t len, uint32_t k);
int vgmi_bloom_fetch(vgmi_ctx *ctx, uint8_t *host_filter_out /* m bytes */);
int vgmi_bloom_load(vgmi_ctx *ctx, const uint8_t *host_filter /* m bytes */);
/* BloomFilter::save / load (src/counting_bloom_filter.cpp:126-190; public, not called by the reference's CLI): the file is
 * u64 size | u32 numHashes | numHashes x u64 seed | size counter bytes. load_file creates the filter from the file. */
int vgmi_bloom_save_file(vgmi_ctx *ctx, const char *path);
int vgmi_bloom_load_file(vgmi_ctx *ctx, const char *path);
/* batch BloomFilter::count (min over hashes) and ::find (all non-zero); outputs may be NULL */
int vgmi_bloom_query(vgmi_ctx *ctx, const uint64_t *host_keys, size_t n, uint8_t *host_min_out,
                     uint8_t *host_all_nonzero_out);

/* ---- forward / backward recursion of the genotyping HMM in the reference's `long double` arithmetic -------------------
 * replaces (inner loops of): GenotypeNameSpace::forward / backward (src/genotype.cpp:1170-1380) for windows whose genotypes
 * all have `ploidy` haplotypes and whose transition is "rec".  A chain is one window walked in one direction; per step the
 * host supplies the node's row of emission scores, the two tables of powers (libm stays on the host: no_recomb^0..ploidy,
 * then recomb^0..ploidy) and whether the chain (re)starts there (the first node, or the node behind one without k-mers).
 * At most 2048 genotypes; beyond 128 the keep matrices must be symmetric (they are by construction: what two genotypes share).
 * Every value is an x86-64 `long double` in its 16-byte memory form; out[step * n_gt + g] is the normalised score the
 * reference stores in HMMScore::a (forward chains) or ::b (backward chains), bit for bit (csrc/vg_x80.h). All pointers host. */
typedef struct vgmi_hmm_chain {
    uint64_t first_step, n_steps;
    uint32_t keep_index, pad;
} vgmi_hmm_chain;
int vgmi_hmm_recursion(vgmi_ctx *ctx, uint32_t n_gt, uint32_t ploidy, const uint8_t *keep, uint32_t n_windows,
                       const void *obs, uint64_t n_rows, const uint32_t *row, const uint8_t *restart, const void *pow,
                       uint64_t n_steps, const void *uniform, const vgmi_hmm_chain *chains, uint32_t n_chains, void *out);
/* The same recursion followed by the posterior of every node (src/genotype.cpp:1387-1522) while alpha and beta are still
 * on the device: per row (node) the host gives the genotype STRING of every entry (gid: the reference keys a std::map by
 * the alleles as decimal strings, sorted as strings) and the strings in string order (order, 0xFF behind the last), and the
 * steps that hold the row's alpha and beta.  Back come the winning string's probability (16-byte long double) and the entry
 * that makes the call (winner; 0xFFFFFFFF: none, as when every posterior is NaN or zero on the host).  alpha_beta_or_null as
 * `out` above when the caller wants them too. */
int vgmi_hmm_calls(vgmi_ctx *ctx, uint32_t n_gt, uint32_t ploidy, const uint8_t *keep, uint32_t n_windows, const void *obs,
                   uint64_t n_rows, const uint32_t *row, const uint8_t *restart, const void *pow, uint64_t n_steps,
                   const void *uniform, const vgmi_hmm_chain *chains, uint32_t n_chains, const uint8_t *gid, const uint8_t *order,
                   const uint64_t *fwd_step, const uint64_t *bwd_step, void *prob, uint32_t *winner, void *alpha_beta_or_null);
/* The same on a PART of a run's arrays: every array is the whole run's, indexed by global row / step (row[], fwd_step[],
 * bwd_step[] and the chains hold global numbers); the call reads rows [row_lo, row_hi) and steps [step_lo, step_hi) and writes
 * those rows of prob / winner.  keep holds the matrices of this part's chains (keep_index counts from 0 here).  Each call
 * works on a stream and device buffers of its own: parts may be computed at the same time from several threads on one
 * context -- the windows of a sample become ready one after the other, and a chain is serial from end to end, so the host
 * prepares the next part while the device runs the last (varigraph_amd/csrc/host/genotyper.cpp). */
int vgmi_hmm_calls_part(vgmi_ctx *ctx, uint32_t n_gt, uint32_t ploidy, const uint8_t *keep, uint32_t n_windows, const void *obs,
                        uint64_t row_lo, uint64_t row_hi, const uint32_t *row, const uint8_t *restart, const void *pow,
                        uint64_t step_lo, uint64_t step_hi, const void *uniform, const vgmi_hmm_chain *chains, uint32_t n_chains,
                        const uint8_t *gid, const uint8_t *order, const uint64_t *fwd_step, const uint64_t *bwd_step, void *prob,
                        uint32_t *winner);

/* ---- emission scores of the HMM's nodes on the device ---------------------------------------------------------------------
 * replaces: GenotypeNameSpace::hidden_states + observable_states (src/genotype.cpp:640-830, 960-1000, most_likely_depth :1118-1145)
 * for a DIPLOID sample whose windows all use the same genotype list -- every haplotype of the graph is selected (-n >= haplotypes),
 * so no k-mer list is pruned and nothing is drawn per window.  Everything is in NODE ORDER (entry j = place j of the graph2node
 * lists, the order of vgmi_counts_finish's cov_node):
 *   entries_upload   once per graph: per entry  multiplicity << 8 | haplotype bits << 16  (f and BitVec of its k-mer)
 *   sample_upload    per sample: the coverage of every entry (cov_node) 
 *   emissions        per group of windows ("part"): per row (node) its entries [entry_begin, +entry_count), the bit mask of the
 *                    genotypes' haplotypes that carry the reference allele there (gt0, bit p = used[p]); genotype g is the pair
 *                    (used[pos_a[g]], used[pos_b[g]]); tables = the sample's 256 geometric terms then 256 Poisson terms for one
 *                    and for two copies (16-byte long doubles; libm stays on the host).  Back come, per row, the number of k-mers
 *                    that took part (0: the node has no score) and flags (bit 0: a haplotype's sequence has to be checked --
 *                    the host must score this node itself and hand its row in with part_set_rows; bit 1: a k-mer no selected
 *                    haplotype carries -- the caller's guarantee does not hold, do not use the part).  The scores stay on the
 *                    device inside the part.
 *   part_calls       vgmi_hmm_calls on the part's rows (row / step numbers count from 0 inside the part)
 * A context may hold several parts; calls on different parts may run side by side (own streams). */
typedef struct vgmi_hmm_part vgmi_hmm_part;
int vgmi_hmm_entries_upload(vgmi_ctx *ctx, const uint64_t *host_entries, size_t n_entries);
int vgmi_hmm_sample_upload(vgmi_ctx *ctx, const uint8_t *host_cov_node, size_t n_entries);
int vgmi_hmm_emissions(vgmi_ctx *ctx, uint32_t n_gt, uint32_t n_used, const uint8_t *used, const uint8_t *pos_a, const uint8_t *pos_b,
                       uint64_t top_mask, uint32_t bit_len, float ave, double lower, double upper, const void *tables,
                       uint64_t n_rows, const uint64_t *entry_begin, const uint32_t *entry_count, const uint16_t *gt0,
                       uint32_t *n_kept_out, uint8_t *flags_out, vgmi_hmm_part **out);
/* ... the same for genotypes of `ploidy` haplotypes, 2 .. 4 (a polyploid sample's genotypes are blocks of consecutive haplotypes,
 * src/genotype.cpp:846-873): pos[g * ploidy + q] = the place in `used` of genotype g's q-th haplotype; tables holds (ploidy + 1) x 256
 * terms (geometric for h = 0, Poisson(ave * h) for h = 1 .. ploidy).  vgmi_hmm_emissions is this with ploidy 2. */
int vgmi_hmm_emissions_ploidy(vgmi_ctx *ctx, uint32_t n_gt, uint32_t ploidy, uint32_t n_used, const uint8_t *used, const uint8_t *pos,
                              uint64_t top_mask, uint32_t bit_len, float ave, double lower, double upper, const void *tables, uint64_t n_rows,
                              const uint64_t *entry_begin, const uint32_t *entry_count, const uint16_t *gt0, uint32_t *n_kept_out,
                              uint8_t *flags_out, vgmi_hmm_part **out);
int vgmi_hmm_part_set_rows(vgmi_hmm_part *part, uint64_t n, const uint64_t *rows, const void *obs_rows /* n x n_gt long doubles */);
/* ... or, for the rows flagged with bit 0, leave the products on the device (round 5): the host consults the haplotypes' sequences
 * (strings: src/genotype.cpp:760-800) and says which entries of a row lose which haplotypes -- entry fix_j[i] of row rows[r]
 * (fix_off[r] <= i < fix_off[r + 1], ascending in fix_j) loses the haplotypes of fix_mask[i] (bits over the `used` list) -- and the
 * rows are scored again by the emission kernel with those bits cleared. */
int vgmi_hmm_part_fix_rows(vgmi_hmm_part *part, uint64_t n, const uint64_t *rows, const uint32_t *fix_off, const uint32_t *fix_j,
                           const uint16_t *fix_mask);
int vgmi_hmm_part_calls(vgmi_hmm_part *part, uint32_t ploidy, const uint8_t *keep, uint32_t n_windows, const uint32_t *row,
                        const uint8_t *restart, const void *pow, uint64_t n_steps, const void *uniform, const vgmi_hmm_chain *chains,
                        uint32_t n_chains, const uint8_t *gid, const uint8_t *order, const uint64_t *fwd_step, const uint64_t *bwd_step,
                        void *prob, uint32_t *winner);
/* The calls' k-mer tallies (src/genotype.cpp:1387-1414, read for the CALLED haplotypes): per row with winner[i] < n_gt, genotype
 * winner[i] = the haplotypes (hap_ab[2 g], hap_ab[2 g + 1]); out[4 i ..] = k-mers of the node haplotype a carries, the sum of
 * their coverages, the same for b (a haplotype >= n_hap or outside sel_mask: 0, 0); unique_out[i] = k-mers of multiplicity <= 1,
 * at most 255.  Uses the entries and the sample's coverage uploaded for the emissions. */
/* A part's recursion inputs kept on the device (round 5): everything vgmi_hmm_part_calls uploads -- keep matrix, step tables, rows,
 * restarts, chains, genotype strings' ids and order, the rows' steps -- is a function of the graph and the options, not of the sample.
 * A plan holds it (plain device memory: any context of the device may use it, and it outlives them), made once by
 * vgmi_hmm_plan_create with the arguments of vgmi_hmm_part_calls; vgmi_hmm_part_calls_plan then runs a part's recursion and
 * posterior on the plan's inputs and the part's own emission scores. */
typedef struct vgmi_hmm_plan vgmi_hmm_plan;
int vgmi_hmm_plan_create(vgmi_ctx *ctx, uint32_t n_gt, uint32_t ploidy, const uint8_t *keep, uint32_t n_windows, uint64_t n_rows,
                         const uint32_t *row, const uint8_t *restart, const void *pow, uint64_t n_steps, const void *uniform,
                         const vgmi_hmm_chain *chains, uint32_t n_chains, const uint8_t *gid, const uint8_t *order, const uint64_t *fwd_step,
                         const uint64_t *bwd_step, vgmi_hmm_plan **out);
int vgmi_hmm_part_calls_plan(vgmi_hmm_part *part, const vgmi_hmm_plan *plan, void *prob, uint32_t *winner);
void vgmi_hmm_plan_free(vgmi_hmm_plan *plan);
int vgmi_hmm_tallies(vgmi_ctx *ctx, uint64_t n_rows, const uint64_t *entry_begin, const uint32_t *entry_count, const uint32_t *winner,
                     uint32_t n_gt, const uint8_t *hap_ab, uint32_t n_hap, uint64_t sel_mask, uint32_t *out, uint8_t *unique_out);
/* the part's emission rows back on the host (n_rows x n_gt long doubles): tests and diagnostics */
int vgmi_hmm_part_fetch(vgmi_hmm_part *part, void *obs_out);
void vgmi_hmm_part_free(vgmi_hmm_part *part);

/* (bench / test tooling -- the seeded synthetic workloads -- lives in libvgsynth.so, varigraph_amd/csrc/bench/vgsynth.h: not part of this ABI) */

#ifdef __cplusplus
}
#endif
#endif /* VGMI_H */
