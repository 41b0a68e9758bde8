// vgmi_kernels.h -- host-visible launch interface of vgmi_kernels.hip
#ifndef VGMI_KERNELS_H
#define VGMI_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgmi_ctable.h"
#include "vgmi_device.h"

// VGMI_DBG ablations change the arithmetic (wrong counters / genotypes): they exist only in a library built with
// -DVGMI_ABLATION (VGMI_ABLATION=1 python -m varigraph_amd.build --force; tools/pmc_quick.sh does).  A product build reads no
// debug knob at all, and vgmi_dbg_env() says so loudly if the variable is set.
#ifdef VGMI_ABLATION
#define VG_DBG(x) (x)
#else
#define VG_DBG(x) 0u
#endif
uint32_t vgmi_dbg_env();

namespace vgk {

enum { K_MODE_COUNT = 0, K_MODE_KEYS = 1, K_MODE_BLOOM = 2, K_MODE_DEBIT = 3 };

// table keyed by the read's grid 16-mer (vgmi_xtable.hip): lines of 16 entries, dense counters by id
#define XT_HOPS 3u          // a k-mer sits in its home line or one of the XT_HOPS lines behind it (the table ends in XT_HOPS lines of slack)
struct XTableView {
    unsigned long long* lines;   // 16 * (n_lines + XT_HOPS) entries, or nullptr: not in use
    uint32_t n_lines;            // home lines, any number (not a power of two): line = (h(X) * n_lines) >> 32
    uint32_t tag_bits;           // low bits of h(X) kept in the entry: 2^tag_bits >= the h-values XT_HOPS + 1 consecutive lines
                                 // cover, so an entry's (line it was found in, tag) is X
    uint32_t id_shift;           // 26 + tag_bits: entry = j' | f << 4 | tag << 26 | id << id_shift
    uint32_t* counts;            // n_keys
    const ulonglong2* over;      // {canonical k-mer | XT_EMPTY, id} of the k-mers some 16-mer of which found no room
                                 // (repeats), or nullptr: none
    uint32_t over_mask;          // its capacity - 1 (a power of two)
    // the context table (vgmi_ctable.h; round 4, the default for these graphs): when cb is set, lines is not -- counts, the
    // path-ordered ids and the exact overflow table are shared by both forms
    const uint4* cb;             // 4 * (n_buckets + CT_HOPS) entries of 16 bytes, or nullptr: not in use
    uint32_t n_buckets;          // home buckets of 64 bytes, any number: bucket = (ct_hash(X) * n_buckets) >> 32
    uint32_t k;                  // 27, or 19 .. 25 (context table only; round 5): flanks of k - 16 bases
};

// path table of small graphs (vgmi_ptable.hip, build_ptable in vgmi_api_table.cpp): what count27s_kernel<true> checks candidate runs against
struct PathView {
    const unsigned long long* index;   // 2 << bucket_log2 entries {12-mer : 24, place a : 19, place b : 19}, or nullptr: not in use
    const uint32_t* S;                 // the graph's unitigs, both orientations, 2 bits per base, 16 bases per word (first base most significant)
    const uint32_t* VB;                // bit per base position: a graph k-mer starts here
    uint32_t* SB;                      // bit per base position: ... and its counter has reached the clamp (per sample)
    const uint32_t* SLOT;              // per base position: the hash-table slot of the k-mer that starts here
    const uint32_t* PLACE;             // per hash-table slot: the place (first half of S) where its k-mer starts, 0: none -- for the slow paths' saturation bits
    uint32_t bucket_log2;
    uint32_t Tp;                       // bases in S, pads included: S[Tp - 1 - j] is the complement of S[j]
};

struct TableView {
    VgSlot* slots;              // cap entries (16-byte format) or nullptr
    unsigned long long* slots8; // cap k-mer words (compact 8-byte format) or nullptr
    uint64_t cap_mask;          // cap - 1 (cap is a power of two)
    uint32_t k;                 // the k-mer length of the keys
    uint32_t home_bucket_log2;  // 0: home slot = vg_thash; else minimiser buckets of 1 << this slots (vg_thash_local)
    uint32_t home_by_offset;    // place inside the bucket from the minimiser's offset (neighbouring k-mers -> neighbouring slots)
    const uint32_t* filter;     // blocked-Bloom prefilter, 1 << filter_words_log2 words
    uint32_t filter_words_log2; // >= 2
    uint32_t filter_shift;      // 32 - filter_words_log2
    const uint32_t* grid;       // grid filter (1 << grid_words_log2 words) or nullptr, see vgmi_device.h
    uint32_t grid_words_log2;   // VG_GRID_LDS_WORDS_LOG2 (LDS-resident variant) or larger (global variant)
    uint8_t* sat_dirty;         // compact format: one byte per 2048-slot region, set when a slot of the region gets its
                                // saturation flag; the per-sample reset sweeps only those regions
    XTableView xt;              // when xt.lines is set, every count kernel looks k-mers up there and counts in xt.counts
    PathView pt;                // small graphs: count27s_kernel's run lookups (counters stay per slot)
    uint32_t* counts;           // 16-byte format: dense per-key counters of large graphs (4 B/key, Infinity-Cache
                                // sized) or nullptr (in-slot counters); compact format: per-slot counters (cap)
};

#define VG_BLOOM_MAX_HASH 32
struct BloomView {
    uint8_t* filter;  // m bytes (+ padding to a multiple of 4)
    uint64_t m;
    uint64_t magic;   // floor((2^64-1)/m)
    uint32_t n_hash;
    uint32_t seeds[VG_BLOOM_MAX_HASH];  // low 32 bits of the reference's stored seeds
};

struct RowParams {
    const uint8_t* bases;
    uint64_t n_bytes;
    uint64_t row_begin;   // rows_kernel: first 1 KiB row to process (earlier rows only provide the halo)
    uint64_t row_end;     // count27_kernel: one past the last row; every row below it is complete
    uint64_t emit_from;   // rows_kernel<COUNT>: only k-mers ENDING at stream position >= emit_from are counted (the
                          // fast kernel covers the ends up to its last grid offset)
    uint32_t k;
    uint32_t dbg;         // ablations of count27_kernel only (VGMI_DBG; counts are wrong with any of them): 1 = scan
                          // only (candidate runs dropped), 32 = one probing lane per run, 64 = collisions dropped; 128 = K3 without the per-wave
                          // privatisation (same filter, A/B only)
    uint32_t* status;     // bit0 empty read, bit1 bad key, bit2 duplicate key
    const unsigned long long* n_bytes_dev;   // count kernels: when set, the block's length is read from device memory (the
                          // device-side FASTQ parser knows it, the host does not) and n_bytes / row_end / row_begin /
                          // emit_from are derived from it in the kernel; tail27 tells rows_kernel it runs behind count27_kernel
    uint32_t tail27;      // 1 behind count27_kernel, 2 behind count27x_kernel, 3 behind count27s_kernel, 4 behind count27s_kernel<true, K < 27> (it covers every end inside its rows)
    uint32_t l1_min;      // count27s_kernel<true>: queued runs that start a round of the path-table look-up (VGMI_L1_MIN; default 60, at most 64: one lane per run)
    TableView table;      // MODE_COUNT
    uint64_t* keys_out;   // MODE_KEYS
    BloomView bloom;      // MODE_BLOOM
};

// device-side FASTQ parsing (vgmi_fastq.hip)
struct FqState {                       // device-resident, one per open stream
    unsigned long long n_records;      // accepted since the stream was opened
    unsigned long long n_bases;        // their sequence bytes (mReadBase)
    unsigned long long consumed;       // file text bytes taken by the device: the host reader resumes here
    unsigned long long packed_bytes;   // read block of the chunk just parsed (the count kernels' n_bytes)
    uint32_t stopped;                  // sticky: the device parser has handed the rest of the stream to the host reader
    uint32_t tail_len;                 // bytes of an incomplete record carried into the next chunk
    uint32_t start;                    // where this chunk's text begins in the raw buffer (tail_max - previous tail_len)
    uint32_t consumed_end;             // end of the last accepted record of this chunk
    uint32_t n_lines, n_good, first_bad, dirty;   // per chunk
};
struct FqBuffers {
    const uint8_t* raw;                // tail_max bytes of carry area + the chunk
    uint8_t* raw_next;                 // the other raw buffer (receives the tail)
    uint8_t* packed;                   // '\n'-joined sequences
    uint32_t* tile;                    // newline count / base per 4 KiB tile
    uint32_t* nlpos;                   // cap_lines newline positions
    uint32_t* rec_bytes;               // per record: sequence length + 1
    uint32_t* out_off;                 // per record: offset in the packed block
    uint32_t* block_sum;               // scan scratch
    FqState* state;
    uint32_t cap_lines, tail_max;
};
hipError_t launch_fastq_init(FqState* st, uint32_t tail_max, hipStream_t s);
// n_new_dev (may be null): the chunk is cut to min(n_new, *n_new_dev) bytes on the device (text in front of a block-gzip
// member that did not inflate)
hipError_t launch_fastq_chunk(const FqBuffers& b, uint32_t n_new, hipStream_t s, const uint32_t* n_new_dev = nullptr);

// block-gzip members inflated on the device (vgmi_inflate.hip)
struct BgzfMember {
    uint32_t c_off, c_len;   // DEFLATE bytes of the member inside the compressed batch
    uint32_t u_off, u_len;   // where its text goes inside the chunk, ISIZE
    uint32_t crc, pad;       // CRC-32 from the trailer
};
struct BgzfVerdict {         // device-resident, per stream
    uint32_t first_bad_batch, first_bad_member, reason;   // 0xFFFFFFFF while every member has inflated and checked
    uint32_t good_bytes;     // of the batch just inflated: text in front of the first bad member
    uint32_t batches;
};
// K3 with the updates binned by filter chunk (vgmi_bloom_bin.hip)
struct BloomBinPlan {
    int ok;                          // 0: this filter / call is not for the binned form
    uint32_t n_chunks, n_bins, n_sub, bin_shift, cap1, cap2;
    size_t scratch_bytes;
};
BloomBinPlan bloom_bin_plan(uint64_t m, uint32_t n_hash, uint64_t n_keys_max);
hipError_t launch_bloom_binned(const BloomView& b, const uint64_t* keys, uint64_t n_keys, const BloomBinPlan& plan, uint8_t* scratch, int n_cu, hipStream_t st,
                               int* overflowed);
uint32_t bgzf_wave_slots(int n_cu);
hipError_t launch_bgzf_inflate(const uint8_t* comp, const BgzfMember* members, uint32_t n_members, uint8_t* out_base, uint32_t* status,
                               const uint32_t* crc_table, BgzfVerdict* verdict, hipStream_t s);

// ordinary gzip streams inflated on the device (vgmi_gunzip.hip): block starts guessed per 32 KiB of compressed bytes, the stretches
// between them decoded into symbols (bytes and placeholders for the window in front), windows propagated, symbols resolved
struct GzSegHost {
    uint32_t start_bit, stop_bit, sym_off, sym_cap, win_avail, pad;
};
struct GzSegOutHost {
    uint32_t n_sym, end_bit, status, final_block;
};
hipError_t launch_gz_find(const uint8_t* comp, uint32_t n_bytes, uint32_t sub_bytes, uint32_t n_sub, uint32_t per, uint32_t* starts, hipStream_t st);
hipError_t launch_gz_decode(const uint8_t* comp, uint32_t n_bytes, const void* segs, uint32_t n_seg, uint16_t* pool, void* outs, hipStream_t st);
hipError_t launch_gz_resolve(const uint16_t* pool, const void* segs, const void* outs, const uint64_t* text_off, uint32_t n_seg, uint16_t* w1, uint8_t* t,
                             uint8_t* text, hipStream_t st);
uint32_t gz_groups(uint32_t n_seg);
// CRC-32 / ISIZE of a member's text as the pieces are resolved: the running remainder (linear form) and length, device-resident
struct GzCrcState {
    uint32_t r, pad;
    uint64_t len;
};
hipError_t launch_gz_crc(const uint8_t* text, uint64_t n, uint32_t* chunk_r, GzCrcState* state, hipStream_t st);
size_t gz_crc_chunks(size_t n_text);
uint32_t gz_crc_finish(uint32_t r, uint64_t len);

// ---- HMM recursion (vgmi_hmm.hip) ----
struct HmmChain {
    uint64_t first_step, n_steps;   // its steps in the step arrays, in the order they are taken
    uint32_t keep_index;            // which keep matrix (one per window)
    uint32_t pad;
};
#define VGMI_HMM_MAX_GT 2048u        // genotypes per window the device takes (12 x n x (ploidy + 2) bytes of LDS: 147 KB at the bound)
struct HmmParams {
    uint32_t n_gt, ploidy;          // genotypes per window (<= VGMI_HMM_MAX_GT; <= 128: one lane per genotype, keep matrix in LDS), haplotypes per genotype (<= 4)
    const uint8_t* keep;            // per window: n_gt x n_gt, haplotypes shared by genotype g (row) and previous entry p
    const uint8_t* obs;             // per node: n_gt emission scores, 16 bytes each (x86-64 long double)
    const uint32_t* row;            // per step: the node's row in obs
    const uint8_t* restart;         // per step: the chain begins (again) here: r = obs
    const uint8_t* pow;             // per step: no_recomb^0..ploidy, then recomb^0..ploidy, 16 bytes each
    const uint8_t* uniform;         // 1 / n_gt as the host computes it
    const HmmChain* chains;
    uint8_t* out;                   // per step: n_gt normalised scores
    uint32_t dbg;                   // VGMI_DBG ablations (wrong results): 1 no sum over entries, 2 no terms, 4 no division; 8 no skipped terms (right results)
};
hipError_t launch_hmm_recursion(const HmmParams& P, uint32_t n_chains, hipStream_t st);
struct HmmPostParams {
    uint32_t n_gt;
    uint64_t row0;                  // the launch's first row (workgroup b takes row0 + b)
    const uint8_t* ab;              // the recursion's output: per step n_gt scores
    const uint64_t* fwd_step;       // per row: the step that holds its alpha / its beta
    const uint64_t* bwd_step;
    const uint8_t* gid;             // per row: genotype string of every entry
    const uint8_t* order;           // per row: the strings in string order, 0xFF behind the last
    uint8_t* prob;                  // per row: the winning string's probability (16 bytes)
    uint32_t* winner;               // per row: the entry that makes the call; 0xFFFFFFFF: none
};
hipError_t launch_hmm_posterior(const HmmPostParams& P, uint64_t n_rows, hipStream_t st);
// emission scores of a window's nodes on the device (hidden states + observable states of src/genotype.cpp:640-830, 960-1000) for the
// case every k-mer list is whole and every genotype is a pair of haplotypes (vgmi_hmm_emissions in vgmi.h)
struct HmmEmitParams {
    const unsigned long long* packed;   // per node-list entry: multiplicity << 8 | haplotype bits << 16 (the low byte is not used)
    const uint8_t* cov;                 // per node-list entry: this sample's coverage
    const uint64_t* entry_begin;        // per row: its node's entries
    const uint32_t* entry_count;
    const uint16_t* gt0;                // per row: bit p = haplotype used[p] carries the reference allele at this node
    uint64_t row_lo;                    // the launch's first row (workgroup b takes row_lo + b)
    uint32_t n_gt, n_used, bl8;         // genotypes (<= 128), haplotypes in them (<= 16), bit length of a k-mer's haplotype bits (8 * bitlen)
    uint8_t used[16], pos_a[128], pos_b[128];
    uint8_t pos_more[2][128];           // genotypes of three or four haplotypes (ploidy): the third and the fourth one's places in `used`
    uint32_t ploidy;                    // 2 .. 4: haplotypes per genotype
    unsigned long long top_mask;
    float ave;
    double lower, upper;
    const uint8_t* tables;              // 256 geometric terms (h = 0), then 256 Poisson terms per h = 1 .. ploidy: 16-byte long doubles
    uint8_t* obs;                       // out, per row: n_gt scores
    uint32_t* n_kept;                   // out, per row: k-mers that took part
    uint8_t* flags;                     // out, per row: bit 0 the host must score this node (a haplotype's sequence has to be checked), bit 1 a k-mer no selected haplotype carries
    // the second launch over the flagged rows (round 5): what the haplotypes' SEQUENCES said (src/genotype.cpp:760-800).  Workgroup b
    // scores row fix_rows[b] again; entry fix_j of the row loses the haplotypes of fix_mask (bits over `used`): the host found that
    // their sequence does not hold this under-covered multi-copy k-mer.  Null in the first launch.
    const uint64_t* fix_rows;
    const uint32_t* fix_off;            // per fixed row: its stretch of (fix_j, fix_mask), ascending in fix_j
    const uint32_t* fix_j;              // 32 bits like entry_count and the kernel's j
    const uint16_t* fix_mask;
};
hipError_t launch_hmm_emissions(const HmmEmitParams& P, uint64_t n_rows, hipStream_t st);
hipError_t launch_hmm_scatter_rows(uint8_t* obs, const uint64_t* rows, const uint8_t* src, uint32_t n_gt, uint64_t n, hipStream_t st);
size_t hmm_lds_bytes(uint32_t n_gt, uint32_t ploidy);
hipError_t launch_hmm_tally(const unsigned long long* packed, const uint8_t* cov, const uint64_t* entry_begin, const uint32_t* entry_count, const uint32_t* winner,
                            const uint8_t* hap_ab, uint32_t n_gt, uint32_t n_hap, unsigned long long sel_mask, uint64_t n_rows, uint32_t* out, uint8_t* uniq,
                            hipStream_t st);

hipError_t launch_xtable_build(const XTableView& t, const unsigned long long* slots8, const uint32_t* key_slot, const uint32_t* id_of_key,
                               uint64_t n_keys, uint32_t* over_list, uint32_t over_cap, unsigned long long* over_n, hipStream_t st);
hipError_t launch_xtable_over(ulonglong2* over, uint32_t over_mask, const unsigned long long* slots8, const uint32_t* key_slot,
                              const uint32_t* id_of_key, const uint32_t* over_list, uint64_t n_over, hipStream_t st);
hipError_t launch_xtable_number(const XTableView& t, const unsigned long long* slots8, const uint32_t* key_slot, uint64_t n, uint32_t* link,
                                uint32_t* link2, uint32_t* id_of_key, unsigned long long* cursor, uint32_t* mark, uint32_t* status,
                                hipStream_t st);
hipError_t launch_count27x(const RowParams& p, const XTableView& t, uint32_t grid, hipStream_t st);
hipError_t launch_ctable_okmer(const TableView& t, const uint32_t* key_slot, uint32_t* pos_of_key, const uint32_t* link2, uint64_t n, bool identity,
                               unsigned long long* okmer, uint32_t* id_of_key, unsigned long long* n_unitigs, hipStream_t st);
hipError_t launch_ctable_build(const XTableView& t, const unsigned long long* okmer, uint64_t n_places, uint32_t* over_list, uint32_t over_cap,
                               unsigned long long* over_n, unsigned long long* n_moved, hipStream_t st);
hipError_t launch_ctable_over(ulonglong2* over, uint32_t over_mask, const unsigned long long* okmer, const uint32_t* over_list, uint64_t n_over,
                              uint32_t k, hipStream_t st);
// Deferred counter updates of the context-table kernels (round 6, vgmi_ctdefer.hip): the count kernel writes its runs of hits {id0, windows | dir
// << 12} to `rec` in chunks of CTD_CHUNK records a wavefront reserves (`cursor`, records reserved so far) instead of issuing the atomics in
// its row loop; ctd_scatter_kernel partitions them by 32 768-counter region (`bin_cursor`, `binned`: n_bins rooms of `room` 4-byte
// records), ctd_accumulate_kernel adds a region's runs up in LDS and hands the sums to the counters.  rec == nullptr: the plain kernel.
#define CTD_CHUNK 256u
#ifndef VGMI_CT_DEFER_DEFAULT
#define VGMI_CT_DEFER_DEFAULT 1      // (round 6 A/B, chr20 class, nine pairs of processes on three boxes: 7.68-7.78 ms deferred, 7.79-8.37 ms in the row loop)
#endif
#define CTD_MAX_BINS 2048u
struct CtDefer {
    uint2* rec;
    uint32_t cap;                 // records `rec` holds, a multiple of CTD_CHUNK
    unsigned int* cursor;         // records reserved (may pass cap: a chunk that does not fit is not written, its runs leave as atomics)
    uint32_t n_bins, room;
    unsigned int* bin_cursor;     // [n_bins * n_wg]: records in the room of (bin, workgroup of the scatter kernel)
    uint32_t* binned;             // [n_bins * n_wg * room]
    uint64_t n_counts;
    uint32_t region, inv;         // counters a region, 2^32 / region rounded up
    uint32_t n_wg;                // workgroups of the scatter kernel = rooms a bin
};
hipError_t launch_count27c(const RowParams& p, const XTableView& t, uint32_t n_cu, hipStream_t st, const CtDefer* defer = nullptr);
size_t ctd_scratch_bytes(uint64_t n_bytes, uint64_t n_counts, uint32_t n_cu, CtDefer* layout);      // 0: the table has too many counters for one level of bins
void ctd_layout(uint8_t* scratch, CtDefer* d);
hipError_t launch_ctd_reset(const CtDefer& d, hipStream_t st);
hipError_t launch_ctd_apply(const XTableView& t, const CtDefer& d, uint32_t n_cu, hipStream_t st);
hipError_t launch_xclamp(const XTableView& t, uint64_t n, hipStream_t st);
hipError_t launch_xcov(const XTableView& t, const uint32_t* id_of_key, uint64_t n, const uint8_t* flag, uint8_t* cov, unsigned long long* hist,
                       hipStream_t st);
hipError_t launch_xcounts_xfer(const XTableView& t, const uint32_t* id_of_key, uint32_t* ext, uint64_t n, bool import, hipStream_t st);
hipError_t launch_count27(bool lds_bitmap, const RowParams& p, uint32_t grid, uint32_t block, hipStream_t st);
hipError_t launch_count27s(const RowParams& p, uint32_t grid, hipStream_t st);   // small graphs: 12-mer grid, 1 024-byte rows
hipError_t launch_ptable_order(const TableView& t, const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot, uint32_t* link, uint32_t* link2,
                               uint32_t* pos_of_key, unsigned long long* cursor, uint32_t* mark, uint32_t* status, hipStream_t st, uint32_t align = 1);
hipError_t launch_ptable_check(const uint32_t* pos_of_key, uint64_t n, uint64_t total, uint32_t* mark, uint32_t* status, hipStream_t st);
hipError_t launch_ptable_fill(const TableView& t, const uint32_t* key_slot, const uint32_t* pos_of_key, uint64_t n, ulonglong2* P, hipStream_t st);
hipError_t launch_rows(int mode, bool flds, const RowParams& p, uint32_t grid, uint32_t block, hipStream_t st);
hipError_t launch_bloom_even(const RowParams& p, hipStream_t st);   // even k, one long sequence (K3)
hipError_t launch_seq(int mode, const RowParams& p, const uint64_t* read_off, uint64_t n_reads, hipStream_t st);
#define VG_DEBIT_LIST (1u << 20)      // positions of non-bases one launch can hand to its walk launch (8 MiB per stream; the rest is walked in the scan)
#define VG_DEBIT_SUBLISTS 256u       // ... in that many lists of equal size, one counter each (64 bytes apart, behind the positions)
hipError_t launch_even_debit(const RowParams& p, const uint64_t* read_off, uint64_t n_reads, unsigned long long* list, uint32_t list_cap, hipStream_t st);
hipError_t launch_table_clear(const TableView& t, hipStream_t st);
hipError_t launch_table_insert(const TableView& t, const uint64_t* keys, uint64_t n, uint32_t k, uint32_t* key_slot,
                               uint32_t* filter_rw, uint32_t* grid_rw, bool grid12, uint32_t* status, hipStream_t st);
hipError_t launch_table_key_of_slot(const uint32_t* key_slot, uint64_t n, uint32_t* key_of_slot, hipStream_t st);
hipError_t launch_table_lookup(const TableView& t, const uint64_t* keys, uint64_t n, uint32_t k, const uint32_t* key_of_slot, uint32_t* out,
                               hipStream_t st);
hipError_t launch_counts_reset(const TableView& t, hipStream_t st);
hipError_t launch_cov(const TableView& t, const uint32_t* key_slot, uint64_t n, const uint8_t* flag,
                      uint8_t* cov, unsigned long long* hist, hipStream_t st);
hipError_t launch_counts_xfer(const TableView& t, const uint32_t* key_slot, uint32_t* ext, uint64_t n, bool import,
                              hipStream_t st);
hipError_t launch_node_gather(const uint8_t* cov, const uint32_t* key_index, uint64_t n, uint8_t* cov_node, hipStream_t st);
hipError_t launch_bloom_query(const BloomView& b, const uint64_t* keys, uint64_t n, uint8_t* min_out, uint8_t* nz_out,
                              hipStream_t st);

}  // namespace vgk
#endif
