// vgmi_ctx.h -- what the files of the C ABI (vgmi_api*.cpp; include/vgmi.h) share: the context, the table image's header, the error
// macro and the helpers one file defines for the others (namespace vgapi).  Internal: nothing here is part of the ABI.
//
//   vgmi_api.cpp         contexts, read counting (launch_count: which kernel serves which graph), read-out, timing
//   vgmi_api_table.cpp   the table image: layout, upload / import / export / clone, the path / context / grid-16-mer tables built from it
//   vgmi_api_rccl.cpp    the image over RCCL (one process per GPU)
//   vgmi_api_fastq.cpp   FASTQ text on the device: records, block-gzip and gzip members
//   vgmi_api_bloom.cpp   the construct side's counting Bloom filter
//   vgmi_api_hmm.cpp     the HMM's emissions, recursion, posterior and tallies
#ifndef VGMI_CTX_H
#define VGMI_CTX_H
#include "../../include/vgmi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vgmi_device.h"
#include "vgmi_kernels.h"

using namespace vgk;

struct ImageHeader {  // first 256 bytes of the table image
    char magic[8];    // "VGMITBL1"
    uint32_t k;
    uint32_t filter_words_log2;
    uint64_t n_keys;
    uint64_t cap;
    uint64_t off_slots, off_key_slot, off_filter, off_grid, total_bytes;
    uint32_t grid_words_log2;
    uint32_t slot_bytes;   // 16: VgSlot, 8: compact k-mer words (vgmi_device.h)
    uint32_t home_bucket_log2;   // 0: vg_thash home slots, else minimiser buckets (vg_thash_local)
    uint32_t home_by_offset;     // place inside the bucket = minimiser offset (vgmi_device.h)
    uint32_t grid_mer;           // 16: grid filter over 16-mers (step 12); 12: over 12-mers (step 16; small graphs, count27s_kernel)
    uint8_t pad[256 - 8 - 4 - 4 - 8 - 8 - 40 - 8 - 4 - 4 - 4];
};
static_assert(sizeof(ImageHeader) == 256, "image header is 256 bytes");

struct Stage {
    char* h = nullptr;        // pinned
    char* d = nullptr;
    uint64_t* d_off = nullptr;
    size_t d_off_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // last kernel that used this stage
    bool busy = false;
    bool after_reset = false;   // the next launch on this stage's stream must wait for the per-sample reset (main stream)
};

struct ncclUniqueIdBytes { char internal[128]; };      // rccl.h: ncclUniqueId (passed by value to ncclCommInitRank)

struct vgmi_ctx {
    uint64_t id = 0;               // never repeats in a process: what a thread's own last error is matched against (an address may come back)
    int device = 0;
    int n_cu = 0;
    size_t buffer_bytes = 0;
    std::string err;
    hipStream_t stream = nullptr;  // main stream: table build, device submits, finish
    // working memory of the HMM calls, kept between them: hipFree waits for every stream of the device, so a part that
    // finished would wait for the parts still running (vgmi_hmm_calls_part); a sample reuses the last sample's blocks
    unsigned long long* d_hmm_entries = nullptr;     // vgmi_hmm_entries_upload: per node-list entry f << 8 | haplotype bits << 16
    uint8_t* d_hmm_cov = nullptr;                    // vgmi_hmm_sample_upload: this sample's coverage per entry
    size_t hmm_n_entries = 0;
    std::mutex hmm_mu;
    std::vector<std::pair<uint8_t*, size_t>> hmm_blocks;   // not in use

    // table image (one allocation) and views into it
    uint8_t* d_image = nullptr;
    uint8_t* d_snapshot = nullptr;      // vgmi_table_snapshot: the image as uploaded, for a broadcast that leaves after counting has begun
    size_t image_bytes = 0;
    ImageHeader hdr{};
    bool has_table = false;
    TableView tv{};
    uint32_t* d_key_slot = nullptr;
    uint64_t xt_bytes_since_clamp = 0;
    uint64_t xt_n_counts = 0;                   // counters of the grid-16-mer / context table: n_keys, or more (chains aligned to sectors)
    unsigned long long* d_xt_lines = nullptr;   // table keyed by the grid 16-mer (vgmi_xtable.hip), VGMI_XTABLE=1
    uint32_t* d_xt_counts = nullptr;
    uint32_t* d_xt_id = nullptr;                // key index -> counter id (path order), nullptr: identity
    ulonglong2* d_xt_over = nullptr;            // exact table of the k-mers that overflowed their lines (repeats), or nullptr
    uint4* d_ct_buckets = nullptr;              // context table (vgmi_ctable.hip): the default form of the large-graph table
    size_t ct_vmm_bytes = 0;                    // non-zero: d_ct_buckets is a mapping made by big_alloc (virtual memory API), of this size
    hipMemGenericAllocationHandle_t ct_vmm_handle{};
    uint64_t ct_entries = 0, ct_unitigs = 0, ct_moved = 0;   // entries built, unitigs they came from, entries not in their home bucket
    unsigned long long* d_pt_index = nullptr;   // path table of small graphs (build_ptable): 12-mer -> places in the unitig sequence
    uint32_t *d_pt_S = nullptr, *d_pt_VB = nullptr, *d_pt_SB = nullptr, *d_pt_SLOT = nullptr, *d_pt_PLACE = nullptr;   // sequence, k-mer starts, saturation bits, slots, places by slot
    size_t pt_sb_bytes = 0;
    uint64_t pt_slow_cx = 0, pt_bucket_ovf = 0;  // 12-mers with more than two places / buckets with a third 12-mer (those runs take the hash table)
    uint64_t xt_over_keys = 0;                  // pairs (key, 16-mer) that overflowed in the last build
    uint8_t* d_sat_dirty = nullptr;   // compact format: 2048-slot regions holding a saturation flag (the reset sweeps those)
    uint64_t n_sat_regions = 0;
    uint32_t* d_counts = nullptr;   // counter array (per-sample state, not part of the image): per key (large graphs)
                                    // or per slot (compact format); nullptr: in-slot counters
    uint64_t n_counts = 0;
    bool filter_in_lds = false;
    bool fast27 = false;         // k = 27: count27_kernel
    bool fast27_lds = false;     // ... with the 128 KiB grid filter resident in LDS
    bool fast27_small = false;   // ... over 12-mers: count27s_kernel (the default for graphs of <= 65 536 k-mers)
    bool fastk_small = false;    // odd k = 19 .. 25, graphs of <= 65 536 k-mers: the same kernel on a grid of 8 (two grid 12-mers per lane and row)
    uint32_t wgs_per_cu = 0;     // VGMI_WGS_PER_CU: tuning override for the global-bitmap variant
    bool force_generic = false;  // VGMI_GENERIC_KERNEL=1: take the generic row kernel (A/B testing)

    // nodes / flags / outputs
    size_t n_nodes = 0;
    uint64_t n_node_entries = 0;
    uint32_t* d_node_key_index = nullptr;
    uint8_t* d_flag = nullptr;
    uint8_t* d_cov = nullptr;
    uint8_t* d_cov_node = nullptr;
    unsigned long long* d_hist = nullptr;
    uint32_t* d_status = nullptr;
    std::map<hipStream_t, std::pair<uint8_t*, size_t>> ctd_scratch;      // deferred counter updates (vgmi_ctdefer.hip): per stream that counts, records + rooms
    std::map<hipStream_t, unsigned long long*> debit_lists;      // even k on the fast path: per stream that counts, VG_DEBIT_LIST positions + a counter

    // per-sample state
    std::mutex mu;                 // event list / counters below when several FASTQ streams submit from their own threads
    int open_fastq = 0;
    std::vector<struct vgmi_fastq*> fastq_pool;   // closed streams keep their pinned and device buffers for the next file
    uint64_t read_base = 0;
    hipEvent_t reset_done = nullptr;   // recorded on the main stream behind the per-sample reset
    Stage stage[2];
    int next_stage = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;  // (start, stop) of count launches
    std::vector<hipEvent_t> event_pool;
    float kernel_ms = 0.f;
    uint64_t launches = 0;

    // bloom
    bool has_bloom = false;
    uint8_t* bb_scratch = nullptr;       // K3, binned form: k-mer keys + the two levels of binned positions (kept between calls)
    size_t bb_cap = 0;
    BloomView bv{};
    uint64_t bloom_seeds64[VG_BLOOM_MAX_HASH] = {0};   // as handed in (the file format keeps all 64 bits)
    size_t bloom_alloc = 0;
};

namespace vgapi {

extern thread_local std::string g_create_error;
int fail(vgmi_ctx* c, int code, const std::string& msg);
uint32_t ceil_log2(uint64_t x);
hipEvent_t get_event(vgmi_ctx* c);
void free_table(vgmi_ctx* c);
void free_nodes(vgmi_ctx* c);
// vgmi_api_table.cpp
void layout_image(ImageHeader& h, uint32_t k, uint64_t n_keys);
int adopt_image(vgmi_ctx* c);
bool xtable_wanted(const ImageHeader& h);
bool ctable_wanted(const ImageHeader& h);
// vgmi_api.cpp: counting
int launch_count(vgmi_ctx* c, const char* d_bases, size_t n_bytes, const uint64_t* d_read_off, size_t n_reads, hipStream_t st,
                 const unsigned long long* n_bytes_dev = nullptr);
int collect_timing(vgmi_ctx* c);
int ensure_stage(vgmi_ctx* c, Stage& s);
int sync_stages(vgmi_ctx* c);
std::vector<uint64_t> offsets_from_newlines(const char* b, size_t n);
int check_status(vgmi_ctx* c);
RowParams row_params(vgmi_ctx* c, const char* d_bases, size_t n_bytes, uint32_t k);
void rows_geometry(vgmi_ctx* c, bool flds, uint32_t& grid, uint32_t& block);
// vgmi_api_fastq.cpp
void fastq_free(vgmi_fastq* f);

}  // namespace vgapi
using namespace vgapi;

#define HIPCHK(c, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char b_[512];                                                                          \
            snprintf(b_, sizeof b_, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call,             \
                     hipGetErrorString(e_));                                                       \
            return fail((c), VGMI_E_HIP, b_);                                                      \
        }                                                                                          \
    } while (0)

#endif
