// vgmi_api.cpp -- the C ABI of include/vgmi.h over the gfx950 kernels (vgmi_kernels.hip).
//
// Host-side plumbing only: contexts, device memory, pinned double-buffered staging, streams and
// events.  There is no CPU implementation of any compute path in here -- without a HIP device
// vgmi_create fails with VGMI_E_NO_DEVICE.
#include "../../include/vgmi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "vg_synth.h"
#include "vgmi_device.h"
#include "vgmi_kernels.h"

using namespace vgk;

namespace {

thread_local std::string g_create_error;

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

}  // namespace

struct ncclUniqueIdBytes { char internal[128]; };      // rccl.h: ncclUniqueId (passed by value to ncclCommInitRank)

struct vgmi_ctx {
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

namespace {

void fastq_free(vgmi_fastq* f);
struct ImageHeader;
bool xtable_wanted(const ImageHeader& h);
bool ctable_wanted(const ImageHeader& h);

int fail(vgmi_ctx* c, int code, const std::string& msg)
{
    static std::mutex mu;       // vgmi_hmm_calls_part may fail on several threads of one context
    std::lock_guard<std::mutex> lock(mu);
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

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

uint32_t ceil_log2(uint64_t x)
{
    uint32_t l = 0;
    while ((1ULL << l) < x) ++l;
    return l;
}

hipEvent_t get_event(vgmi_ctx* c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void free_table(vgmi_ctx* c)
{
    if (c->d_image) (void)hipFree(c->d_image);
    c->d_image = nullptr;
    if (c->d_snapshot) (void)hipFree(c->d_snapshot);
    c->d_snapshot = nullptr;
    c->image_bytes = 0;
    c->has_table = false;
    if (c->d_cov) (void)hipFree(c->d_cov);
    c->d_cov = nullptr;
    if (c->d_flag) (void)hipFree(c->d_flag);
    c->d_flag = nullptr;
    if (c->d_counts) (void)hipFree(c->d_counts);
    c->d_counts = nullptr;
    if (c->d_sat_dirty) (void)hipFree(c->d_sat_dirty);
    c->d_sat_dirty = nullptr;
    if (c->d_xt_lines) (void)hipFree(c->d_xt_lines);
    if (c->d_xt_counts) (void)hipFree(c->d_xt_counts);
    if (c->d_xt_id) (void)hipFree(c->d_xt_id);
    if (c->d_xt_over) (void)hipFree(c->d_xt_over);
    c->d_xt_over = nullptr;
    c->xt_over_keys = 0;
    if (c->d_ct_buckets && c->ct_vmm_bytes) {
        (void)hipMemUnmap(c->d_ct_buckets, c->ct_vmm_bytes);
        (void)hipMemRelease(c->ct_vmm_handle);
        (void)hipMemAddressFree(c->d_ct_buckets, c->ct_vmm_bytes);
        c->ct_vmm_bytes = 0;
    } else if (c->d_ct_buckets) (void)hipFree(c->d_ct_buckets);
    c->d_ct_buckets = nullptr;
    c->ct_entries = c->ct_unitigs = c->ct_moved = 0;
    for (void* q : {(void*)c->d_pt_index, (void*)c->d_pt_S, (void*)c->d_pt_VB, (void*)c->d_pt_SB, (void*)c->d_pt_SLOT, (void*)c->d_pt_PLACE})
        if (q) (void)hipFree(q);
    c->d_pt_index = nullptr;
    c->d_pt_S = c->d_pt_VB = c->d_pt_SB = c->d_pt_SLOT = c->d_pt_PLACE = nullptr;
    c->tv.pt = PathView{};
    c->d_xt_lines = nullptr;
    c->d_xt_counts = nullptr;
    c->d_xt_id = nullptr;
    c->tv.xt = XTableView{};
}

void free_nodes(vgmi_ctx* c)
{
    if (c->d_node_key_index) (void)hipFree(c->d_node_key_index);
    if (c->d_cov_node) (void)hipFree(c->d_cov_node);
    c->d_node_key_index = nullptr;
    c->d_cov_node = nullptr;
    c->n_nodes = 0;
    c->n_node_entries = 0;
}

void layout_image(ImageHeader& h, uint32_t k, uint64_t n_keys)
{
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "VGMITBL1", 8);
    h.k = k;
    h.n_keys = n_keys;
    // k = 27: compact 8-byte slots (a minimiser bucket of 32 is two 128-byte lines, its counters one); small graphs
    // spend the same bytes on twice the slots
    // ... and so do small graphs of k = 19 .. 25: the 12-mer grid and the path table serve them too (count27s_kernel<true, K>, round 5;
    // VGMI_SMALLK=0 keeps them on the generic row kernel, the A/B reference)
    // (even k = 20 .. 24 as well: the kernel's rule is the odd one, the debit pass runs ahead of it with the reference's)
    const bool smallk = k >= 19 && k <= 25 && n_keys <= VG_GRID_LDS_MAX_KEYS && !(getenv("VGMI_SMALLK") && getenv("VGMI_SMALLK")[0] == '0');
    // ... and graphs of k = 19 .. 25 too large for that: the context table is built from the compact image (xtable_wanted)
    const bool off_k = (getenv("VGMI_CTABLE_K") && getenv("VGMI_CTABLE_K")[0] == '0') || (getenv("VGMI_CTABLE") && getenv("VGMI_CTABLE")[0] == '0') ||
                       (getenv("VGMI_XTABLE") && getenv("VGMI_XTABLE")[0] == '0');
    // ... and k = 26 at any size: its runs of k + 7 bases do not fit the path-table kernel's two words, the context table's flanks of 10 do
    const bool largek = ((k >= 19 && k <= 25 && n_keys > VG_GRID_LDS_MAX_KEYS) || (k == 26 && n_keys > 0)) && n_keys < (1ULL << 31) - 16 && !off_k;
    const bool compact = (k == 27 || smallk || largek) && !getenv("VGMI_WIDE_SLOTS");   // 8-byte k-mer words + per-slot counters
    h.slot_bytes = compact ? 8 : 16;
    uint64_t cap = 64;
    uint64_t lf_mul = compact ? 8 : 4;   // load factor <= 0.125 / 0.25: nearly every probe ends at the first slot
    if (const char* e = getenv("VGMI_TABLE_MUL")) lf_mul = (uint64_t)atoi(e) > 1 ? (uint64_t)atoi(e) : 2;
    while (cap < lf_mul * n_keys) cap <<= 1;
    while (cap > (1ULL << 32) && cap / 2 >= 2 * n_keys) cap >>= 1;   // slot numbers are 32-bit (key_slot)
    h.cap = cap;
    // tables that live in HBM (k = 27, global grid filter): home slots in minimiser buckets of 32 slots (512 bytes), so
    // the k-mers of neighbouring read positions probe the same few lines (vg_thash_local; VGMI_LOCALITY=0 switches
    // it off, another value sets the bucket size)
    h.home_bucket_log2 = 0;
    if (k == 27 && n_keys > VG_GRID_LDS_MAX_KEYS) {
        h.home_bucket_log2 = 5;
        if (const char* e = getenv("VGMI_LOCALITY")) h.home_bucket_log2 = (uint32_t)atoi(e) < 16 ? (uint32_t)atoi(e) : 5;
        // A/B knob (off: measured -5 % on the dense chr20-class graph, +4 % on the 1.2 Gb one -- more k-mers share a
        // (minimiser, offset) pair than a hashed place inside the bucket makes collide)
        h.home_by_offset = 0;
        if (const char* e = getenv("VGMI_SLOT_ORDER")) h.home_by_offset = atoi(e) != 0 && h.home_bucket_log2 >= 5;
    }
    // prefilter: >= 16 bits per key, power of two, at least 128 bits
    uint64_t bits = 128;
    while (bits < 16 * n_keys) bits <<= 1;
    h.filter_words_log2 = ceil_log2(bits) - 5;
    auto align = [](uint64_t x) { return (x + 255) & ~255ULL; };
    h.off_slots = 256;
    h.off_key_slot = align(h.off_slots + cap * h.slot_bytes);
    h.off_filter = align(h.off_key_slot + (n_keys ? n_keys : 1) * 4);
    // grid filter of the k = 27 kernels (vgmi_device.h): 2^15 words in LDS while the graph is small,
    // else >= 32 bits per key in global memory (~2.8 sixteen-mers per key, 3 bits each)
    h.grid_words_log2 = 0;
    h.off_grid = 0;
    uint64_t end = h.off_filter + (4ULL << h.filter_words_log2);
    if (k == 27 || (smallk && compact)) {
        uint32_t b = VG_GRID_LDS_WORDS_LOG2;
        uint64_t entry_bytes = 4;
        if (n_keys > VG_GRID_LDS_MAX_KEYS) {
            // global variant: 64-bit entries (Bloom word + offset bits), >= 32 bits per key in all
            b = ceil_log2(32 * n_keys) - 6;
            if (const char* e = getenv("VGMI_GRID_SHIFT")) b = (uint32_t)((int)b + atoi(e));   // A/B: grid filter size
            if (b < VG_GRID_LDS_WORDS_LOG2 + 1) b = VG_GRID_LDS_WORDS_LOG2 + 1;
            if (b > 31) b = 31;   // vg_grid_probe draws the entry index from a 32-bit product word
            entry_bytes = 8;
        }
        h.grid_words_log2 = b;
        // small graphs (LDS-resident filter): 12-mer grid, 16 bytes per lane (count27s_kernel); VGMI_GRID12=0 keeps the
        // 16-mer grid and count27_kernel<true, true> of rounds 1-2 as the A/B reference
        h.grid_mer = 16;
        if (b == VG_GRID_LDS_WORDS_LOG2 && compact) {
            const char* e = getenv("VGMI_GRID12");
            if (!(e && e[0] == '0')) h.grid_mer = 12;
        }
        h.off_grid = align(end);
        end = h.off_grid + (entry_bytes << b);
    }
    h.total_bytes = align(end);
}

// LDS budget of the count kernel with an LDS-resident filter: filter + 16 wave queues + LUTs
bool filter_fits_lds(uint32_t words_log2) { return (4ULL << words_log2) + 16 * 128 * 8 + 512 <= 160 * 1024; }

int adopt_image(vgmi_ctx* c)
{
    const ImageHeader& h = c->hdr;
    const bool compact = h.slot_bytes == 8;
    c->tv.slots = compact ? nullptr : reinterpret_cast<VgSlot*>(c->d_image + h.off_slots);
    c->tv.slots8 = compact ? reinterpret_cast<unsigned long long*>(c->d_image + h.off_slots) : nullptr;
    c->tv.cap_mask = h.cap - 1;
    c->tv.home_bucket_log2 = h.home_bucket_log2;
    c->tv.home_by_offset = h.home_by_offset;
    c->tv.filter = reinterpret_cast<const uint32_t*>(c->d_image + h.off_filter);
    c->tv.filter_words_log2 = h.filter_words_log2;
    c->tv.filter_shift = 32 - h.filter_words_log2;
    c->tv.grid = h.off_grid ? reinterpret_cast<const uint32_t*>(c->d_image + h.off_grid) : nullptr;
    c->tv.grid_words_log2 = h.grid_words_log2;
    const bool lds_grid = h.grid_words_log2 == VG_GRID_LDS_WORDS_LOG2;
    c->fast27 = h.k == 27 && h.off_grid && (!lds_grid || compact);   // count27_kernel applies: LDS filter + compact
                                                                  // slots, or global (64-bit entry) filter + 16-byte slots
    c->fast27_lds = c->fast27 && lds_grid;
    c->fast27_small = c->fast27_lds && h.grid_mer == 12;   // count27s_kernel
    c->fastk_small = h.k != 27 && compact && h.off_grid && lds_grid && h.grid_mer == 12;   // count27s_kernel<true, K>, K = 19 .. 25
    c->tv.k = h.k;
    c->d_key_slot = reinterpret_cast<uint32_t*>(c->d_image + h.off_key_slot);
    c->filter_in_lds = filter_fits_lds(h.filter_words_log2);
    HIPCHK(c, hipMalloc(&c->d_cov, h.n_keys ? h.n_keys : 1));
    c->tv.counts = nullptr;
    c->n_counts = 0;
    if (compact && !xtable_wanted(h)) c->n_counts = h.cap;              // per-slot counters (the grid-16-mer table has its own, per key)
    else if (h.n_keys > VG_GRID_LDS_MAX_KEYS && (h.home_bucket_log2 == 0 || getenv("VGMI_DENSE_COUNTS")) && !(h.k != 27 && xtable_wanted(h)))
        c->n_counts = h.n_keys;   // randomly placed slots: 4 B/key dense counters stay Infinity-Cache resident
    // (minimiser buckets: the counter lives in the slot, the atomic hits the line its probe has just fetched)
    if (c->n_counts) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_counts), c->n_counts * 4));
        HIPCHK(c, hipMemset(c->d_counts, 0, c->n_counts * 4));
        c->tv.counts = c->d_counts;
    }
    c->tv.sat_dirty = nullptr;
    if (compact) {
        c->n_sat_regions = ((h.cap - 1) >> 11) + 1;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_sat_dirty), c->n_sat_regions));
        HIPCHK(c, hipMemset(c->d_sat_dirty, 1, c->n_sat_regions));   // flags of unknown origin (an imported image): the first reset sweeps everything
        c->tv.sat_dirty = c->d_sat_dirty;
    }
    c->has_table = true;
    free_nodes(c);
    return VGMI_OK;
}

// the table keyed by the grid 16-mer, built from the compact image (k-mers = slots8[key_slot[i]]): after an upload, an
// import and a clone alike
// k = 27 graphs that live in HBM count through the grid-16-mer table (VGMI_XTABLE=0: the minimiser-bucket table and
// count27_kernel<false, *> of round 1, kept as the A/B reference)
bool xtable_wanted(const ImageHeader& h)
{
    const char* e = getenv("VGMI_XTABLE");
    if ((e && e[0] == '0') || h.slot_bytes != 8 || (h.n_keys <= VG_GRID_LDS_MAX_KEYS && !(h.k == 26 && h.n_keys > 0))) return false;
    // k = 19 .. 25, odd (round 5): the context table only (flanks of k - 16 bases, vgmi_ctable.h); VGMI_CTABLE_K=0 keeps them on the generic kernel (A/B)
    if (h.k >= 19 && h.k <= 26) {      // (even k too: the pass that takes back what the reference's run counter suppresses runs ahead of the kernel)
        const char* o = getenv("VGMI_CTABLE_K");
        return !(o && o[0] == '0') && ctable_wanted(h);
    }
    return h.k == 27;
}

// lines + overflow table of one numbering of the keys (id_of_key, or the key index).  The (key, 16-mer) pairs that find no
// room within XT_HOPS lines of home are collected on a list (one pass that only counts when the list is too short, then
// again with a list that fits), and their keys go into the exact overflow table.
static int xtable_fill(vgmi_ctx* c, XTableView& x, const uint32_t* id_of_key)
{
    const uint64_t n = c->hdr.n_keys;
    unsigned long long* d_n = nullptr;
    uint32_t* d_list = nullptr;
    uint64_t cap = 1u << 16;
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_n), 8);
    unsigned long long n_over = 0;
    for (int pass = 0; he == hipSuccess && pass < 2; ++pass) {
        he = hipMalloc(reinterpret_cast<void**>(&d_list), cap * 4);
        if (he == hipSuccess) he = hipMemsetAsync(d_n, 0, 8, c->stream);
        if (he == hipSuccess) he = launch_xtable_build(x, c->tv.slots8, c->d_key_slot, id_of_key, n, d_list, (uint32_t)cap, d_n, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he == hipSuccess) he = hipMemcpy(&n_over, d_n, 8, hipMemcpyDeviceToHost);
        if (he != hipSuccess || n_over <= cap) break;
        // which pairs overflow depends on the order the threads arrive in: leave room
        (void)hipFree(d_list);
        d_list = nullptr;
        cap = n_over + n_over / 4 + 1024;
        if (cap >= (1ULL << 32)) {
            (void)hipFree(d_n);
            return fail(c, VGMI_E_NOMEM, "grid-16-mer table: too many k-mers of repeats");
        }
    }
    if (he == hipSuccess && n_over > cap) he = hipErrorOutOfMemory;
    if (c->d_xt_over) (void)hipFree(c->d_xt_over);
    c->d_xt_over = nullptr;
    x.over = nullptr;
    x.over_mask = 0;
    c->xt_over_keys = n_over;
    if (he == hipSuccess && n_over) {
        uint64_t slots = 1024;
        while (slots < 2 * n_over) slots <<= 1;
        he = hipMalloc(reinterpret_cast<void**>(&c->d_xt_over), slots * 16);
        if (he == hipSuccess) he = launch_xtable_over(c->d_xt_over, (uint32_t)(slots - 1), c->tv.slots8, c->d_key_slot, id_of_key, d_list, n_over, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        x.over = c->d_xt_over;
        x.over_mask = (uint32_t)(slots - 1);
    }
    if (d_list) (void)hipFree(d_list);
    if (d_n) (void)hipFree(d_n);
    HIPCHK(c, he);
    return VGMI_OK;
}

// The context table (vgmi_ctable.h), built from the compact image like the grid-16-mer table it replaces: the device orders the
// k-mers along their unitigs (vgmi_ptable.hip's numbering, which also numbers the counters), every occurrence of a 16-mer in a
// unitig becomes one 16-byte entry, buckets of four at <= 30 % load (VGMI_CTABLE_LOAD=percent for A/B), entries that find
// CT_HOPS + 1 buckets full send their k-mers to the exact overflow table.  VGMI_CTABLE=0 keeps the grid-16-mer table and
// count27x_kernel of round 2 as the A/B reference.
bool ctable_wanted(const ImageHeader& h)
{
    const char* e = getenv("VGMI_CTABLE");
    return !(e && e[0] == '0') && h.n_keys < (1ULL << 31) - 16;
}

int build_ctable(vgmi_ctx* c)
{
    const ImageHeader& h = c->hdr;
    const uint64_t n = h.n_keys;
    XTableView x{};
    uint32_t *key_of_slot = nullptr, *link = nullptr, *link2 = nullptr, *pos = nullptr, *mark = nullptr, *d_list = nullptr;
    unsigned long long *cursor = nullptr, *okmer = nullptr;      // cursor[0] numbering, [1] unitigs, [2] overflowed k-mers, [3] moved entries
    auto cleanup = [&]() {
        for (void* q : {(void*)key_of_slot, (void*)link, (void*)link2, (void*)pos, (void*)mark, (void*)d_list, (void*)cursor, (void*)okmer})
            if (q) (void)hipFree(q);
    };
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&key_of_slot), h.cap * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&link), n * 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&link2), n * 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&pos), n * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&cursor), 32);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_xt_id), n * 4);
    if (he == hipSuccess) he = hipMemsetAsync(pos, 0xFF, n * 4, c->stream);
    if (he == hipSuccess) he = hipMemsetAsync(cursor, 0, 32, c->stream);
    // places along the unitigs.  VGMI_CTABLE_ALIGN=16 starts every chain at a multiple of 16, so that the counters of a chain's
    // first 16 k-mers share a 64-byte sector (a read's hits on a site: 1.8 sectors instead of 2.3) -- built, measured, no gain
    // (chr20 class 8.62 against 8.57 ms, gpurun_out/r4d: the atomics cost per lane operation, not per sector), so places are dense
    uint32_t align = 1;
    if (const char* e = getenv("VGMI_CTABLE_ALIGN")) align = atoi(e) >= 1 && atoi(e) <= 64 ? (uint32_t)atoi(e) : align;
    if (he == hipSuccess) he = launch_ptable_order(c->tv, c->d_key_slot, n, key_of_slot, link, link2, pos, cursor, nullptr, c->d_status, c->stream, align);
    unsigned long long cur[4] = {0, 0, 0, 0};
    uint32_t st = 0;
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess) he = hipMemcpy(cur, cursor, 32, hipMemcpyDeviceToHost);
    uint64_t total = cur[0];
    bool identity = total < n || total >= (1ULL << 31) - 16;
    if (he == hipSuccess && !identity) {
        he = hipMalloc(reinterpret_cast<void**>(&mark), total * 4);
        if (he == hipSuccess) he = hipMemsetAsync(mark, 0, total * 4, c->stream);
        if (he == hipSuccess) he = launch_ptable_check(pos, n, total, mark, c->d_status, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he == hipSuccess) he = hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost);
        // two keys on one place (cannot happen): any numbering is correct, the key index is one -- every k-mer a unitig of its own
        if (he == hipSuccess && (st & 16u)) {
            identity = true;
            st &= ~16u;
            he = hipMemcpy(c->d_status, &st, 4, hipMemcpyHostToDevice);
        }
    }
    if (identity) total = n;
    c->xt_n_counts = total;
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&okmer), total * 8);
    if (he == hipSuccess) he = hipMemsetAsync(okmer, 0xFF, total * 8, c->stream);       // places no k-mer has: bit 63 set
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_xt_counts), total * 4);
    if (he == hipSuccess) he = hipMemsetAsync(c->d_xt_counts, 0, total * 4, c->stream);
    if (he == hipSuccess) he = launch_ctable_okmer(c->tv, c->d_key_slot, pos, link2, n, identity, okmer, c->d_xt_id, cursor + 1, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess) he = hipMemcpy(cur, cursor, 32, hipMemcpyDeviceToHost);
    for (void* q : {(void*)key_of_slot, (void*)link, (void*)link2, (void*)pos, (void*)mark})
        if (q) (void)hipFree(q);
    key_of_slot = link = link2 = pos = mark = nullptr;
    if (he != hipSuccess) {
        cleanup();
        HIPCHK(c, he);
    }
    c->ct_unitigs = cur[1];
    c->ct_entries = n + (h.k - 16) * cur[1];  // a unitig of L k-mers holds L + k - 16 occurrences (palindromic 16-mers: two entries, rare)
    double load = 0.30;     // measured, chr20 / whole-genome class kernel ms: 25 % 8.25 / -, 30 % 8.17 / 30.1, 40 % 8.41 / 33.4 (gpurun_out/r4c)
    if (const char* e = getenv("VGMI_CTABLE_LOAD")) load = atoi(e) >= 5 && atoi(e) <= 95 ? atoi(e) / 100.0 : load;
    uint64_t n_buckets = (uint64_t)((double)c->ct_entries / (4.0 * load)) + 1;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && n_buckets * 64 > free_b / 2) n_buckets = free_b / 2 / 64;
    if (n_buckets < (1u << 16)) n_buckets = 1u << 16;
    if (n_buckets >= (1ULL << 32) - 8 || n_buckets * 8 < c->ct_entries) {      // (what does not fit goes to the overflow table; below half, that table is the table)
        cleanup();
        return fail(c, VGMI_E_NOMEM, "not enough device memory for the context table");
    }
    // VGMI_CT_VMM=<MiB>: the table as ONE physical allocation mapped at a virtual address aligned to that many MiB (the virtual-memory
    // API), instead of hipMalloc's placement -- the experiment on the process-to-process spread of the whole-genome-class kernel
    // (29-34 ms in round 4: 20 GB of random 64-byte reads are one address translation each)
    {
        const size_t want = (size_t)64 * (n_buckets + CT_HOPS);
        const char* ev = getenv("VGMI_CT_VMM");
        const size_t align_mib = ev ? (size_t)atol(ev) : 0;
        bool done = false;
        if (align_mib >= 2) {
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = c->device;
            size_t gran = 0;
            if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
                const size_t bytes = (want + gran - 1) / gran * gran;
                void* va = nullptr;
                hipMemGenericAllocationHandle_t h{};
                if (hipMemAddressReserve(&va, bytes, align_mib << 20, nullptr, 0) == hipSuccess) {
                    if (hipMemCreate(&h, bytes, &prop, 0) == hipSuccess) {
                        hipMemAccessDesc acc{};
                        acc.location = prop.location;
                        acc.flags = hipMemAccessFlagsProtReadWrite;
                        if (hipMemMap(va, bytes, 0, h, 0) == hipSuccess && hipMemSetAccess(va, bytes, &acc, 1) == hipSuccess) {
                            c->d_ct_buckets = static_cast<uint4*>(va);
                            c->ct_vmm_bytes = bytes;
                            c->ct_vmm_handle = h;
                            done = true;
                            if (getenv("VGMI_VERBOSE")) fprintf(stderr, "[vgmi] context table: %zu bytes mapped at %p (granularity %zu)\n", bytes, va, gran);
                        } else {
                            (void)hipMemRelease(h);
                            (void)hipMemAddressFree(va, bytes);
                        }
                    } else (void)hipMemAddressFree(va, bytes);
                }
                (void)hipGetLastError();
            }
        }
        he = done ? hipSuccess : hipMalloc(reinterpret_cast<void**>(&c->d_ct_buckets), want);
        if (!done && getenv("VGMI_VERBOSE")) fprintf(stderr, "[vgmi] context table: %zu bytes by hipMalloc at %p\n", want, (void*)c->d_ct_buckets);
    }
    x.cb = c->d_ct_buckets;
    x.k = h.k;
    x.n_buckets = (uint32_t)n_buckets;
    x.counts = c->d_xt_counts;
    uint64_t cap = 1u << 16;
    unsigned long long n_over = 0;
    for (int pass = 0; he == hipSuccess && pass < 2; ++pass) {
        he = hipMalloc(reinterpret_cast<void**>(&d_list), cap * 4);
        if (he == hipSuccess) he = hipMemsetAsync(cursor + 2, 0, 16, c->stream);
        if (he == hipSuccess) he = launch_ctable_build(x, okmer, total, d_list, (uint32_t)cap, cursor + 2, cursor + 3, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he == hipSuccess) he = hipMemcpy(cur, cursor, 32, hipMemcpyDeviceToHost);
        n_over = cur[2];
        if (he != hipSuccess || n_over <= cap) break;
        // which entries overflow depends on the order the threads arrive in: leave room
        (void)hipFree(d_list);
        d_list = nullptr;
        cap = n_over + n_over / 4 + 1024;
        if (cap >= (1ULL << 32)) {
            cleanup();
            return fail(c, VGMI_E_NOMEM, "context table: too many k-mers of repeats");
        }
    }
    if (he == hipSuccess && n_over > cap) he = hipErrorOutOfMemory;
    c->ct_moved = cur[3];
    c->xt_over_keys = n_over;
    if (he == hipSuccess && n_over) {
        uint64_t slots = 1024;
        while (slots < 2 * n_over) slots <<= 1;
        he = hipMalloc(reinterpret_cast<void**>(&c->d_xt_over), slots * 16);
        if (he == hipSuccess) he = launch_ctable_over(c->d_xt_over, (uint32_t)(slots - 1), okmer, d_list, n_over, h.k, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        x.over = c->d_xt_over;
        x.over_mask = (uint32_t)(slots - 1);
    }
    cleanup();
    HIPCHK(c, he);
    c->tv.xt = x;
    return VGMI_OK;
}

int build_xtable(vgmi_ctx* c)
{
    const ImageHeader& h = c->hdr;
    if (!xtable_wanted(h)) return VGMI_OK;
    if (ctable_wanted(h)) return build_ctable(c);
    XTableView x{};
    // lines of 16 slots at 25 % load (measured, chr20 / WGS class: 31 % 11.4 / 45.9 ms, 25 % 10.1 / 43.4, 20 % 9.9 / 41.8;
    // VGMI_XTABLE_LOAD=percent for A/B); never more than half of the free device memory
    double load = 0.25;
    if (const char* e = getenv("VGMI_XTABLE_LOAD")) load = atoi(e) >= 5 && atoi(e) <= 90 ? atoi(e) / 100.0 : load;
    uint64_t n_lines = (uint64_t)((double)h.n_keys * 12.0 / (16.0 * load)) + 1;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && n_lines * 128 > free_b / 2) n_lines = free_b / 2 / 128;
    if (n_lines < (1u << 20)) n_lines = 1u << 20;
    if (n_lines >= (1ULL << 31) || n_lines * 16 < h.n_keys * 13) return fail(c, VGMI_E_NOMEM, "not enough device memory for the grid-16-mer table");
    x.n_lines = (uint32_t)n_lines;
    x.k = 27;
    // an entry found in line P has its home in P - XT_HOPS .. P: at most (XT_HOPS + 1) * ceil(2^32 / n_lines) + 1 consecutive
    // h-values, told apart by their low tag_bits
    x.tag_bits = ceil_log2((XT_HOPS + 1) * (((1ULL << 32) + n_lines - 1) / n_lines) + 1);
    x.id_shift = 26 + x.tag_bits;
    if (h.n_keys >= (1ULL << (64 - x.id_shift)) - 1) return fail(c, VGMI_E_INVALID, "too many keys for the grid-16-mer table");
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_xt_lines), (size_t)128 * (n_lines + XT_HOPS)));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_xt_counts), h.n_keys * 4));
    HIPCHK(c, hipMemsetAsync(c->d_xt_counts, 0, h.n_keys * 4, c->stream));
    c->xt_n_counts = h.n_keys;
    x.lines = c->d_xt_lines;
    x.counts = c->d_xt_counts;
    int rc = xtable_fill(c, x, nullptr);
    if (rc != VGMI_OK) return rc;
    // counter ids in path order (xtable_number_*; any numbering is correct, VGMI_XTABLE_ORDER=0 keeps the key index)
    const char* ord = getenv("VGMI_XTABLE_ORDER");
    if (!(ord && ord[0] == '0')) {
        const uint64_t n = h.n_keys;
        uint32_t *link = nullptr, *link2 = nullptr, *mark = nullptr;
        unsigned long long* cursor = nullptr;
        uint32_t st = 0;
        hipError_t he = hipMalloc(reinterpret_cast<void**>(&link), n * 8);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&link2), n * 8);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&mark), n * 4);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&cursor), 8);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_xt_id), n * 4);
        if (he == hipSuccess) he = hipMemsetAsync(c->d_xt_id, 0xFF, n * 4, c->stream);
        if (he == hipSuccess) he = hipMemsetAsync(mark, 0, n * 4, c->stream);
        if (he == hipSuccess) he = hipMemsetAsync(cursor, 0, 8, c->stream);
        if (he == hipSuccess) he = launch_xtable_number(x, c->tv.slots8, c->d_key_slot, n, link, link2, c->d_xt_id, cursor, mark, c->d_status, c->stream);
        unsigned long long used = 0;
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he == hipSuccess) he = hipMemcpy(&used, cursor, 8, hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost);
        for (void* q : {(void*)link, (void*)link2, (void*)mark, (void*)cursor})
            if (q) (void)hipFree(q);
        HIPCHK(c, he);
        if ((st & 16u) || used != n) {      // not a permutation (cannot happen; the identity numbering is always right)
            (void)hipFree(c->d_xt_id);
            c->d_xt_id = nullptr;
            HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4, c->stream));
        } else {
            rc = xtable_fill(c, x, c->d_xt_id);
            if (rc != VGMI_OK) return rc;
        }
    }
    c->tv.xt = x;
    return VGMI_OK;
}

// the path table of small graphs, derived from the compact image like the grid-16-mer table of large ones (after an upload,
// an import and a clone alike); VGMI_PTABLE=0 keeps count27s_kernel on the hash table alone (A/B).  The device orders the
// k-mers along their unitigs (vgmi_ptable.hip); the layout of the sequence and the index is host work over <= 65 536 k-mers.
int build_ptable(vgmi_ctx* c)
{
    const ImageHeader& h = c->hdr;
    c->tv.pt = PathView{};
    if (!(c->fast27_small || c->fastk_small) || h.n_keys == 0) return VGMI_OK;
    const uint32_t K = h.k;                     // 27, or 19 .. 25 (the grid of 8)
    // the run a lane compares: lead bases in front of the grid 12-mer, the 12-mer, the bases behind it (vgmi_kernels.hip)
    const uint32_t lead = K == 27 ? 15u : K - 12u;
    if (const char* e = getenv("VGMI_PTABLE"))
        if (e[0] == '0') return VGMI_OK;
    const uint64_t n = h.n_keys;
    // >= 8 buckets of 32 bytes per k-mer (16 MiB for a 6.5e4-k-mer graph with its ~7e4 distinct canonical 12-mers): a third 12-mer is
    // wanted in ~0.03 % of the buckets.  Every run of such a 12-mer takes the hash table, window by window: with 2^17 buckets (1 %)
    // that was 1.8 ms of 6.0 per 1e8 reads, with 2^18 (0.4 %) 0.56 of 5.0 (VGMI_DBG=4096 ablation); only the buckets of 12-mers
    // that occur are ever read twice, so the size costs address space, not cache.
    uint32_t bucket_log2 = 13;
    while (bucket_log2 < 19 && (1ull << bucket_log2) < 8 * n) ++bucket_log2;
    uint32_t *key_of_slot = nullptr, *link = nullptr, *link2 = nullptr, *pos = nullptr, *mark = nullptr;
    unsigned long long* cursor = nullptr;
    ulonglong2* d_P = nullptr;
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&key_of_slot), h.cap * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&link), n * 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&link2), n * 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&pos), n * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&mark), n * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&cursor), 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_P), 2 * n * sizeof(ulonglong2));
    if (he == hipSuccess) he = hipMemsetAsync(pos, 0xFF, n * 4, c->stream);
    if (he == hipSuccess) he = hipMemsetAsync(mark, 0, n * 4, c->stream);
    if (he == hipSuccess) he = hipMemsetAsync(cursor, 0, 8, c->stream);
    if (he == hipSuccess) he = launch_ptable_order(c->tv, c->d_key_slot, n, key_of_slot, link, link2, pos, cursor, mark, c->d_status, c->stream);
    unsigned long long used = 0;
    uint32_t st = 0;
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess) he = hipMemcpy(&used, cursor, 8, hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost);
    const bool ordered = !(st & 16u) && used == n;      // else (cannot happen): chains of one, in key order -- any layout is correct
    if (he == hipSuccess && (st & 16u)) he = hipMemsetAsync(c->d_status, 0, 4, c->stream);
    if (he == hipSuccess) he = launch_ptable_fill(c->tv, c->d_key_slot, ordered ? pos : nullptr, n, d_P, c->stream);
    std::vector<ulonglong2> P(2 * n);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess) he = hipMemcpy(P.data(), d_P, 2 * n * sizeof(ulonglong2), hipMemcpyDeviceToHost);
    for (void* q : {(void*)key_of_slot, (void*)link, (void*)link2, (void*)pos, (void*)mark, (void*)cursor, (void*)d_P})
        if (q) (void)hipFree(q);
    HIPCHK(c, he);

    // ---- layout (host).  P[0, n): the k-mers chain after chain, each in the orientation its chain is walked in; P[2n - 1 - i] is
    // the reverse complement of P[i].  A chain of L k-mers is L + k - 1 bases; the chains follow each other without a gap, the
    // second half of S is the reverse complement of the first, 32 bases of padding at either end.
    const uint64_t M54 = (1ULL << (2 * K)) - 1;      // (the k-mer's 2k bits)
    std::vector<uint32_t> kpos(n);
    uint64_t chains = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t a = i ? (uint64_t)P[i - 1].x & M54 : 0, b = (uint64_t)P[i].x & M54;
        if (i == 0 || (b >> 2) != (a & (M54 >> 2))) ++chains;
        kpos[i] = (uint32_t)(i + (K - 1) * (chains - 1));
    }
    const uint64_t Th = n + (K - 1) * chains, T = 2 * Th, Tp = T + 64;
    if (Tp + 64 >= (1u << 19) - 1) return VGMI_OK;      // places are 19-bit fields: a graph of very many very short chains keeps the hash table
    std::vector<uint8_t> base(Tp, 0), vb(Tp, 0);
    std::vector<uint32_t> slot(Tp, 0), chain_end(Tp, 0);       // chain_end[start of a chain's span] = its end (first half, unpadded)
    {
        uint64_t span_start = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t km = (uint64_t)P[i].x & M54;
            for (uint32_t t = 0; t < K; ++t) base[32 + kpos[i] + t] = (uint8_t)((km >> (2 * (K - 1 - t))) & 3u);
            // (even k: a k-mer that is its own reverse complement is never emitted, src/kmer.cpp:134 -- no start bit, never counted)
            bool own_rc = false;
            if (!(K & 1u)) {
                uint64_t r = 0;
                for (uint32_t t = 0; t < K; ++t) r |= (3ull - ((km >> (2 * t)) & 3ull)) << (2 * (K - 1 - t));
                own_rc = r == km;
            }
            vb[32 + kpos[i]] = !own_rc;
            slot[32 + kpos[i]] = (uint32_t)P[i].y;
            vb[32 + T - K - kpos[i]] = !own_rc;
            slot[32 + T - K - kpos[i]] = (uint32_t)P[i].y;
            const bool last_of_chain = i + 1 == n || kpos[i + 1] != kpos[i] + 1;
            if (last_of_chain) {
                chain_end[span_start] = (uint32_t)(kpos[i] + K);
                span_start = kpos[i] + K;
            }
        }
        for (uint64_t j = 0; j < Th; ++j) base[32 + T - 1 - j] = (uint8_t)(3u - base[32 + j]);
        for (uint64_t j = 0; j < 32; ++j) base[Tp - 1 - j] = (uint8_t)(3u - base[j]);     // the pads mirror each other too (zeros / threes)
    }
    const size_t s_words = (size_t)(Tp + 15) / 16 + 8, b_words = (size_t)(Tp + 31) / 32 + 4;
    std::vector<uint32_t> S(s_words, 0), VB(b_words, 0);
    for (uint64_t j = 0; j < Tp; ++j) {
        S[j >> 4] |= (uint32_t)base[j] << (2 * (15 - (j & 15)));
        if (vb[j]) VB[j >> 5] |= 1u << (j & 31);
    }
    // index: every occurrence of a 12-mer inside a chain's span that reads as its canonical form lists the place of the run's
    // first base (`lead` bases in front of it); the occurrence on the other strand is listed from the mirrored half
    // bucket = two 16-byte entries.  Entry: word 0 = 12-mer | place 0 << 24 | place 1 << 43 | (first entry only) "a 12-mer found no
    // entry here" << 62; word 1 = place 2 | place 3 << 19 | "more than four places" << 38.  Place 0 = 0: the entry is free.
    std::vector<unsigned long long> index((size_t)4 << bucket_log2, 0ULL);
    uint64_t slow_cx = 0, bucket_ovf = 0;
    auto add = [&](uint32_t x, uint32_t place) {
        unsigned long long* B = &index[(size_t)(vg_idx_hash(x) >> (32 - bucket_log2)) << 2];
        for (int e = 0; e < 2; ++e) {
            unsigned long long& lo = B[2 * e];
            unsigned long long& hi = B[2 * e + 1];
            const uint32_t q0 = (uint32_t)(lo >> 24) & 0x7FFFFu, q1 = (uint32_t)(lo >> 43) & 0x7FFFFu;
            const uint32_t q2 = (uint32_t)hi & 0x7FFFFu, q3 = (uint32_t)(hi >> 19) & 0x7FFFFu;
            if (q0 == 0) {
                lo = (lo & (1ULL << 62)) | x | (unsigned long long)place << 24;
                return;
            }
            if (((uint32_t)lo & 0xFFFFFFu) != x) continue;
            if (q1 == 0) lo |= (unsigned long long)place << 43;
            else if (q2 == 0) hi |= place;
            else if (q3 == 0) hi |= (unsigned long long)place << 19;
            else if (!(hi >> 38 & 1)) {
                hi |= 1ULL << 38;              // a fifth place: runs with this 12-mer take the hash table
                ++slow_cx;
            }
            return;
        }
        if (!(B[0] >> 62 & 1)) ++bucket_ovf;
        B[0] |= 1ULL << 62;                    // a third 12-mer in this bucket: lookups that miss here take the hash table
    };
    for (uint64_t s0 = 0; s0 < Th;) {
        const uint64_t e0 = chain_end[s0];
        for (int halfno = 0; halfno < 2; ++halfno) {
            const uint64_t lo = halfno ? T - e0 : s0, hi = halfno ? T - s0 : e0;      // the chain's span in this half (unpadded)
            uint32_t x = 0;
            for (uint64_t b = lo; b < hi; ++b) {
                x = ((x << 2) | base[32 + b]) & 0xFFFFFFu;
                if (b + 1 < lo + 12) continue;
                const uint64_t first = b + 1 - 12;                       // the 12-mer is bases first .. first + 11
                if (x <= vg_revcomp12(x)) add(x, (uint32_t)(32 + first - lead));
            }
        }
        s0 = e0;
    }
    c->pt_slow_cx = slow_cx;
    c->pt_bucket_ovf = bucket_ovf;
    he = hipMalloc(reinterpret_cast<void**>(&c->d_pt_index), index.size() * 8);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_pt_S), S.size() * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_pt_VB), VB.size() * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_pt_SB), VB.size() * 4);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_pt_SLOT), (size_t)(Tp + 64) * 4);
    if (he == hipSuccess) he = hipMemcpy(c->d_pt_index, index.data(), index.size() * 8, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(c->d_pt_S, S.data(), S.size() * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(c->d_pt_VB, VB.data(), VB.size() * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemset(c->d_pt_SB, 0, VB.size() * 4);
    if (he == hipSuccess) he = hipMemset(c->d_pt_SLOT, 0, (size_t)(Tp + 64) * 4);
    if (he == hipSuccess) he = hipMemcpy(c->d_pt_SLOT, slot.data(), (size_t)Tp * 4, hipMemcpyHostToDevice);
    {   // slot -> place: the slow paths -- the hash-table fallback of runs the index does not cover, the generic kernel on the
        // ragged tail -- know a k-mer by its slot; the increment of theirs that takes a counter to the clamp sets the path table's bits too
        std::vector<uint32_t> place_of_slot(h.cap, 0u);
        for (uint64_t i = 0; i < n; ++i) place_of_slot[(uint32_t)P[i].y] = (uint32_t)(32 + kpos[i]);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&c->d_pt_PLACE), h.cap * 4);
        if (he == hipSuccess) he = hipMemcpy(c->d_pt_PLACE, place_of_slot.data(), h.cap * 4, hipMemcpyHostToDevice);
    }
    HIPCHK(c, he);
    c->pt_sb_bytes = VB.size() * 4;
    if (getenv("VGMI_VERBOSE"))
        fprintf(stderr, "[vgmi] path table: %llu k-mers in %llu chains, %llu bases, %llu 12-mers with a fifth place, %llu buckets with a third 12-mer\n",
                (unsigned long long)n, (unsigned long long)chains, (unsigned long long)Tp, (unsigned long long)slow_cx, (unsigned long long)bucket_ovf);
    c->tv.pt.index = c->d_pt_index;
    c->tv.pt.S = c->d_pt_S;
    c->tv.pt.VB = c->d_pt_VB;
    c->tv.pt.SB = c->d_pt_SB;
    c->tv.pt.SLOT = c->d_pt_SLOT;
    c->tv.pt.PLACE = c->d_pt_PLACE;
    c->tv.pt.bucket_log2 = bucket_log2;
    c->tv.pt.Tp = (uint32_t)Tp;
    return VGMI_OK;
}

// grid-16-mer table: counters are bumped without a return value; before any could wrap (2^32 hits need > 2^31 submitted
// bytes), counters far above the read-out clamp are pulled back
// even k on the fast paths: the stream's list of non-base positions (launch_even_debit), made on first use
int debit_list_of(vgmi_ctx* c, hipStream_t st, unsigned long long** out)
{
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->debit_lists.find(st);
        if (it != c->debit_lists.end()) {
            *out = it->second;
            return VGMI_OK;
        }
    }
    unsigned long long* list = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&list), (size_t)VG_DEBIT_LIST * 8 + (size_t)VG_DEBIT_SUBLISTS * 64));
    std::lock_guard<std::mutex> lk(c->mu);
    c->debit_lists[st] = list;
    *out = list;
    return VGMI_OK;
}

// Deferred counter updates of the context-table kernels (vgmi_ctdefer.hip): whether this launch uses them, and the stream's scratch.
// VGMI_CT_DEFER=0|1 (A/B), VGMI_CT_DEFER_MIN: the smallest block in bytes that defers (smaller ones are not worth two more launches).
int ctd_prepare(vgmi_ctx* c, size_t n_bytes, hipStream_t st, CtDefer* d)
{
    // (read per launch: the test matrix switches them inside one process)
    const char* const e_on = getenv("VGMI_CT_DEFER");
    const int on = e_on ? atoi(e_on) : VGMI_CT_DEFER_DEFAULT;
    const char* const e_min = getenv("VGMI_CT_DEFER_MIN");
    const size_t min_bytes = e_min ? (size_t)atoll(e_min) : (size_t)8 << 20;
    *d = CtDefer{};
    if (!on || !c->tv.xt.cb || n_bytes < min_bytes) return VGMI_OK;
    const size_t need = ctd_scratch_bytes(n_bytes, c->xt_n_counts, (uint32_t)c->n_cu, d);
    if (!need) return VGMI_OK;
    uint8_t* buf = nullptr;
    size_t have = 0;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->ctd_scratch.find(st);
        if (it != c->ctd_scratch.end()) {
            buf = it->second.first;
            have = it->second.second;
        }
    }
    if (have < need) {
        // a larger block than this stream has seen: its scratch grows (what is queued on the stream may still use the old one)
        if (buf) {
            HIPCHK(c, hipStreamSynchronize(st));
            (void)hipFree(buf);
            buf = nullptr;
        }
        if (hipMalloc(reinterpret_cast<void**>(&buf), need) != hipSuccess) {      // no memory for it: the plain kernel
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(c->mu);
            c->ctd_scratch.erase(st);
            *d = CtDefer{};
            return VGMI_OK;
        }
        std::lock_guard<std::mutex> lk(c->mu);
        c->ctd_scratch[st] = std::make_pair(buf, need);
    }
    ctd_layout(buf, d);
    HIPCHK(c, launch_ctd_reset(*d, st));
    return VGMI_OK;
}

// the count kernel over the context table with or without deferred counter updates
int launch_ctable_count(vgmi_ctx* c, const RowParams& p, size_t n_bytes, hipStream_t st)
{
    CtDefer d;
    int rc = ctd_prepare(c, n_bytes, st, &d);
    if (rc) return rc;
    HIPCHK(c, launch_count27c(p, c->tv.xt, (uint32_t)c->n_cu, st, &d));
    if (d.rec) HIPCHK(c, launch_ctd_apply(c->tv.xt, d, (uint32_t)c->n_cu, st));
    return VGMI_OK;
}

int xt_clamp_if_due(vgmi_ctx* c, size_t n_bytes, hipStream_t st)
{
    bool due = false;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->xt_bytes_since_clamp += n_bytes;
        if (c->xt_bytes_since_clamp >= (1ULL << 31)) {
            c->xt_bytes_since_clamp = 0;
            due = true;
        }
    }
    if (due) HIPCHK(c, launch_xclamp(c->tv.xt, c->xt_n_counts, st));
    return VGMI_OK;
}

}  // namespace

// the ablation knob: honoured by -DVGMI_ABLATION builds only; anywhere else a set VGMI_DBG is reported once and ignored
uint32_t vgmi_dbg_env()
{
    static const uint32_t v = [] {
        const char* d = getenv("VGMI_DBG");
        const uint32_t x = d ? (uint32_t)atoi(d) : 0u;
#ifdef VGMI_ABLATION
        if (x) fprintf(stderr, "[vgmi] WARNING: VGMI_DBG=%u in an ablation build: counters and genotypes are WRONG on purpose\n", x);
        return x;
#else
        if (x) fprintf(stderr, "[vgmi] VGMI_DBG=%u ignored: this library was built without -DVGMI_ABLATION\n", x);
        return 0u;
#endif
    }();
    return v;
}

namespace {
RowParams row_params(vgmi_ctx* c, const char* d_bases, size_t n_bytes, uint32_t k)
{
    RowParams p{};
    p.bases = reinterpret_cast<const uint8_t*>(d_bases);
    p.n_bytes = n_bytes;
    p.k = k;
    p.status = c->d_status;
    p.table = c->tv;
    p.keys_out = nullptr;
    p.dbg = vgmi_dbg_env();
    static const uint32_t l1_min = [] {
        const char* e = getenv("VGMI_L1_MIN");
        const int v = e ? atoi(e) : 60;      // measured (C2, kernel ms per 1e8 reads): 8 4.93, 16 4.67, 24 4.44, 32 4.35, 40 4.29, 48 4.23, 56 4.23
        return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v);
    }();
    p.l1_min = l1_min;
    p.bloom = c->bv;
    return p;
}

// launch geometry of the row kernel
void rows_geometry(vgmi_ctx* c, bool flds, uint32_t& grid, uint32_t& block)
{
    if (flds) { block = 1024; grid = (uint32_t)c->n_cu; }
    else      { block = 256;  grid = (uint32_t)c->n_cu * 8; }
}

// n_bytes_dev != nullptr: the block's length lives in device memory (device-side FASTQ parser); n_bytes is then only an
// upper bound and the kernels derive their geometry themselves (odd k only)
int launch_count(vgmi_ctx* c, const char* d_bases, size_t n_bytes, const uint64_t* d_read_off, size_t n_reads,
                 hipStream_t st, const unsigned long long* n_bytes_dev = nullptr)
{
    if (n_bytes == 0) return VGMI_OK;
    const uint32_t k = c->hdr.k;
    RowParams p = row_params(c, d_bases, n_bytes, k);
    p.n_bytes_dev = n_bytes_dev;
    hipEvent_t e0, e1;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        e0 = get_event(c);
        e1 = get_event(c);
    }
    if (!e0 || !e1) return fail(c, VGMI_E_HIP, "hipEventCreate failed");
    HIPCHK(c, hipEventRecord(e0, st));
    if (n_bytes_dev) {
        if (!(k & 1)) return fail(c, VGMI_E_INVALID, "device-side block length: odd k only");
        uint32_t grid, block;
        rows_geometry(c, c->filter_in_lds, grid, block);
        if ((c->tv.xt.lines || c->tv.xt.cb) && !c->force_generic) {
            int rcx = xt_clamp_if_due(c, n_bytes, st);
            if (rcx) return rcx;
            if (c->tv.xt.cb) { int rcc = launch_ctable_count(c, p, n_bytes, st); if (rcc) return rcc; }
            else HIPCHK(c, launch_count27x(p, c->tv.xt, (uint32_t)c->n_cu * 8, st));
            p.tail27 = 2;
            HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, 1, block, st));
        } else if ((c->fast27_small || (c->fastk_small && c->tv.pt.index)) && !c->force_generic) {
            HIPCHK(c, launch_count27s(p, (uint32_t)c->n_cu, st));
            p.tail27 = c->fast27_small ? 3 : 4;
            HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, 1, block, st));
        } else if (c->fast27 && !c->force_generic) {
            uint32_t g27, b27;
            if (c->fast27_lds) { b27 = 1024; g27 = (uint32_t)c->n_cu; }
            else { b27 = 256; g27 = (uint32_t)c->n_cu * (c->wgs_per_cu ? c->wgs_per_cu : 4); }
            HIPCHK(c, launch_count27(c->fast27_lds, p, g27, b27, st));
            p.tail27 = 1;
            HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, 1, block, st));
        } else {
            HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, grid, block, st));
        }
    } else if (k & 1) {
        uint32_t grid, block;
        rows_geometry(c, c->filter_in_lds, grid, block);
        if ((c->tv.xt.lines || c->tv.xt.cb) && !c->force_generic) {
            // context table / grid-16-mer table: complete 768-byte rows -> count27c_kernel / count27x_kernel, the ends behind them -> the generic kernel
            const uint64_t rows = n_bytes / 768;
            uint64_t emit_from = 0;
            int rcx = xt_clamp_if_due(c, n_bytes, st);
            if (rcx) return rcx;
            if (rows) {
                if (c->tv.xt.cb) { int rcc = launch_ctable_count(c, p, n_bytes, st); if (rcc) return rcc; }
                else HIPCHK(c, launch_count27x(p, c->tv.xt, (uint32_t)c->n_cu * 8, st));
                emit_from = rows * 768 - 1;
            }
            if (emit_from < n_bytes) {
                p.emit_from = emit_from;
                p.row_begin = emit_from >> 10;
                HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, 1, block, st));
            }
        } else if ((c->fast27_small || (c->fastk_small && c->tv.pt.index)) && !c->force_generic) {
            // complete pairs of 1 024-byte rows -> count27s_kernel: lane L of a row covers the k-mers ending at stream positions
            // 16 L - 1 .. 16 L + 14 of the row (k < 27, the grid of 8: 16 L .. 16 L + 15); the generic kernel takes the ends behind
            // the last pair, from the last position of the last full row on (k < 27: from the first position behind it)
            p.row_end = (n_bytes / 2048) * 2;
            uint64_t emit_from = 0;
            if (p.row_end) {
                HIPCHK(c, launch_count27s(p, (uint32_t)c->n_cu, st));
                emit_from = p.row_end * 1024 - (c->fast27_small ? 1 : 0);
            }
            if (emit_from < n_bytes) {
                p.emit_from = emit_from;
                p.row_begin = emit_from >> 10;
                HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, 1, block, st));
            }
        } else if (c->fast27 && !c->force_generic) {
            // complete 768-byte rows -> fast kernel.  It covers every k-mer whose run starts at one of its
            // grid positions; the generic kernel takes the ends after that: the ragged tail plus the last
            // position of the last full row.
            p.row_end = (n_bytes / 1536) * 2;   // VG_ROW27: count27_kernel walks complete 768-byte rows, two per iteration
            uint64_t emit_from = 0;
            if (p.row_end) {
                if (c->fast27_lds) { block = 1024; grid = (uint32_t)c->n_cu; }
                else {
                    // global-filter variant: random-access bound; more than ~16 waves per CU only adds
                    // L2 thrash
                    block = 256;
                    grid = (uint32_t)c->n_cu * (c->wgs_per_cu ? c->wgs_per_cu : 4);
                }
                HIPCHK(c, launch_count27(c->fast27_lds, p, grid, block, st));
                // lane L of a row covers the k-mers ending at stream positions 12L - 1 .. 12L + 10
                emit_from = p.row_end * 768 - 1;
            }
            if (emit_from < n_bytes) {
                p.emit_from = emit_from;
                p.row_begin = emit_from >> 10;
                rows_geometry(c, c->filter_in_lds, grid, block);
                HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, 1, block, st));
            }
        } else {
            HIPCHK(c, launch_rows(K_MODE_COUNT, c->filter_in_lds, p, grid, block, st));
        }
    } else {
        if (!d_read_off) return fail(c, VGMI_E_INVALID, "even k needs read offsets");
        if (c->tv.xt.cb && !c->force_generic && n_bytes >= 768) {
            // k = 20 .. 24 on a large graph: complete 768-byte rows through countkc_kernel<K> over the context table, the debit pass ahead of it,
            // the literal state machine for the ends behind the rows (its lookups go through the same table: table_count)
            const uint64_t rows = n_bytes / 768;
            p.emit_from = rows * 768 - 1;
            int rcx = xt_clamp_if_due(c, n_bytes, st);
            if (rcx) return rcx;
            unsigned long long* list = nullptr;
            int rcl = debit_list_of(c, st, &list);
            if (rcl) return rcl;
            HIPCHK(c, launch_even_debit(p, d_read_off, n_reads, list, VG_DEBIT_LIST, st));
            { int rcc = launch_ctable_count(c, p, n_bytes, st); if (rcc) return rcc; }
            HIPCHK(c, launch_seq(K_MODE_COUNT, p, d_read_off, n_reads, st));
        } else if (c->fastk_small && c->tv.pt.index && !c->force_generic && n_bytes >= 2048) {
            // k = 20 .. 24 on a small graph: the windows of k bases through count27s_kernel<true, K> (complete pairs of rows), in front of
            // it the pass that takes back what the reference's run counter suppresses, behind it the literal state machine for the
            // ends the rows do not cover
            p.row_end = (n_bytes / 2048) * 2;
            p.emit_from = p.row_end * 1024;
            unsigned long long* list = nullptr;
            int rcl = debit_list_of(c, st, &list);
            if (rcl) return rcl;
            HIPCHK(c, launch_even_debit(p, d_read_off, n_reads, list, VG_DEBIT_LIST, st));
            HIPCHK(c, launch_count27s(p, (uint32_t)c->n_cu, st));
            if (p.emit_from < n_bytes) HIPCHK(c, launch_seq(K_MODE_COUNT, p, d_read_off, n_reads, st));
        } else {
            HIPCHK(c, launch_seq(K_MODE_COUNT, p, d_read_off, n_reads, st));
        }
    }
    HIPCHK(c, hipEventRecord(e1, st));
    std::lock_guard<std::mutex> lk(c->mu);
    c->timed.emplace_back(e0, e1);
    c->launches++;
    return VGMI_OK;
}

int collect_timing(vgmi_ctx* c)
{
    for (auto& pr : c->timed) {
        float ms = 0.f;
        HIPCHK(c, hipEventSynchronize(pr.second));
        HIPCHK(c, hipEventElapsedTime(&ms, pr.first, pr.second));
        c->kernel_ms += ms;
        c->event_pool.push_back(pr.first);
        c->event_pool.push_back(pr.second);
    }
    c->timed.clear();
    return VGMI_OK;
}

int ensure_stage(vgmi_ctx* c, Stage& s)
{
    if (s.h) return VGMI_OK;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&s.h), c->buffer_bytes, hipHostMallocDefault));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&s.d), c->buffer_bytes + 16));
    HIPCHK(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    return VGMI_OK;
}

int sync_stages(vgmi_ctx* c)
{
    for (auto& s : c->stage) {
        if (s.busy) {
            HIPCHK(c, hipEventSynchronize(s.done));
            s.busy = false;
        }
    }
    return VGMI_OK;
}

std::vector<uint64_t> offsets_from_newlines(const char* b, size_t n)
{
    std::vector<uint64_t> off;
    off.push_back(0);
    const char* p = b;
    const char* end = b + n;
    while (p < end) {
        const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        if (!nl) { off.push_back(n + 1); break; }  // unterminated last read: pretend a '\n' follows
        off.push_back((uint64_t)(nl - b) + 1);
        p = nl + 1;
    }
    return off;
}

int check_status(vgmi_ctx* c)
{
    uint32_t st = 0;
    HIPCHK(c, hipMemcpy(&st, c->d_status, 4, hipMemcpyDeviceToHost));
    if (st & 2u) return fail(c, VGMI_E_BAD_KEY, "table key with low byte != k or payload >= 2^(2k)");
    if (st & 4u) return fail(c, VGMI_E_DUPLICATE_KEY, "duplicate key in table upload");
    if (st & 1u) return fail(c, VGMI_E_EMPTY_READ, "zero-length read in a read block (reference: assert(len > 0), kmer.cpp:124)");
    return VGMI_OK;
}

}  // namespace

extern "C" {

int vgmi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vgmi_create(int device, size_t buffer_mib, vgmi_ctx** out)
{
    if (!out) return fail(nullptr, VGMI_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, VGMI_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(nullptr, VGMI_E_NO_DEVICE, "device ordinal out of range");
    vgmi_ctx* c = new (std::nothrow) vgmi_ctx();
    if (!c) return fail(nullptr, VGMI_E_NOMEM, "out of host memory");
    c->device = device;
    if (buffer_mib == 0) buffer_mib = 100;  // reference default --buffer 100 (include/varigraph.cuh:28)
    c->buffer_bytes = buffer_mib << 20;
    auto bail = [&](const char* what, hipError_t e) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        vgmi_destroy(c);
        return (int)VGMI_E_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* g = getenv("VGMI_GENERIC_KERNEL")) c->force_generic = g[0] == '1';
    if (const char* g = getenv("VGMI_WGS_PER_CU")) c->wgs_per_cu = (uint32_t)atoi(g);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipMalloc(&c->d_status, 4)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(c->d_status, 0, 4)) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipMalloc(&c->d_hist, 256 * 8)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipEventCreateWithFlags(&c->reset_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    *out = c;
    return VGMI_OK;
}

void vgmi_destroy(vgmi_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_table(c);
    free_nodes(c);
    for (auto& s : c->stage) {
        if (s.h) (void)hipHostFree(s.h);
        if (s.d) (void)hipFree(s.d);
        if (s.d_off) (void)hipFree(s.d_off);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (vgmi_fastq* f : c->fastq_pool) fastq_free(f);
    c->fastq_pool.clear();
    for (auto& pr : c->timed) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& e : c->event_pool) (void)hipEventDestroy(e);
    if (c->bv.filter) (void)hipFree(c->bv.filter);
    if (c->bb_scratch) (void)hipFree(c->bb_scratch);
    c->bb_scratch = nullptr;
    c->bb_cap = 0;
    if (c->d_status) (void)hipFree(c->d_status);
    for (auto& kv : c->debit_lists) (void)hipFree(kv.second);
    for (auto& kv : c->ctd_scratch) (void)hipFree(kv.second.first);
    if (c->d_hist) (void)hipFree(c->d_hist);
    if (c->reset_done) (void)hipEventDestroy(c->reset_done);
    for (auto& b : c->hmm_blocks) (void)hipFree(b.first);
    if (c->d_hmm_entries) (void)hipFree(c->d_hmm_entries);
    if (c->d_hmm_cov) (void)hipFree(c->d_hmm_cov);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* vgmi_last_error(const vgmi_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

void* vgmi_stream(vgmi_ctx* c) { return c ? (void*)c->stream : nullptr; }

/* ---------------------------------------------------------------- table */

int vgmi_table_upload(vgmi_ctx* c, const uint64_t* keys, size_t n_keys, uint32_t k)
{
    if (!c) return VGMI_E_INVALID;
    if (k < 1 || k > 28) return fail(c, VGMI_E_INVALID, "k must be in 1..28 (reference assert, kmer.cpp:124)");
    if (n_keys && !keys) return fail(c, VGMI_E_INVALID, "keys is NULL");
    if (n_keys >= (1ULL << 31)) return fail(c, VGMI_E_INVALID, "too many keys");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_table(c);
    layout_image(c->hdr, k, n_keys);
    c->image_bytes = c->hdr.total_bytes;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_image), c->image_bytes));
    HIPCHK(c, hipMemsetAsync(c->d_image, 0, c->image_bytes, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_image, &c->hdr, sizeof c->hdr, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4, c->stream));
    int rc = adopt_image(c);
    if (rc) return rc;
    HIPCHK(c, launch_table_clear(c->tv, c->stream));
    uint64_t* d_keys = nullptr;
    if (n_keys) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_keys), n_keys * 8));
        HIPCHK(c, hipMemcpyAsync(d_keys, keys, n_keys * 8, hipMemcpyHostToDevice, c->stream));
        hipError_t e = launch_table_insert(c->tv, d_keys, n_keys, k, c->d_key_slot,
                                           const_cast<uint32_t*>(c->tv.filter), const_cast<uint32_t*>(c->tv.grid),
                                           c->hdr.grid_mer == 12, c->d_status, c->stream);
        if (e != hipSuccess) { (void)hipFree(d_keys); HIPCHK(c, e); }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (d_keys) (void)hipFree(d_keys);
    HIPCHK(c, e);
    rc = check_status(c);
    if (rc) { free_table(c); return rc; }
    rc = build_xtable(c);
    if (rc) { free_table(c); return rc; }
    rc = build_ptable(c);
    if (rc) { free_table(c); return rc; }
    c->read_base = 0;
    return VGMI_OK;
}

// Batched exact lookup: index_out[i] = the index keys[i] has in the uploaded key array, 0xFFFFFFFF when the table does not hold it
// (a key of another k included).  Works on a stream and buffers of its own and only reads the table, so it may run while another
// thread counts reads on the same context.
int vgmi_table_lookup(vgmi_ctx* c, const uint64_t* keys, size_t n, uint32_t* index_out)
{
    if (!c || (n && (!keys || !index_out))) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (n == 0) return VGMI_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const TableView tv = c->tv;
    const uint64_t cap = c->hdr.cap, n_keys = c->hdr.n_keys;
    const size_t chunk = std::min<size_t>(n, (size_t)1 << 25);      // 256 MiB of keys per round, two rounds in flight
    hipStream_t st = nullptr;
    uint32_t* key_of_slot = nullptr;
    uint64_t* d_keys[2] = {nullptr, nullptr};
    uint32_t* d_out[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (he == hipSuccess && tv.slots8) {
        he = hipMalloc(reinterpret_cast<void**>(&key_of_slot), cap * 4);
        if (he == hipSuccess) he = hipMemsetAsync(key_of_slot, 0xFF, cap * 4, st);
        if (he == hipSuccess) he = launch_table_key_of_slot(c->d_key_slot, n_keys, key_of_slot, st);
    }
    for (int b = 0; b < 2 && he == hipSuccess; ++b) {
        he = hipMalloc(reinterpret_cast<void**>(&d_keys[b]), chunk * 8);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_out[b]), chunk * 4);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
    }
    size_t at[2] = {0, 0}, len[2] = {0, 0};
    auto collect = [&](int b) {
        if (he != hipSuccess || !len[b]) return;
        he = hipEventSynchronize(done[b]);
        len[b] = 0;
    };
    int b = 0;
    for (size_t off = 0; off < n && he == hipSuccess; off += chunk, b ^= 1) {
        collect(b);
        if (he != hipSuccess) break;
        at[b] = off;
        len[b] = std::min(chunk, n - off);
        he = hipMemcpyAsync(d_keys[b], keys + off, len[b] * 8, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = launch_table_lookup(tv, d_keys[b], len[b], c->hdr.k, key_of_slot, d_out[b], st);
        if (he == hipSuccess) he = hipMemcpyAsync(index_out + off, d_out[b], len[b] * 4, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipEventRecord(done[b], st);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    for (int q = 0; q < 2; ++q) {
        if (d_keys[q]) (void)hipFree(d_keys[q]);
        if (d_out[q]) (void)hipFree(d_out[q]);
        if (done[q]) (void)hipEventDestroy(done[q]);
    }
    if (key_of_slot) (void)hipFree(key_of_slot);
    if (st) (void)hipStreamDestroy(st);
    (void)at;
    HIPCHK(c, he);
    return VGMI_OK;
}

int vgmi_table_image_bytes(vgmi_ctx* c, size_t* bytes)
{
    if (!c || !bytes) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    *bytes = c->image_bytes;
    return VGMI_OK;
}

int vgmi_table_export(vgmi_ctx* c, void* dev_dst, size_t bytes)
{
    if (!c || !dev_dst) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (bytes < c->image_bytes) return fail(c, VGMI_E_INVALID, "destination smaller than the image");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dev_dst, c->d_image, c->image_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VGMI_OK;
}

// A copy of the image as it stands (no sample counted yet): vgmi_table_broadcast_comm then sends the copy, so the root may start
// counting -- which sets per-sample bits inside the image -- before the communicator is up.  Freed by the broadcast.
int vgmi_table_snapshot(vgmi_ctx* c)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->d_snapshot) (void)hipFree(c->d_snapshot);
    c->d_snapshot = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_snapshot), c->image_bytes));
    HIPCHK(c, hipMemcpyAsync(c->d_snapshot, c->d_image, c->image_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VGMI_OK;
}

int vgmi_table_import(vgmi_ctx* c, const void* dev_src, size_t bytes)
{
    if (!c || !dev_src) return VGMI_E_INVALID;
    if (bytes < sizeof(ImageHeader)) return fail(c, VGMI_E_INVALID, "image too small");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ImageHeader h;
    HIPCHK(c, hipMemcpy(&h, dev_src, sizeof h, hipMemcpyDeviceToHost));
    if (memcmp(h.magic, "VGMITBL1", 8) != 0 || h.total_bytes > bytes || h.k < 1 || h.k > 28)
        return fail(c, VGMI_E_INVALID, "not a table image");
    ImageHeader chk;
    layout_image(chk, h.k, h.n_keys);
    if (memcmp(&chk, &h, sizeof h) != 0) return fail(c, VGMI_E_INVALID, "table image layout mismatch");
    free_table(c);
    c->hdr = h;
    c->image_bytes = h.total_bytes;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_image), c->image_bytes));
    HIPCHK(c, hipMemcpy(c->d_image, dev_src, c->image_bytes, hipMemcpyDeviceToDevice));
    int rc = adopt_image(c);
    if (rc) return rc;
    HIPCHK(c, launch_counts_reset(c->tv, c->stream));   // the exporter's per-sample state travels with the image
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = build_xtable(c);
    if (rc == VGMI_OK) rc = build_ptable(c);
    if (rc) {
        free_table(c);      // as vgmi_table_upload does: no half-built table behind an error code
        return rc;
    }
    c->read_base = 0;
    return VGMI_OK;
}

// ---- the table image over RCCL, for one process per GPU (the north_star's "single RCCL broadcast of the read-only graph index over
// xGMI"; the reference is single-device: main.cu:221,444 select one).  librccl is loaded on first use: the library itself carries
// no dependency on it, a node without RCCL still runs everything else.
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, ncclUniqueIdBytes, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl* rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) {
            const char* m = dlerror();      // (once: the call hands the message over and clears it)
            x.err = std::string("librccl.so: ") + (m ? m : "not found");
            return x;
        }
        x.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<int (*)(void**, int, ncclUniqueIdBytes, int)>(dlsym(x.lib, "ncclCommInitRank"));
        x.Broadcast = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(x.lib, "ncclBroadcast"));
        x.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(x.lib, "ncclAllReduce"));
        x.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(x.lib, "ncclCommDestroy"));
        x.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(x.lib, "ncclGetErrorString"));
        if (!x.GetUniqueId || !x.CommInitRank || !x.Broadcast || !x.AllReduce || !x.CommDestroy) x.err = "librccl.so lacks an entry point";
        return x;
    }();
    return &r;
}
}  // namespace

int vgmi_rccl_unique_id(void* id128)
{
    if (!id128) return VGMI_E_INVALID;
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, VGMI_E_STATE, r->err);
    const int rc = r->GetUniqueId(id128);
    if (rc) return fail(nullptr, VGMI_E_HIP, std::string("ncclGetUniqueId: ") + (r->GetErrorString ? r->GetErrorString(rc) : "failed"));
    return VGMI_OK;
}

// The communicator on its own: ncclCommInitRank takes seconds (topology, kernels of every rank's device) and needs neither a
// table nor a context -- a rank calls it beside its graph load / table build, and joins the broadcast when both are there.
struct vgmi_comm {
    void* comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

int vgmi_comm_create(int device, int rank, int world, const void* id128, vgmi_comm** out)
{
    if (!out) return VGMI_E_INVALID;
    *out = nullptr;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return VGMI_E_INVALID;
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, VGMI_E_STATE, r->err);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VGMI_E_HIP, "vgmi_comm_create: hipSetDevice failed");
    ncclUniqueIdBytes id;
    memcpy(id.internal, id128, sizeof id.internal);
    vgmi_comm* m = new (std::nothrow) vgmi_comm();
    if (!m) return fail(nullptr, VGMI_E_NOMEM, "out of memory");
    m->device = device;
    m->rank = rank;
    m->world = world;
    const int rc = r->CommInitRank(&m->comm, world, id, rank);
    if (rc) {
        delete m;
        return fail(nullptr, VGMI_E_HIP, std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "failed"));
    }
    *out = m;
    return VGMI_OK;
}

void vgmi_comm_destroy(vgmi_comm* m)
{
    if (!m) return;
    if (m->comm) {
        (void)hipSetDevice(m->device);
        (void)rccl()->CommDestroy(m->comm);
    }
    delete m;
}

// Root = rank 0.  Every rank goes through the same three collectives whatever happens on its side -- the image's size (0: the
// root has none to give), an agreement that every receiver has its buffer (all-reduce, minimum), the image -- so that a rank that
// cannot go on says so to the others instead of leaving them inside a collective.
int vgmi_table_broadcast_comm(vgmi_ctx* c, vgmi_comm* m)
{
    if (!c || !m || !m->comm) return VGMI_E_INVALID;
    if (m->device != c->device) return fail(c, VGMI_E_INVALID, "vgmi_table_broadcast_comm: the communicator is on another device than the context");
    Rccl* r = rccl();
    HIPCHK(c, hipSetDevice(c->device));
    const int rank = m->rank;
    // a stream of its own: a root that sends a snapshot may be counting on the context's streams meanwhile
    hipStream_t st = nullptr;
    HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long* d_n = nullptr;      // [0] the size, [1] the receivers' agreement
    uint8_t* d_recv = nullptr;
    struct Cleanup {
        hipStream_t& st; unsigned long long*& d_n; uint8_t*& d_recv;
        ~Cleanup() { if (d_n) (void)hipFree(d_n); if (d_recv) (void)hipFree(d_recv); if (st) (void)hipStreamDestroy(st); }
    } cleanup{st, d_n, d_recv};
    if (!(rank == 0 && c->d_snapshot)) HIPCHK(c, hipStreamSynchronize(c->stream));
    auto nccl_text = [&](const char* what, int rc) { return std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "failed"); };
    // (16 bytes: if even that fails the device is gone, and so is this rank's part in the collectives)
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_n), 16));
    unsigned long long h[2] = {rank == 0 && c->has_table ? (unsigned long long)c->image_bytes : 0ull, 1ull};
    HIPCHK(c, hipMemcpy(d_n, h, 16, hipMemcpyHostToDevice));
    int rc = r->Broadcast(d_n, d_n, 8, /* ncclChar */ 0, 0, m->comm, st);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    if (rc) return fail(c, VGMI_E_HIP, nccl_text("ncclBroadcast (size)", rc));
    HIPCHK(c, hipMemcpy(h, d_n, 8, hipMemcpyDeviceToHost));
    const unsigned long long n = h[0];
    if (n == 0) return fail(c, VGMI_E_STATE, "the root has no table to broadcast");
    uint8_t* d_buf = rank == 0 ? (c->d_snapshot ? c->d_snapshot : c->d_image) : nullptr;
    if (rank != 0) {
        if (hipMalloc(reinterpret_cast<void**>(&d_recv), n) != hipSuccess) {
            d_recv = nullptr;
            h[1] = 0;
            (void)hipGetLastError();
            HIPCHK(c, hipMemcpy(d_n + 1, h + 1, 8, hipMemcpyHostToDevice));
        }
        d_buf = d_recv;
    }
    rc = r->AllReduce(d_n + 1, d_n + 1, 1, /* ncclUint64 */ 5, /* ncclMin */ 3, m->comm, st);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    unsigned long long all_ready = 0;
    if (rc == 0 && hipMemcpy(&all_ready, d_n + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    if (rc) return fail(c, VGMI_E_HIP, nccl_text("ncclAllReduce (buffers)", rc));
    if (!all_ready) return fail(c, VGMI_E_NOMEM, "vgmi_table_broadcast_comm: a rank has no room for the table image");
    rc = r->Broadcast(d_buf, d_buf, n, /* ncclChar */ 0, 0, m->comm, st);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    if (rc) return fail(c, VGMI_E_HIP, nccl_text("ncclBroadcast (image)", rc));
    if (rank == 0 && c->d_snapshot) {
        (void)hipFree(c->d_snapshot);
        c->d_snapshot = nullptr;
    }
    return rank != 0 ? vgmi_table_import(c, d_recv, n) : VGMI_OK;
}

int vgmi_table_broadcast(vgmi_ctx* c, int rank, int world, const void* id128)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return VGMI_E_INVALID;
    vgmi_comm* m = nullptr;
    const int rc = vgmi_comm_create(c->device, rank, world, id128, &m);
    if (rc) return fail(c, rc, vgmi_last_error(nullptr));
    const int out = vgmi_table_broadcast_comm(c, m);
    vgmi_comm_destroy(m);
    return out;
}

int vgmi_table_clone(vgmi_ctx* dst, vgmi_ctx* src)
{
    if (!dst || !src || dst == src) return VGMI_E_INVALID;
    if (!src->has_table) return fail(dst, VGMI_E_STATE, "the source context has no table");
    HIPCHK(dst, hipSetDevice(src->device));
    HIPCHK(dst, hipStreamSynchronize(src->stream));
    HIPCHK(dst, hipSetDevice(dst->device));
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    free_table(dst);
    dst->hdr = src->hdr;
    dst->image_bytes = src->image_bytes;
    HIPCHK(dst, hipMalloc(reinterpret_cast<void**>(&dst->d_image), dst->image_bytes));
    if (dst->device == src->device) {
        HIPCHK(dst, hipMemcpy(dst->d_image, src->d_image, dst->image_bytes, hipMemcpyDeviceToDevice));
    } else {
        // one device-to-device transfer over xGMI (peer access when the link allows it, the runtime stages otherwise)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, src->device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIPCHK(dst, e);
            (void)hipGetLastError();
        }
        HIPCHK(dst, hipMemcpyPeer(dst->d_image, dst->device, src->d_image, src->device, dst->image_bytes));
    }
    int rc = adopt_image(dst);
    if (rc) return rc;
    HIPCHK(dst, launch_counts_reset(dst->tv, dst->stream));   // the source's per-sample state travels with the image
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    rc = build_xtable(dst);
    if (rc == VGMI_OK) rc = build_ptable(dst);
    if (rc) {
        free_table(dst);
        return rc;
    }
    dst->read_base = 0;
    return VGMI_OK;
}

int vgmi_xtable_info(vgmi_ctx* c, size_t* n_lines, size_t* overflow_pairs)
{
    if (!c) return VGMI_E_INVALID;
    if (n_lines) *n_lines = c->tv.xt.lines ? c->tv.xt.n_lines : 0;
    if (overflow_pairs) *overflow_pairs = c->tv.xt.lines ? c->xt_over_keys : 0;
    return VGMI_OK;
}

int vgmi_ctable_info(vgmi_ctx* c, size_t* n_buckets, size_t* n_entries, size_t* n_unitigs, size_t* moved_entries, size_t* overflow_kmers)
{
    if (!c) return VGMI_E_INVALID;
    const bool on = c->tv.xt.cb != nullptr;
    if (n_buckets) *n_buckets = on ? c->tv.xt.n_buckets : 0;
    if (n_entries) *n_entries = on ? c->ct_entries : 0;
    if (n_unitigs) *n_unitigs = on ? c->ct_unitigs : 0;
    if (moved_entries) *moved_entries = on ? c->ct_moved : 0;
    if (overflow_kmers) *overflow_kmers = on ? c->xt_over_keys : 0;
    return VGMI_OK;
}

int vgmi_table_info(vgmi_ctx* c, size_t* n_keys, uint32_t* k, size_t* n_slots, size_t* filter_bits)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (n_keys) *n_keys = c->hdr.n_keys;
    if (k) *k = c->hdr.k;
    if (n_slots) *n_slots = c->hdr.cap;
    if (filter_bits) *filter_bits = 32ULL << c->hdr.filter_words_log2;
    return VGMI_OK;
}

int vgmi_nodes_upload(vgmi_ctx* c, const uint64_t* node_off, const uint32_t* key_index, size_t n_nodes)
{
    if (!c || !node_off) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "upload the table first");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t total = node_off[n_nodes];
    if (total && !key_index) return fail(c, VGMI_E_INVALID, "key_index is NULL");
    for (uint64_t i = 0; i < total; ++i)
        if (key_index[i] >= c->hdr.n_keys) return fail(c, VGMI_E_INVALID, "node key index out of range");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_nodes(c);
    c->n_nodes = n_nodes;
    c->n_node_entries = total;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_node_key_index), (total ? total : 1) * 4));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_cov_node), total ? total : 1));
    if (total) HIPCHK(c, hipMemcpy(c->d_node_key_index, key_index, total * 4, hipMemcpyHostToDevice));
    return VGMI_OK;
}

int vgmi_flags_upload(vgmi_ctx* c, const uint8_t* flag)
{
    if (!c || !flag) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "upload the table first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->d_flag) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_flag), c->hdr.n_keys ? c->hdr.n_keys : 1));
    if (c->hdr.n_keys) HIPCHK(c, hipMemcpy(c->d_flag, flag, c->hdr.n_keys, hipMemcpyHostToDevice));
    return VGMI_OK;
}

/* ---------------------------------------------------------------- per sample */

int vgmi_counts_reset(vgmi_ctx* c)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (c->open_fastq) return fail(c, VGMI_E_STATE, "close the FASTQ streams first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = sync_stages(c);
    if (rc) return rc;
    rc = collect_timing(c);
    if (rc) return rc;
    if (c->tv.xt.counts) HIPCHK(c, hipMemsetAsync(c->d_xt_counts, 0, c->xt_n_counts * 4, c->stream));
    if (c->d_counts) HIPCHK(c, hipMemsetAsync(c->d_counts, 0, c->n_counts * 4, c->stream));
    if (!c->tv.xt.counts && (!c->d_counts || c->tv.slots8)) HIPCHK(c, launch_counts_reset(c->tv, c->stream));   // in-slot counters / saturation flags
    if (c->tv.pt.SB) HIPCHK(c, hipMemsetAsync(c->d_pt_SB, 0, c->pt_sb_bytes, c->stream));                       // ... and their copies in the path table
    HIPCHK(c, hipMemsetAsync(c->d_status, 0, 4, c->stream));
    // host blocks are counted on the stages' own (non-blocking) streams: their next launch waits for this reset
    HIPCHK(c, hipEventRecord(c->reset_done, c->stream));
    for (auto& s : c->stage) s.after_reset = true;
    c->read_base = 0;
    c->kernel_ms = 0.f;
    c->launches = 0;
    return VGMI_OK;
}

int vgmi_reads_submit_device(vgmi_ctx* c, const char* dev_bases, size_t n_bytes, const uint64_t* dev_read_off,
                             size_t n_reads)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (n_bytes && !dev_bases) return fail(c, VGMI_E_INVALID, "dev_bases is NULL");
    if (reinterpret_cast<uintptr_t>(dev_bases) & 15) return fail(c, VGMI_E_INVALID, "dev_bases must be 16-byte aligned");
    if (n_reads > n_bytes) return fail(c, VGMI_E_INVALID, "n_reads > n_bytes");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = launch_count(c, dev_bases, n_bytes, dev_read_off, n_reads, c->stream);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    c->read_base += n_bytes - n_reads;
    return VGMI_OK;
}

int vgmi_reads_submit(vgmi_ctx* c, const char* bases, size_t n_bytes, const uint64_t* read_off, size_t n_reads)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (n_bytes == 0) return VGMI_OK;
    if (!bases) return fail(c, VGMI_E_INVALID, "bases is NULL");
    if (bases[n_bytes - 1] != '\n') return fail(c, VGMI_E_INVALID, "a read block must end with '\\n'");
    if (n_reads > n_bytes) return fail(c, VGMI_E_INVALID, "n_reads > n_bytes");
    HIPCHK(c, hipSetDevice(c->device));
    const bool need_off = (c->hdr.k & 1) == 0;
    std::vector<uint64_t> derived;
    if (need_off && !read_off) {
        derived = offsets_from_newlines(bases, n_bytes);
        if (derived.size() != n_reads + 1) return fail(c, VGMI_E_INVALID, "n_reads does not match the number of '\\n'");
        read_off = derived.data();
    }
    // cut the block into staging-buffer sized pieces at read boundaries
    size_t pos = 0, read_i = 0;
    while (pos < n_bytes) {
        size_t len = n_bytes - pos;
        size_t piece_reads = n_reads - read_i;
        if (len > c->buffer_bytes) {
            len = c->buffer_bytes;
            while (len > 0 && bases[pos + len - 1] != '\n') --len;
            if (len == 0) return fail(c, VGMI_E_INVALID, "a single read exceeds the staging buffer (--buffer)");
            if (need_off) {
                size_t j = read_i;
                while (read_off[j] < pos + len) ++j;
                piece_reads = j - read_i;
            } else {
                piece_reads = 0;  // not needed by the odd-k kernel
            }
        }
        Stage& s = c->stage[c->next_stage];
        c->next_stage ^= 1;
        int rc = ensure_stage(c, s);
        if (rc) return rc;
        if (s.busy) { HIPCHK(c, hipEventSynchronize(s.done)); s.busy = false; }
        if (s.after_reset) { HIPCHK(c, hipStreamWaitEvent(s.stream, c->reset_done, 0)); s.after_reset = false; }
        memcpy(s.h, bases + pos, len);
        HIPCHK(c, hipMemcpyAsync(s.d, s.h, len, hipMemcpyHostToDevice, s.stream));
        const uint64_t* d_off = nullptr;
        if (need_off) {
            if (s.d_off_cap < piece_reads + 1) {
                if (s.d_off) (void)hipFree(s.d_off);
                s.d_off_cap = (piece_reads + 1) * 2;
                HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&s.d_off), s.d_off_cap * 8));
            }
            std::vector<uint64_t> rel(piece_reads + 1);
            for (size_t j = 0; j <= piece_reads; ++j) rel[j] = read_off[read_i + j] - pos;
            HIPCHK(c, hipMemcpy(s.d_off, rel.data(), rel.size() * 8, hipMemcpyHostToDevice));
            d_off = s.d_off;
        }
        rc = launch_count(c, s.d, len, d_off, piece_reads, s.stream);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(s.done, s.stream));
        s.busy = true;
        pos += len;
        read_i += piece_reads;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    c->read_base += n_bytes - n_reads;
    return VGMI_OK;
}

int vgmi_read_base(vgmi_ctx* c, uint64_t* rb)
{
    if (!c || !rb) return VGMI_E_INVALID;
    *rb = c->read_base;
    return VGMI_OK;
}

static int finish_common(vgmi_ctx* c, uint8_t* d_cov, uint8_t* d_cov_node, unsigned long long* d_hist)
{
    // the main stream must see every staged kernel
    for (auto& s : c->stage)
        if (s.busy) HIPCHK(c, hipStreamWaitEvent(c->stream, s.done, 0));
    if (d_hist) HIPCHK(c, hipMemsetAsync(d_hist, 0, 256 * 8, c->stream));
    if (c->tv.xt.counts) HIPCHK(c, launch_xcov(c->tv.xt, c->d_xt_id, c->hdr.n_keys, c->d_flag, d_cov, d_hist, c->stream));
    else HIPCHK(c, launch_cov(c->tv, c->d_key_slot, c->hdr.n_keys, c->d_flag, d_cov, d_hist, c->stream));
    if (d_cov_node && c->n_node_entries)
        HIPCHK(c, launch_node_gather(d_cov, c->d_node_key_index, c->n_node_entries, d_cov_node, c->stream));
    return VGMI_OK;
}

int vgmi_counts_finish(vgmi_ctx* c, uint8_t* cov_out, uint8_t* cov_node_out, uint64_t* hist_out)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (cov_node_out && !c->d_node_key_index) return fail(c, VGMI_E_STATE, "no nodes uploaded");
    if (hist_out && !c->d_flag) return fail(c, VGMI_E_STATE, "no flags uploaded");
    if (c->open_fastq) return fail(c, VGMI_E_STATE, "close the FASTQ streams first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = finish_common(c, c->d_cov, cov_node_out ? c->d_cov_node : nullptr, hist_out ? c->d_hist : nullptr);
    if (rc) return rc;
    if (cov_out && c->hdr.n_keys)
        HIPCHK(c, hipMemcpyAsync(cov_out, c->d_cov, c->hdr.n_keys, hipMemcpyDeviceToHost, c->stream));
    if (cov_node_out && c->n_node_entries)
        HIPCHK(c, hipMemcpyAsync(cov_node_out, c->d_cov_node, c->n_node_entries, hipMemcpyDeviceToHost, c->stream));
    if (hist_out) HIPCHK(c, hipMemcpyAsync(hist_out, c->d_hist, 256 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = sync_stages(c);
    if (rc) return rc;
    rc = collect_timing(c);
    if (rc) return rc;
    return check_status(c);
}

int vgmi_counts_finish_device(vgmi_ctx* c, uint8_t* dev_cov, uint8_t* dev_cov_node, uint64_t* dev_hist)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (dev_cov_node && !c->d_node_key_index) return fail(c, VGMI_E_STATE, "no nodes uploaded");
    if (dev_hist && !c->d_flag) return fail(c, VGMI_E_STATE, "no flags uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = finish_common(c, dev_cov ? dev_cov : c->d_cov, dev_cov_node,
                           reinterpret_cast<unsigned long long*>(dev_hist));
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rc = sync_stages(c);
    if (rc) return rc;
    rc = collect_timing(c);
    if (rc) return rc;
    return check_status(c);
}

static int counts_xfer(vgmi_ctx* c, uint32_t* dev, bool import)
{
    if (!c || !dev) return VGMI_E_INVALID;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (reinterpret_cast<uintptr_t>(dev) & 3) return fail(c, VGMI_E_INVALID, "device pointer must be 4-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    for (auto& s : c->stage)
        if (s.busy) HIPCHK(c, hipStreamWaitEvent(c->stream, s.done, 0));
    if (c->tv.xt.counts) HIPCHK(c, launch_xcounts_xfer(c->tv.xt, c->d_xt_id, dev, c->hdr.n_keys, import, c->stream));
    else HIPCHK(c, launch_counts_xfer(c->tv, c->d_key_slot, dev, c->hdr.n_keys, import, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VGMI_OK;
}

int vgmi_counts_export_device(vgmi_ctx* c, uint32_t* dev_counts_out) { return counts_xfer(c, dev_counts_out, false); }

int vgmi_counts_import_device(vgmi_ctx* c, const uint32_t* dev_counts)
{
    return counts_xfer(c, const_cast<uint32_t*>(dev_counts), true);
}

int vgmi_count_kernel_ms(vgmi_ctx* c, float* ms, uint64_t* launches)
{
    if (!c) return VGMI_E_INVALID;
    int rc = collect_timing(c);
    if (rc) return rc;
    if (ms) *ms = c->kernel_ms;
    if (launches) *launches = c->launches;
    return VGMI_OK;
}

/* ---------------------------------------------------------------- device-side FASTQ parsing */

struct vgmi_fastq {
    vgmi_ctx* c = nullptr;
    hipStream_t stream = nullptr;
    char* h_stage[2] = {nullptr, nullptr};      // pinned
    hipEvent_t h_done[2] = {nullptr, nullptr};  // the H2D copy out of that staging buffer has finished
    bool h_busy[2] = {false, false};
    uint8_t* d_raw[2] = {nullptr, nullptr};
    uint8_t* d_packed = nullptr;
    uint32_t *d_tile = nullptr, *d_nlpos = nullptr, *d_rec = nullptr, *d_off = nullptr, *d_bsum = nullptr;
    FqState* d_state = nullptr;
    size_t cap = 0;          // pinned staging buffers (the context's --buffer size)
    size_t text_cap = 0;     // text per chunk on the device: >= cap
    uint32_t cap_lines = 0, tail_max = 0;
    int next = 0, acquired = -1;
    // block-gzip input inflated on the device (allocated by the first vgmi_fastq_commit_bgzf)
    uint8_t* d_comp = nullptr;
    BgzfMember* d_members = nullptr;
    BgzfMember* h_members[2] = {nullptr, nullptr};   // pinned
    uint32_t* d_status = nullptr;
    uint32_t* d_crc = nullptr;
    BgzfVerdict* d_verdict = nullptr;
    uint32_t max_members = 0;
    uint32_t bgzf_round = 0;            // members the inflate kernel runs at once (0: unknown)
    double bgzf_avg_c = 0, bgzf_avg_u = 0;      // compressed / text bytes per member of the last commit
    // ordinary gzip inflated on the device (vgmi_fastq_commit_gzip): scratch of the pipeline and where the stream stands
    void* gz = nullptr;                              // GzScratch
    bool gz_in_member = false;                       // false: the next staged byte is a member header (or the data is over)
    uint32_t gz_bit = 0;                             // the next block starts this many bits into the first staged byte
    uint32_t gz_avail = 0;                           // text bytes of this member so far, 32768 at most (the window that exists)
    uint32_t gz_skip = 0;                            // bytes of a member's trailer still to come (the front of the next piece)
    unsigned char gz_trailer[8] = {0};               // the trailer as it arrives: CRC-32, ISIZE
    uint64_t gz_member_text = 0;                     // text bytes of the member being decoded
    uint32_t gz_reason = 0;                          // why the device gave the stream up (GzSegOut::status), 0: it did not
    uint64_t gz_text = 0;                            // text bytes the device produced
    std::vector<uint32_t> batch_members;             // members per committed batch
    std::vector<uint64_t> member_size;               // compressed size of every member committed, in stream order
};

namespace {
void gz_scratch_free(void* g);
void fastq_free(vgmi_fastq* f)
{
    if (!f) return;
    gz_scratch_free(f->gz);
    f->gz = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (f->h_stage[i]) (void)hipHostFree(f->h_stage[i]);
        if (f->h_done[i]) (void)hipEventDestroy(f->h_done[i]);
        if (f->d_raw[i]) (void)hipFree(f->d_raw[i]);
    }
    for (void* p : {(void*)f->d_packed, (void*)f->d_tile, (void*)f->d_nlpos, (void*)f->d_rec, (void*)f->d_off, (void*)f->d_bsum,
                    (void*)f->d_state, (void*)f->d_comp, (void*)f->d_members, (void*)f->d_status, (void*)f->d_crc, (void*)f->d_verdict})
        if (p) (void)hipFree(p);
    for (int i = 0; i < 2; ++i)
        if (f->h_members[i]) (void)hipHostFree(f->h_members[i]);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}
}  // namespace

int vgmi_fastq_open(vgmi_ctx* c, vgmi_fastq** out)
{
    if (!c || !out) return VGMI_E_INVALID;
    *out = nullptr;
    if (!c->has_table) return fail(c, VGMI_E_STATE, "no table");
    if (!(c->hdr.k & 1)) return fail(c, VGMI_E_STATE, "the device-side FASTQ parser serves odd k (even k: host reader + vgmi_reads_submit)");
    HIPCHK(c, hipSetDevice(c->device));
    size_t want_cap = c->buffer_bytes < (16u << 20) ? (16u << 20) : (c->buffer_bytes > (1u << 30) ? (1u << 30) : c->buffer_bytes);
    // Text per chunk on the device.  Block-gzip input inflates one member (64 KiB of text) per wavefront: a chunk below
    // 256 MiB leaves wavefront slots empty (measured: 6.4e7 reads/s with 100 MiB chunks, 9.9e7 with 256 MiB), so the
    // device side is sized for that whatever the staging buffers are; plain text arrives in staging-buffer pieces.
    // (round 4: 512 MiB -- an ordinary gzip stream is inflated a stretch of ~160 KB of text per wavefront, and two streams of 256 MiB chunks
    // leave a quarter of the device's wavefront slots empty: 4.2e7 reads/s with 256 MiB chunks, 5.9e7 with 512, 5.7e7 with 1 024)
    size_t want_text = want_cap < ((size_t)512 << 20) ? ((size_t)512 << 20) : want_cap;
    if (const char* e = getenv("VGMI_FASTQ_TEXT_MB"))    // A/B: text per chunk of compressed input (more members / stretches in flight per launch)
        if (atoi(e) >= 256 && atoi(e) <= 4096) want_text = (size_t)atoi(e) << 20;
    if (const char* e = getenv("VGMI_FASTQ_CHUNK_KB"))   // tests: small chunks put every kind of record across a boundary
        if (atoi(e) >= 4) want_text = want_cap = (size_t)atoi(e) << 10;
    {   // a closed stream of the same geometry: its buffers are reused (pinned allocations cost more than a small file)
        vgmi_fastq* r = nullptr;
        {
            std::lock_guard<std::mutex> lk(c->mu);
            for (size_t i = 0; i < c->fastq_pool.size() && !r; ++i)
                if (c->fastq_pool[i]->cap == want_cap && c->fastq_pool[i]->text_cap == want_text) {
                    r = c->fastq_pool[i];
                    c->fastq_pool.erase(c->fastq_pool.begin() + (long)i);
                }
        }
        if (r) {
            r->next = 0;
            r->acquired = -1;
            r->h_busy[0] = r->h_busy[1] = false;
            r->batch_members.clear();
            r->member_size.clear();
            r->gz_in_member = false;
            r->gz_bit = r->gz_avail = r->gz_skip = r->gz_reason = 0;
            r->gz_text = 0;
            if (r->d_verdict) (void)hipMemsetAsync(r->d_verdict, 0xFF, 12, r->stream), (void)hipMemsetAsync(&r->d_verdict->good_bytes, 0, 8, r->stream);
            hipError_t e = launch_fastq_init(r->d_state, r->tail_max, r->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(r->stream, c->reset_done, 0);
            if (e != hipSuccess) {
                fastq_free(r);
                HIPCHK(c, e);
            }
            std::lock_guard<std::mutex> lk(c->mu);
            c->open_fastq++;
            *out = r;
            return VGMI_OK;
        }
    }
    vgmi_fastq* f = new (std::nothrow) vgmi_fastq();
    if (!f) return fail(c, VGMI_E_NOMEM, "out of host memory");
    f->c = c;
    f->cap = want_cap;
    f->text_cap = want_text;
    f->tail_max = 1u << 20;                          // an incomplete record carried between chunks: up to 1 MiB
    f->cap_lines = (uint32_t)((f->text_cap + f->tail_max) / 6);
    const size_t raw_bytes = f->tail_max + f->text_cap + 256;
    const uint32_t n_tiles = (uint32_t)((raw_bytes + 4095) / 4096) + 1;
    const uint32_t cap_rec = f->cap_lines / 4 + 1;
    hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc(reinterpret_cast<void**>(&f->h_stage[i]), f->cap, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->h_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_raw[i]), raw_bytes);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_packed), f->text_cap + f->tail_max + 256);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_tile), (size_t)n_tiles * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_nlpos), (size_t)f->cap_lines * 4 + 64);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_rec), (size_t)cap_rec * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_off), (size_t)cap_rec * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_bsum), (size_t)(cap_rec / 1024 + 2) * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_state), sizeof(FqState));
    if (e == hipSuccess) e = launch_fastq_init(f->d_state, f->tail_max, f->stream);
    // the per-sample reset runs on the context's main stream: this stream starts behind it
    if (e == hipSuccess) e = hipStreamWaitEvent(f->stream, c->reset_done, 0);
    if (e != hipSuccess) {
        fastq_free(f);
        HIPCHK(c, e);
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->open_fastq++;
    }
    *out = f;
    return VGMI_OK;
}

int vgmi_fastq_acquire(vgmi_fastq* f, char** host_buf, size_t* capacity)
{
    if (!f || !host_buf || !capacity) return VGMI_E_INVALID;
    vgmi_ctx* c = f->c;
    if (f->acquired >= 0) return fail(c, VGMI_E_STATE, "commit the buffer acquired before");
    HIPCHK(c, hipSetDevice(c->device));
    const int i = f->next;
    if (f->h_busy[i]) {
        HIPCHK(c, hipEventSynchronize(f->h_done[i]));
        f->h_busy[i] = false;
    }
    f->acquired = i;
    *host_buf = f->h_stage[i];
    *capacity = f->cap;
    return VGMI_OK;
}

int vgmi_fastq_text_capacity(vgmi_fastq* f, size_t* text_bytes)
{
    if (!f || !text_bytes) return VGMI_E_INVALID;
    *text_bytes = f->text_cap;
    return VGMI_OK;
}

int vgmi_fastq_bgzf_want(vgmi_fastq* f, size_t* comp_bytes)
{
    if (!f || !comp_bytes) return VGMI_E_INVALID;
    *comp_bytes = 0;
    if (!f->bgzf_round || f->bgzf_avg_c <= 0 || f->bgzf_avg_u <= 0) return VGMI_OK;
    const double per_round = (double)f->bgzf_round * f->bgzf_avg_u;
    const double rounds = std::floor(0.9 * (double)f->text_cap / per_round);
    if (rounds < 1) return VGMI_OK;
    *comp_bytes = (size_t)(0.99 * rounds * (double)f->bgzf_round * f->bgzf_avg_c);
    return VGMI_OK;
}

int vgmi_fastq_commit(vgmi_fastq* f, size_t n_bytes)
{
    if (!f) return VGMI_E_INVALID;
    vgmi_ctx* c = f->c;
    if (f->acquired < 0) return fail(c, VGMI_E_STATE, "no buffer acquired");
    if (n_bytes > f->cap) return fail(c, VGMI_E_INVALID, "more bytes than the buffer holds");
    const int i = f->acquired;
    f->acquired = -1;
    if (n_bytes == 0) return VGMI_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(f->d_raw[i] + f->tail_max, f->h_stage[i], n_bytes, hipMemcpyHostToDevice, f->stream));
    HIPCHK(c, hipEventRecord(f->h_done[i], f->stream));
    f->h_busy[i] = true;
    FqBuffers b{};
    b.raw = f->d_raw[i];
    b.raw_next = f->d_raw[i ^ 1];
    b.packed = f->d_packed;
    b.tile = f->d_tile;
    b.nlpos = f->d_nlpos;
    b.rec_bytes = f->d_rec;
    b.out_off = f->d_off;
    b.block_sum = f->d_bsum;
    b.state = f->d_state;
    b.cap_lines = f->cap_lines;
    b.tail_max = f->tail_max;
    HIPCHK(c, launch_fastq_chunk(b, (uint32_t)n_bytes, f->stream));
    // the read block's length is on the device: the count kernels fetch it (upper bound here: tail + chunk)
    int rc = launch_count(c, reinterpret_cast<const char*>(f->d_packed), f->tail_max + n_bytes, nullptr, 0, f->stream,
                          &f->d_state->packed_bytes);
    if (rc) return rc;
    f->next = i ^ 1;
    return VGMI_OK;
}

namespace {
// one BGZF member header at p (n bytes available): total size, DEFLATE range, trailer.  0 = not (yet) a whole member,
// -1 = not a block-gzip member at all (SAM spec 4.1: gzip member with FEXTRA and a 'BC' subfield of 2 bytes)
int bgzf_member(const unsigned char* p, size_t n, uint32_t& total, uint32_t& d_off, uint32_t& d_len, uint32_t& crc, uint32_t& isize)
{
    if (n < 18) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return -1;   // FLG: FEXTRA and nothing else, as bgzip writes
    const uint32_t xlen = p[10] | (uint32_t)p[11] << 8;
    if (n < 12 + (size_t)xlen) return xlen > 4096 ? -1 : 0;
    uint32_t bsize = 0;
    bool found = false;
    for (uint32_t q = 0; q + 4 <= xlen;) {
        const unsigned char* sf = p + 12 + q;
        const uint32_t slen = sf[2] | (uint32_t)sf[3] << 8;
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && q + 6 <= xlen) {
            bsize = sf[4] | (uint32_t)sf[5] << 8;
            found = true;
        }
        q += 4 + slen;
    }
    if (!found) return -1;
    total = bsize + 1;
    if (total < 12 + xlen + 8) return -1;
    if (n < total) return 0;
    d_off = 12 + xlen;
    d_len = total - d_off - 8;
    memcpy(&crc, p + total - 8, 4);
    memcpy(&isize, p + total - 4, 4);
    if (isize > 65536) return -1;
    return 1;
}
}  // namespace

int vgmi_fastq_commit_bgzf(vgmi_fastq* f, size_t n_bytes, size_t* taken, size_t* n_text, int* not_bgzf)
{
    if (!f || !taken) return VGMI_E_INVALID;
    vgmi_ctx* c = f->c;
    *taken = 0;
    if (n_text) *n_text = 0;
    if (not_bgzf) *not_bgzf = 0;
    if (f->acquired < 0) return fail(c, VGMI_E_STATE, "no buffer acquired");
    if (n_bytes > f->cap) return fail(c, VGMI_E_INVALID, "more bytes than the buffer holds");
    const int i = f->acquired;
    HIPCHK(c, hipSetDevice(c->device));
    if (!f->d_members) {      // (d_comp may be there already: a stream of the pool that served an ordinary gzip file)
        f->max_members = (uint32_t)(f->text_cap / 4096) + 1024;     // bgzip members compress 64 KiB each; tiny ones are rare
        {
            const char* e = getenv("VGMI_BGZF_ROUNDS");
            f->bgzf_round = e && e[0] == '0' ? 0u : bgzf_wave_slots(c->n_cu);
        }
        // 512 KiB of zeroed slack behind the staged bytes: inside one damaged DEFLATE block the decoder can run up to
        // ~390 KB past its member before the per-block bound stops it (65 536 symbols x 48 bits); those reads must stay
        // inside the allocation (and see zeros) whatever the last member of a full batch contains
        constexpr size_t kCompSlack = 512u << 10;
        hipError_t e = hipSuccess;
        if (!f->d_comp) {
            e = hipMalloc(reinterpret_cast<void**>(&f->d_comp), f->cap + kCompSlack);
            if (e == hipSuccess) e = hipMemsetAsync(f->d_comp + f->cap, 0, kCompSlack, f->stream);
        }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_members), (size_t)f->max_members * sizeof(BgzfMember));
        for (int b = 0; b < 2 && e == hipSuccess; ++b)
            e = hipHostMalloc(reinterpret_cast<void**>(&f->h_members[b]), (size_t)f->max_members * sizeof(BgzfMember), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_status), (size_t)f->max_members * 4);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_crc), 1024);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&f->d_verdict), sizeof(BgzfVerdict));
        if (e == hipSuccess) {
            uint32_t tab[256];
            for (uint32_t n = 0; n < 256; ++n) {
                uint32_t v = n;
                for (int k = 0; k < 8; ++k) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
                tab[n] = v;
            }
            e = hipMemcpy(f->d_crc, tab, sizeof tab, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            const BgzfVerdict v{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
            e = hipMemcpy(f->d_verdict, &v, sizeof v, hipMemcpyHostToDevice);
        }
        HIPCHK(c, e);
    }
    // walk the member headers of the staged bytes: whole members whose text fits one chunk
    const unsigned char* p = reinterpret_cast<const unsigned char*>(f->h_stage[i]);
    BgzfMember* tab = f->h_members[i];
    uint32_t n_mem = 0, text = 0;
    size_t pos = 0;
    int stop = 0;
    while (pos < n_bytes && n_mem < f->max_members) {
        uint32_t total, d_off, d_len, crc, isize;
        const int r = bgzf_member(p + pos, n_bytes - pos, total, d_off, d_len, crc, isize);
        if (r <= 0) { stop = r; break; }
        if ((size_t)text + isize > f->text_cap) break;
        tab[n_mem] = BgzfMember{(uint32_t)(pos + d_off), d_len, text, isize, crc, 0u};
        f->member_size.push_back(total);
        ++n_mem;
        text += isize;
        pos += total;
    }
    if (stop < 0 && not_bgzf) *not_bgzf = 1;
    if (n_mem) {
        f->bgzf_avg_c = (double)pos / n_mem;
        f->bgzf_avg_u = (double)text / n_mem;
    }
    // no round of wavefronts for a handful of members: the caller asks for a little less than a whole number of rounds
    // (vgmi_fastq_bgzf_want); the few members a commit holds beyond one come again with the next bytes
    uint32_t round = f->bgzf_round;
    if (const char* e = getenv("VGMI_BGZF_ROUND_MEMBERS")) round = (uint32_t)atoi(e);      // tests: a round of a few members, so that small files are cut too
    if (round && n_mem > round && n_mem % round && n_mem % round <= (round + 7) / 8 && !stop) {
        const uint32_t keep = n_mem / round * round;
        for (uint32_t k = keep; k < n_mem; ++k) {
            pos -= f->member_size.back();
            f->member_size.pop_back();
        }
        n_mem = keep;
        text = tab[keep - 1].u_off + tab[keep - 1].u_len;
    }
    f->acquired = -1;
    *taken = pos;
    if (n_text) *n_text = text;
    if (n_mem == 0) return VGMI_OK;   // nothing whole yet (or not block gzip): the staging buffer stays with the caller
    f->batch_members.push_back(n_mem);
    HIPCHK(c, hipMemcpyAsync(f->d_comp, f->h_stage[i], pos, hipMemcpyHostToDevice, f->stream));
    HIPCHK(c, hipMemcpyAsync(f->d_members, tab, (size_t)n_mem * sizeof(BgzfMember), hipMemcpyHostToDevice, f->stream));
    HIPCHK(c, hipEventRecord(f->h_done[i], f->stream));
    f->h_busy[i] = true;
    HIPCHK(c, launch_bgzf_inflate(f->d_comp, f->d_members, n_mem, f->d_raw[i] + f->tail_max, f->d_status, f->d_crc, f->d_verdict, f->stream));
    FqBuffers b{};
    b.raw = f->d_raw[i];
    b.raw_next = f->d_raw[i ^ 1];
    b.packed = f->d_packed;
    b.tile = f->d_tile;
    b.nlpos = f->d_nlpos;
    b.rec_bytes = f->d_rec;
    b.out_off = f->d_off;
    b.block_sum = f->d_bsum;
    b.state = f->d_state;
    b.cap_lines = f->cap_lines;
    b.tail_max = f->tail_max;
    if (text) {
        HIPCHK(c, launch_fastq_chunk(b, text, f->stream, &f->d_verdict->good_bytes));
        int rc = launch_count(c, reinterpret_cast<const char*>(f->d_packed), f->tail_max + (size_t)text, nullptr, 0, f->stream,
                              &f->d_state->packed_bytes);
        if (rc) return rc;
    }
    f->next = i ^ 1;
    return VGMI_OK;
}

/* ---------------------------------------------------------------- ordinary gzip on the device (vgmi_gunzip.hip) */
namespace {
// RFC 1952 member header at p: bytes to the DEFLATE data, 0 if it is not one / does not fit n
size_t gzip_header_len(const unsigned char* p, size_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const unsigned flg = p[3];
    size_t q = 10;
    if (flg & 4) {
        if (q + 2 > n) return 0;
        q += 2 + ((size_t)p[q] | (size_t)p[q + 1] << 8);
    }
    for (unsigned bit : {8u, 16u})
        if (flg & bit) {
            while (q < n && p[q]) ++q;
            ++q;
        }
    if (flg & 2) q += 2;
    return q < n ? q : 0;
}

// One piece of a DEFLATE stream, on the device: comp[0, n) (n + >= 64 readable zero bytes behind it), first_bit = where a block
// starts (known), window = the 32 KiB of text in front (device; ignored at a member's start).  Decodes whole stretches into
// d_text and reports how far: *end_bit = the bit behind the last block taken (a block start, or the member's end when *final),
// *n_text its text.  Stretches the device cannot vouch for are left (end_bit says where they start).
struct GzScratch {
    uint32_t* d_starts = nullptr;
    GzSegHost* d_segs = nullptr;
    GzSegOutHost* d_outs = nullptr;
    uint64_t* d_toff = nullptr;
    uint16_t *d_pool = nullptr, *d_w1 = nullptr;      // symbols; the 16-bit window behind every stretch
    uint8_t* d_win = nullptr;                         // byte windows: in front of the piece, then behind every group of stretches
    uint32_t* d_chunk_r = nullptr;                    // CRC remainders of the text's 16 KiB chunks
    GzCrcState* d_crc = nullptr;                      // the member's running remainder and length: kept when the scratch grows
    size_t cap_seg = 0, cap_pool = 0, cap_sub = 0, cap_crc = 0;
    // everything but what a member carries from piece to piece (its CRC state; the window in front is handed back to the caller)
    uint8_t* release_scratch()
    {
        uint8_t* const win = d_win;
        for (void* q : {(void*)d_starts, (void*)d_segs, (void*)d_outs, (void*)d_toff, (void*)d_pool, (void*)d_w1, (void*)d_chunk_r})
            if (q) (void)hipFree(q);
        GzCrcState* const keep = d_crc;
        *this = GzScratch{};
        d_crc = keep;
        return win;
    }
    void release()
    {
        uint8_t* const win = release_scratch();
        if (win) (void)hipFree(win);
        if (d_crc) (void)hipFree(d_crc);
        d_crc = nullptr;
    }
};
// compressed bytes per guessed start (VGMI_GZ_SEG_KB for A/B; >= 32 KiB of compressed bytes hold a window of text for sure)
const uint32_t kGzSeg = [] {
    const char* e = getenv("VGMI_GZ_SEG_KB");
    const int v = e ? atoi(e) : 48;      // measured, reads/s with four host threads: 32 KiB 5.9e7, 48 KiB 6.3e7, 64 KiB 5.1e7 (first form of the decoder,
                                         // gpurun_out/r4q); wide batches: 32 KiB 7.5e7, 40 KiB 7.8e7, 48 KiB 8.4e7 (with a 2 048-entry ring, gpurun_out/r4w12)
    return ((uint32_t)(v < 32 ? 32 : v > 1024 ? 1024 : v) + 7u) / 8u * 8u << 10;      // a multiple of the search's sub-ranges
}();
// the scratch of a stream sized once for the largest piece its buffer can stage (VGMI_GZ_RESERVE=0: grown piece by piece -- the test of that path)
bool gz_reserve()
{
    const char* e = getenv("VGMI_GZ_RESERVE");
    return !(e && e[0] == '0');
}
constexpr uint32_t kGzRatio = 12;         // symbols of room per compressed byte of a stretch (FASTQ: 4-6)
constexpr uint32_t kGzSub = 2048;         // the block-start search: compressed bytes per wavefront (each reports the first start of its sub-range)

// n_reserve: the largest piece this stream will present (the scratch is sized once); member_start: the piece opens a member (its CRC
// state starts over, there is no window in front).
int gz_piece(vgmi_ctx* c, GzScratch& g, const uint8_t* d_comp, uint32_t n, uint32_t first_bit, uint32_t win_avail, uint8_t* d_text, size_t text_cap,
             hipStream_t st, uint32_t* end_bit, size_t* n_text, int* final_member, uint32_t* reason, size_t n_reserve = 0, bool member_start = true)
{
    *end_bit = first_bit;
    *n_text = 0;
    *final_member = 0;
    *reason = 0;
    if (n < 64 || (uint64_t)n * 8 >= (1ull << 32) - 4096) return VGMI_OK;
    const uint32_t n_nom = (n + kGzSeg - 1) / kGzSeg;
    const size_t pool_syms = (size_t)kGzRatio * n + (size_t)n_nom * 1024 + 65536;
    const uint32_t n_sub = (n + kGzSub - 1) / kGzSub;
    const size_t n_crc = gz_crc_chunks(text_cap) + 1;
    if (g.cap_seg < n_nom + 1 || g.cap_pool < pool_syms || g.cap_sub < n_sub || g.cap_crc < n_crc) {
        // (a piece larger than any before it: the window in front of it is the one thing in the scratch that the member still
        // needs -- it moves to the new allocation; the CRC state is not part of the scratch)
        const size_t m = std::max<size_t>(n, std::min<size_t>(n_reserve, (1ull << 29) - 4096));
        const uint32_t m_nom = (uint32_t)((m + kGzSeg - 1) / kGzSeg);
        uint8_t* const old_win = g.release_scratch();
        g.cap_seg = m_nom + 1;
        g.cap_pool = (size_t)kGzRatio * m + (size_t)m_nom * 1024 + 65536;
        g.cap_sub = (m + kGzSub - 1) / kGzSub;
        g.cap_crc = n_crc;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&g.d_starts), g.cap_sub * 4);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_segs), g.cap_seg * sizeof(GzSegHost));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_outs), g.cap_seg * sizeof(GzSegOutHost));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_toff), g.cap_seg * 8);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_pool), g.cap_pool * 2);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_win), ((size_t)gz_groups((uint32_t)g.cap_seg) + 2) * 32768);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_w1), g.cap_seg * 65536);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g.d_chunk_r), g.cap_crc * 4);
        if (e == hipSuccess && !g.d_crc) {
            e = hipMalloc(reinterpret_cast<void**>(&g.d_crc), sizeof(GzCrcState));
            if (e == hipSuccess) e = hipMemsetAsync(g.d_crc, 0, sizeof(GzCrcState), st);
        }
        if (e == hipSuccess && old_win) {
            e = hipMemcpyAsync(g.d_win, old_win, 32768, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (old_win) (void)hipFree(old_win);
        HIPCHK(c, e);
    }
    if (member_start) HIPCHK(c, hipMemsetAsync(g.d_crc, 0, sizeof(GzCrcState), st));
    // 1. guessed block starts: the first of every sub-range of kGzSub bytes; stretch j starts at the first one found at or behind
    // j * kGzSeg (and in front of (j + 2) * kGzSeg), the piece's first start is known.  (Stretches that run from a start to the first
    // one kGzSeg or more behind it come out half again as long -- DEFLATE blocks of FASTQ text are ~28 KiB apart -- and the decode
    // kernel, a single round of wavefronts, is as slow as its longest stretch: 19 against 14 ms, gpurun_out/r4w3.)
    HIPCHK(c, hipMemsetAsync(g.d_starts, 0xFF, (size_t)n_sub * 4, st));
    HIPCHK(c, launch_gz_find(d_comp, n, kGzSub, n_sub, kGzSeg / kGzSub, g.d_starts, st));
    std::vector<uint32_t> starts(n_sub);
    HIPCHK(c, hipMemcpyAsync(starts.data(), g.d_starts, (size_t)n_sub * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::vector<GzSegHost> segs;
    uint32_t at = first_bit;
    segs.push_back(GzSegHost{first_bit, 0xFFFFFFFFu, 0, 0, win_avail, 0});
    const uint32_t per = kGzSeg / kGzSub;
    for (uint32_t j = 1; j < n_nom; ++j) {
        uint32_t i = j * per;
        const uint32_t i_end = std::min<uint64_t>(n_sub, (uint64_t)(j + 2) * per);
        while (i < i_end && starts[i] == 0xFFFFFFFFu) ++i;
        if (i >= i_end || starts[i] <= at) continue;
        segs.back().stop_bit = starts[i];
        // (text in front of a later stretch: at least what the compressed bytes in front of it hold, a whole window almost always --
        // an understatement only makes a legal far reference an error, i.e. hands the stretch to the host decoder)
        segs.push_back(GzSegHost{starts[i], 0xFFFFFFFFu, 0, 0, (uint32_t)std::min<uint64_t>(32768, (uint64_t)win_avail + (starts[i] - first_bit) / 8), 0});
        at = starts[i];
    }
    // room in the symbol pool: by the compressed bytes of the stretch
    size_t off = 0;
    for (size_t i = 0; i < segs.size(); ++i) {
        const uint32_t stop = segs[i].stop_bit != 0xFFFFFFFFu ? segs[i].stop_bit : n * 8u;
        const size_t bytes = (stop - segs[i].start_bit + 7) / 8;
        size_t cap = (size_t)kGzRatio * bytes + 1024;
        cap = (cap + 1) & ~(size_t)1;
        if (off + cap > g.cap_pool) cap = (g.cap_pool - off) & ~(size_t)1;
        segs[i].sym_off = (uint32_t)off;
        segs[i].sym_cap = (uint32_t)cap;
        off += cap;
        if (off >= (1ull << 32)) return fail(c, VGMI_E_INVALID, "gzip piece too large for the symbol pool");
    }
    const uint32_t n_seg = (uint32_t)segs.size();
    // 2. decode
    HIPCHK(c, hipMemcpyAsync(g.d_segs, segs.data(), n_seg * sizeof(GzSegHost), hipMemcpyHostToDevice, st));
    HIPCHK(c, launch_gz_decode(d_comp, n, g.d_segs, n_seg, g.d_pool, g.d_outs, st));
    std::vector<GzSegOutHost> outs(n_seg);
    HIPCHK(c, hipMemcpyAsync(outs.data(), g.d_outs, n_seg * sizeof(GzSegOutHost), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    // the chain: a stretch counts when every one before it does and it ended exactly where the next starts (or with its member)
    uint32_t n_ok = 0;
    std::vector<uint64_t> toff(n_seg + 1, 0);
    for (uint32_t i = 0; i < n_seg; ++i) {
        const GzSegOutHost& o = outs[i];
        // (9 on the last stretch: the data ends inside it -- the next piece brings the rest; 3: no room, here or in the text chunk)
        if (o.status != 0) { *reason = o.status; break; }
        if (toff[i] + o.n_sym > text_cap) { *reason = 3; break; }
        toff[i + 1] = toff[i] + o.n_sym;
        n_ok = i + 1;
        *end_bit = o.end_bit;
        if (o.final_block) { *final_member = 1; break; }
    }
    if (n_ok == 0) return VGMI_OK;
    // 3. + 4. windows, then bytes
    HIPCHK(c, hipMemcpyAsync(g.d_toff, toff.data(), (size_t)n_ok * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, launch_gz_resolve(g.d_pool, g.d_segs, g.d_outs, g.d_toff, n_ok, g.d_w1, g.d_win, d_text, st));
    *n_text = (size_t)toff[n_ok];
    HIPCHK(c, launch_gz_crc(d_text, (uint64_t)toff[n_ok], g.d_chunk_r, g.d_crc, st));
    // the window behind the last stretch becomes the window in front of the next piece
    HIPCHK(c, hipMemcpyAsync(g.d_win, g.d_win + (size_t)gz_groups(n_ok) * 32768, 32768, hipMemcpyDeviceToDevice, st));
    return VGMI_OK;
}

// A member's trailer (CRC-32, ISIZE; RFC 1952) against the text the device resolved for it: what zlib checks behind gzread.
int gz_check_trailer(vgmi_ctx* c, GzScratch& g, const unsigned char* tr, hipStream_t st, bool* ok)
{
    *ok = true;
    if (!g.d_crc) return VGMI_OK;      // (a member without a single decoded piece never gets here)
    GzCrcState h;
    HIPCHK(c, hipMemcpyAsync(&h, g.d_crc, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const uint32_t want_crc = (uint32_t)tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
    const uint32_t want_len = (uint32_t)tr[4] | (uint32_t)tr[5] << 8 | (uint32_t)tr[6] << 16 | (uint32_t)tr[7] << 24;
    *ok = gz_crc_finish(h.r, h.len) == want_crc && (uint32_t)h.len == want_len;
    return VGMI_OK;
}
}  // namespace

// A whole gzip file from host memory to host memory through the device pipeline (test and bench of the primitive; the streaming
// form is vgmi_fastq_commit_gzip).  *consumed = compressed bytes the device took (whole file when it reached the member's end).
int vgmi_gunzip_buffer(vgmi_ctx* c, const void* host_gz, size_t n, void* host_out, size_t cap, size_t* n_out, size_t* consumed, int* member_end,
                       uint32_t* reason)
{
    if (!c || !host_gz || !host_out || !n_out) return VGMI_E_INVALID;
    *n_out = 0;
    if (consumed) *consumed = 0;
    if (member_end) *member_end = 0;
    if (reason) *reason = 0;
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned char* p = static_cast<const unsigned char*>(host_gz);
    const size_t hdr = gzip_header_len(p, n);
    if (!hdr) return fail(c, VGMI_E_INVALID, "not a gzip member");
    if (n >= (1u << 29)) return fail(c, VGMI_E_INVALID, "vgmi_gunzip_buffer: at most 512 MiB of compressed bytes per call");
    uint8_t *d_comp = nullptr, *d_text = nullptr;
    GzScratch g;
    int rc = VGMI_OK;
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_comp), n + 4096);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_text), cap ? cap : 1);
    if (he == hipSuccess) he = hipMemsetAsync(d_comp + n, 0, 4096, c->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(d_comp, host_gz, n, hipMemcpyHostToDevice, c->stream);
    uint32_t end_bit = 0, why = 0;
    size_t n_text = 0;
    int fin = 0;
    if (he == hipSuccess) rc = gz_piece(c, g, d_comp, (uint32_t)n, (uint32_t)hdr * 8u, 0, d_text, cap, c->stream, &end_bit, &n_text, &fin, &why);
    if (he == hipSuccess && rc == VGMI_OK) he = hipMemcpyAsync(host_out, d_text, n_text, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess && rc == VGMI_OK && fin && (size_t)(end_bit + 7) / 8 + 8 <= n) {      // reason 11: CRC-32 / ISIZE of the text
        bool ok = true;
        rc = gz_check_trailer(c, g, p + (size_t)(end_bit + 7) / 8, c->stream, &ok);
        if (rc == VGMI_OK && !ok) why = 11;
    }
    g.release();
    if (d_comp) (void)hipFree(d_comp);
    if (d_text) (void)hipFree(d_text);
    HIPCHK(c, he);
    if (rc) return rc;
    *n_out = n_text;
    if (consumed) *consumed = (end_bit + 7) / 8;
    if (member_end) *member_end = fin;
    if (reason) *reason = why;
    return VGMI_OK;
}

namespace {
void gz_scratch_free(void* g)
{
    if (!g) return;
    static_cast<GzScratch*>(g)->release();
    delete static_cast<GzScratch*>(g);
}
}  // namespace

// The streaming form: the staged bytes [0, n_bytes) of the acquired buffer continue an ordinary gzip stream -- at a member header
// when the stream is at a member's start, else at the byte that holds the next block's first bit (what the previous call left
// untaken).  Whole stretches between block starts are inflated into the chunk's text and parsed and counted like any text chunk;
// a member that ends inside the staged bytes is checked against its trailer (CRC-32, ISIZE) and the member behind it follows in the
// same call, as gzread runs members together.  *taken = staged bytes used up (the caller presents the rest again, in front of the
// bytes that follow).  *stop: 0 go on; 1 the gzip data is over (a member ended and what follows is no member header: gzread ignores
// it); 2 the device cannot take these bytes (vgmi_fastq_gzip_status says why; reason 11: the text of a member does not match its
// trailer): the host decoder carries on from the text the device parser has consumed.
int vgmi_fastq_commit_gzip(vgmi_fastq* f, size_t n_bytes, int at_eof, size_t* taken, size_t* n_text, int* stop)
{
    if (!f || !taken || !stop) return VGMI_E_INVALID;
    vgmi_ctx* c = f->c;
    *taken = 0;
    *stop = 0;
    if (n_text) *n_text = 0;
    if (f->acquired < 0) return fail(c, VGMI_E_STATE, "no buffer acquired");
    if (n_bytes > f->cap) return fail(c, VGMI_E_INVALID, "more bytes than the buffer holds");
    const int i = f->acquired;
    f->acquired = -1;
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned char* p = reinterpret_cast<const unsigned char*>(f->h_stage[i]);
    size_t pos = 0;               // staged bytes dealt with
    size_t text_total = 0;        // text of this call, behind one another in the chunk
    bool staged = false;          // the bytes are on the device
    uint32_t members_ended = 0;
    for (;;) {
        if (f->gz_skip) {                      // the rest of the last member's trailer
            const size_t k = std::min<size_t>(f->gz_skip, n_bytes - pos);
            memcpy(f->gz_trailer + (8 - f->gz_skip), p + pos, k);
            f->gz_skip -= (uint32_t)k;
            pos += k;
            if (f->gz_skip) {                  // (a trailer the file cuts short: the data is over, as for gzread)
                *taken = n_bytes;
                if (at_eof) *stop = 1;
                break;
            }
            bool ok = true;
            const int rc = f->gz ? gz_check_trailer(c, *static_cast<GzScratch*>(f->gz), f->gz_trailer, f->stream, &ok) : VGMI_OK;
            if (rc) return rc;
            if (!ok) { f->gz_reason = 11; *taken = pos; *stop = 2; break; }
        }
        uint32_t first_bit;
        if (!f->gz_in_member) {
            if (n_bytes - pos < 2 || p[pos] != 0x1f || p[pos + 1] != 0x8b) {
                if (n_bytes - pos >= 2 || at_eof) { *taken = n_bytes; *stop = 1; }      // no further member: the data is over
                else *taken = pos;
                break;
            }
            // (members of a few kilobytes each, one after another: a piece per member is launch-bound -- the host decoder's case)
            if (members_ended >= 8 && text_total < ((size_t)members_ended << 20)) { f->gz_reason = 12; *taken = pos; *stop = 2; break; }
            const size_t hdr = gzip_header_len(p + pos, n_bytes - pos);
            if (!hdr) {
                if (n_bytes - pos >= 65536 + 64 || at_eof) { f->gz_reason = 10; *stop = 2; }      // a header that does not parse
                *taken = pos;
                break;
            }
            first_bit = (uint32_t)(pos + hdr) * 8u;
            f->gz_avail = 0;
            f->gz_member_text = 0;
            f->gz_in_member = true;
        } else first_bit = (uint32_t)pos * 8u + f->gz_bit;
        if (!f->gz) f->gz = new (std::nothrow) GzScratch();
        if (!f->gz) return fail(c, VGMI_E_NOMEM, "out of memory");
        if (!staged) {
            if (!f->d_comp) {
                constexpr size_t kCompSlack = 512u << 10;
                hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->d_comp), f->cap + kCompSlack);
                if (e == hipSuccess) e = hipMemsetAsync(f->d_comp + f->cap, 0, kCompSlack, f->stream);
                HIPCHK(c, e);
            }
            HIPCHK(c, hipMemcpyAsync(f->d_comp, f->h_stage[i], n_bytes, hipMemcpyHostToDevice, f->stream));
            if (n_bytes < f->cap) HIPCHK(c, hipMemsetAsync(f->d_comp + n_bytes, 0, std::min<size_t>(4096, f->cap - n_bytes), f->stream));
            HIPCHK(c, hipEventRecord(f->h_done[i], f->stream));
            f->h_busy[i] = true;
            staged = true;
        }
        // a member behind the call's first: the piece starts at the (aligned) bytes it starts in
        const size_t base = members_ended ? (size_t)(first_bit / 8u) / kGzSub * kGzSub : 0;
        uint32_t end_bit = 0, why = 0;
        size_t text = 0;
        int fin = 0;
        int rc = gz_piece(c, *static_cast<GzScratch*>(f->gz), f->d_comp + base, (uint32_t)(n_bytes - base), first_bit - (uint32_t)base * 8u, f->gz_avail,
                          f->d_raw[i] + f->tail_max + text_total, f->text_cap - text_total, f->stream, &end_bit, &text, &fin, &why, gz_reserve() ? f->cap : 0,
                          f->gz_member_text == 0);
        if (rc) return rc;
        end_bit += (uint32_t)base * 8u;
        const bool broken = why != 0 && why != 9 && why != 3;       // a stretch that does not decode / does not meet the next one
        if (text == 0 && !fin) {
            // no whole stretch in these bytes: more may help -- unless there are no more, the buffer is full already, or it is no DEFLATE
            // (behind a member that ended in this call: the caller presents the rest again, and the question is asked then)
            if ((at_eof || n_bytes == f->cap || broken) && members_ended == 0) {
                f->gz_reason = why ? why : 9;
                *stop = 2;
            }
            *taken = first_bit / 8u;      // (a header just read is taken; the block's byte stays)
            f->gz_bit = first_bit & 7u;
            break;
        }
        f->gz_text += text;
        f->gz_member_text += text;
        text_total += text;
        f->gz_avail = (uint32_t)std::min<uint64_t>(32768, (uint64_t)f->gz_avail + text);
        if (!fin) {
            *taken = end_bit / 8u;
            f->gz_bit = end_bit & 7u;
            if (broken) { f->gz_reason = why; *stop = 2; }       // a stretch behind the ones taken went wrong: the host goes on from the text so far
            break;
        }
        // the member's end: its trailer (the part of it that is here), then whatever follows
        f->gz_in_member = false;
        f->gz_bit = 0;
        f->gz_skip = 8;
        pos = (size_t)(end_bit + 7) / 8;
        ++members_ended;
        if (pos >= n_bytes && !at_eof) { *taken = n_bytes; break; }
    }
    if (n_text) *n_text = text_total;
    if (text_total) {
        FqBuffers b{};
        b.raw = f->d_raw[i];
        b.raw_next = f->d_raw[i ^ 1];
        b.packed = f->d_packed;
        b.tile = f->d_tile;
        b.nlpos = f->d_nlpos;
        b.rec_bytes = f->d_rec;
        b.out_off = f->d_off;
        b.block_sum = f->d_bsum;
        b.state = f->d_state;
        b.cap_lines = f->cap_lines;
        b.tail_max = f->tail_max;
        HIPCHK(c, launch_fastq_chunk(b, (uint32_t)text_total, f->stream));
        const int rc = launch_count(c, reinterpret_cast<const char*>(f->d_packed), f->tail_max + text_total, nullptr, 0, f->stream, &f->d_state->packed_bytes);
        if (rc) return rc;
        f->next = i ^ 1;
    }
    return VGMI_OK;
}

int vgmi_fastq_gzip_status(vgmi_fastq* f, uint64_t* device_text_bytes, uint32_t* reason)
{
    if (!f) return VGMI_E_INVALID;
    if (device_text_bytes) *device_text_bytes = f->gz_text;
    if (reason) *reason = f->gz_reason;
    return VGMI_OK;
}

int vgmi_fastq_bgzf_status(vgmi_fastq* f, int* failed, uint64_t* good_compressed_bytes, uint32_t* reason)
{
    if (!f || !failed) return VGMI_E_INVALID;
    vgmi_ctx* c = f->c;
    *failed = 0;
    if (good_compressed_bytes) *good_compressed_bytes = 0;
    if (!f->d_verdict) return VGMI_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(f->stream));
    BgzfVerdict v;
    HIPCHK(c, hipMemcpy(&v, f->d_verdict, sizeof v, hipMemcpyDeviceToHost));
    uint64_t bytes = 0;
    size_t mi = 0;
    if (v.first_bad_batch == 0xFFFFFFFFu) {
        for (uint64_t sz : f->member_size) bytes += sz;
    } else {
        *failed = 1;
        if (reason) *reason = v.reason;
        for (uint32_t b = 0; b < v.first_bad_batch && b < f->batch_members.size(); ++b)
            for (uint32_t k = 0; k < f->batch_members[b]; ++k) bytes += f->member_size[mi++];
        for (uint32_t k = 0; k < v.first_bad_member && mi < f->member_size.size(); ++k) bytes += f->member_size[mi++];
    }
    if (good_compressed_bytes) *good_compressed_bytes = bytes;
    return VGMI_OK;
}

int vgmi_fastq_close(vgmi_fastq* f, uint64_t* n_records, uint64_t* n_bases, uint64_t* consumed_bytes, int* stopped,
                     char* tail_out, size_t tail_cap, size_t* tail_len)
{
    if (!f) return VGMI_E_INVALID;
    vgmi_ctx* c = f->c;
    int rc = VGMI_OK;
    FqState st{};
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamSynchronize(f->stream);
    if (e == hipSuccess) e = hipMemcpy(&st, f->d_state, sizeof st, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        if (n_records) *n_records = st.n_records;
        if (n_bases) *n_bases = st.n_bases;
        if (consumed_bytes) *consumed_bytes = st.consumed;
        if (stopped) *stopped = (int)st.stopped;
        if (tail_len) *tail_len = st.stopped ? 0 : st.tail_len;
        if (!st.stopped && st.tail_len) {
            if (!tail_out || tail_cap < st.tail_len) rc = fail(c, VGMI_E_INVALID, "tail buffer too small (1 MiB suffices)");
            // the carry kernel left the tail in front of the landing area of the buffer the next chunk would have used
            else e = hipMemcpy(tail_out, f->d_raw[f->next] + f->tail_max - st.tail_len, st.tail_len, hipMemcpyDeviceToHost);
        }
    }
    bool keep = false;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->open_fastq--;
        if (e == hipSuccess) c->read_base += st.n_bases;
        if (e == hipSuccess && c->fastq_pool.size() < 4) {
            c->fastq_pool.push_back(f);
            keep = true;
        }
    }
    if (!keep) fastq_free(f);
    if (e != hipSuccess) HIPCHK(c, e);
    return rc;
}

/* ---------------------------------------------------------------- K1 trace */

int vgmi_sketch_keys(vgmi_ctx* c, const char* bases, size_t n_bytes, const uint64_t* read_off, size_t n_reads,
                     uint32_t k, uint64_t* keys_out)
{
    if (!c) return VGMI_E_INVALID;
    if (k < 1 || k > 28) return fail(c, VGMI_E_INVALID, "k must be in 1..28");
    if (n_bytes == 0) return VGMI_OK;
    if (!bases || !keys_out) return fail(c, VGMI_E_INVALID, "NULL buffer");
    if (bases[n_bytes - 1] != '\n') return fail(c, VGMI_E_INVALID, "a read block must end with '\\n'");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> derived;
    if (!(k & 1) && !read_off) {
        derived = offsets_from_newlines(bases, n_bytes);
        if (derived.size() != n_reads + 1) return fail(c, VGMI_E_INVALID, "n_reads does not match the number of '\\n'");
        read_off = derived.data();
    }
    char* d_b = nullptr;
    uint64_t* d_k = nullptr;
    uint64_t* d_off = nullptr;
    int rc = VGMI_OK;
    auto cleanup = [&]() {
        if (d_b) (void)hipFree(d_b);
        if (d_k) (void)hipFree(d_k);
        if (d_off) (void)hipFree(d_off);
    };
#define HIPCHK_CL(call) do { hipError_t ecl_ = (call); if (ecl_ != hipSuccess) { cleanup(); HIPCHK(c, ecl_); } } while (0)
    HIPCHK_CL(hipMalloc(reinterpret_cast<void**>(&d_b), n_bytes + 16));
    HIPCHK_CL(hipMalloc(reinterpret_cast<void**>(&d_k), n_bytes * 8));
    HIPCHK_CL(hipMemcpyAsync(d_b, bases, n_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK_CL(hipMemsetAsync(c->d_status, 0, 4, c->stream));
    RowParams p = row_params(c, d_b, n_bytes, k);
    p.keys_out = d_k;
    if (k & 1) {
        uint32_t grid, block;
        rows_geometry(c, false, grid, block);
        HIPCHK_CL(launch_rows(K_MODE_KEYS, false, p, grid, block, c->stream));
    } else {
        HIPCHK_CL(hipMalloc(reinterpret_cast<void**>(&d_off), (n_reads + 1) * 8));
        HIPCHK_CL(hipMemcpyAsync(d_off, read_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK_CL(launch_seq(K_MODE_KEYS, p, d_off, n_reads, c->stream));
    }
    HIPCHK_CL(hipMemcpyAsync(keys_out, d_k, n_bytes * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK_CL(hipStreamSynchronize(c->stream));
    rc = check_status(c);
    cleanup();
    return rc;
}

/* ---------------------------------------------------------------- Bloom */

int vgmi_bloom_params(uint64_t n, double p, uint64_t* m, uint32_t* n_hash)
{
    // BloomFilter::_calculate_size / _calculate_num_hashes (src/counting_bloom_filter.cpp:70-77)
    const uint64_t mm = (uint64_t)std::ceil(((double)n * std::log(p)) / std::log(1.0 / std::pow(2.0, std::log(2.0))));
    if (m) *m = mm;
    if (n_hash) *n_hash = (uint32_t)std::round((double)mm * std::log(2.0) / (double)n);
    return VGMI_OK;
}

int vgmi_bloom_create(vgmi_ctx* c, uint64_t m, uint32_t n_hash, const uint64_t* seeds)
{
    if (!c || !seeds) return VGMI_E_INVALID;
    if (m == 0 || n_hash == 0 || n_hash > VG_BLOOM_MAX_HASH) return fail(c, VGMI_E_INVALID, "bad Bloom geometry");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->bv.filter) (void)hipFree(c->bv.filter);
    if (c->bb_scratch) (void)hipFree(c->bb_scratch);
    c->bb_scratch = nullptr;
    c->bb_cap = 0;
    c->bv = BloomView{};
    c->has_bloom = false;
    c->bloom_alloc = ((m + 3) & ~3ULL) + 16;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->bv.filter), c->bloom_alloc));
    HIPCHK(c, hipMemset(c->bv.filter, 0, c->bloom_alloc));
    c->bv.m = m;
    c->bv.magic = UINT64_MAX / m;
    c->bv.n_hash = n_hash;
    for (uint32_t i = 0; i < n_hash; ++i) c->bv.seeds[i] = (uint32_t)seeds[i];  // `unsigned int seed`
    for (uint32_t i = 0; i < n_hash; ++i) c->bloom_seeds64[i] = seeds[i];
    c->has_bloom = true;
    return VGMI_OK;
}

int vgmi_bloom_add_seq_device(vgmi_ctx* c, const char* dev_bases, uint64_t len, uint32_t k)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    if (k < 1 || k > 28) return fail(c, VGMI_E_INVALID, "k must be in 1..28");
    if (len == 0) return fail(c, VGMI_E_EMPTY_READ, "empty sequence (reference: assert(len > 0), kmer.cpp:27)");
    if (reinterpret_cast<uintptr_t>(dev_bases) & 15) return fail(c, VGMI_E_INVALID, "dev_bases must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    RowParams p = row_params(c, dev_bases, len, k);
    if (k & 1) {
        uint32_t grid, block;
        rows_geometry(c, false, grid, block);
        // long sequences: positions binned by 128 KiB chunk of the filter and counted in LDS (vgmi_bloom_bin.hip) -- worth it when
        // every chunk gets a few thousand positions; VGMI_BLOOM_BINNED=0 keeps the direct form
        static const bool binned = !(getenv("VGMI_BLOOM_BINNED") && getenv("VGMI_BLOOM_BINNED")[0] == '0');
        const BloomBinPlan plan = binned && len >= (4u << 20) ? bloom_bin_plan(c->bv.m, c->bv.n_hash, len) : BloomBinPlan{};
        if (plan.ok && (double)len * c->bv.n_hash >= 2048.0 * plan.n_chunks) {
            const size_t keys_bytes = (len * 8 + 255) & ~(size_t)255, need = keys_bytes + plan.scratch_bytes;
            if (c->bb_cap < need) {
                if (c->bb_scratch) (void)hipFree(c->bb_scratch);
                c->bb_scratch = nullptr;
                c->bb_cap = 0;
                if (hipMalloc(reinterpret_cast<void**>(&c->bb_scratch), need) == hipSuccess) c->bb_cap = need;
                else (void)hipGetLastError();          // no room: the direct form
            }
            if (c->bb_cap >= need) {
                RowParams pk = p;
                pk.keys_out = reinterpret_cast<uint64_t*>(c->bb_scratch);
                HIPCHK(c, launch_rows(K_MODE_KEYS, false, pk, grid, block, c->stream));
                int overflowed = 0;
                HIPCHK(c, launch_bloom_binned(c->bv, pk.keys_out, len, plan, c->bb_scratch + keys_bytes, c->n_cu, c->stream, &overflowed));
                if (!overflowed) return VGMI_OK;       // (a bin out of room -- one k-mer repeated through the call: nothing applied, the direct form does it)
            }
        }
        HIPCHK(c, launch_rows(K_MODE_BLOOM, false, p, grid, block, c->stream));
    } else {
        // even k: the sequential state machine, one lane per 1 KiB segment with its state rebuilt by look-back
        HIPCHK(c, launch_bloom_even(p, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return VGMI_OK;
}

int vgmi_bloom_add_seq(vgmi_ctx* c, const char* bases, uint64_t len, uint32_t k)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    if (len == 0) return fail(c, VGMI_E_EMPTY_READ, "empty sequence (reference: assert(len > 0), kmer.cpp:27)");
    if (!bases) return fail(c, VGMI_E_INVALID, "bases is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    char* d = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), len + 16));
    hipError_t e = hipMemcpy(d, bases, len, hipMemcpyHostToDevice);
    int rc = VGMI_OK;
    if (e == hipSuccess) {
        rc = vgmi_bloom_add_seq_device(c, d, len, k);
        if (rc == VGMI_OK) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d);
    if (rc) return rc;
    HIPCHK(c, e);
    return VGMI_OK;
}

namespace {
// Device working memory of the HMM calls is kept in the context between calls: hipFree waits for every stream of the device --
// other parts', other samples' chains -- so nothing is freed while samples are genotyped.
uint8_t* hmm_block_take(vgmi_ctx* c, size_t bytes, size_t& got)
{
    {
        std::lock_guard<std::mutex> lock(c->hmm_mu);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < c->hmm_blocks.size(); ++i)
            if (c->hmm_blocks[i].second >= bytes && (best == SIZE_MAX || c->hmm_blocks[i].second < c->hmm_blocks[best].second)) best = i;
        if (best != SIZE_MAX) {
            uint8_t* d = c->hmm_blocks[best].first;
            got = c->hmm_blocks[best].second;
            c->hmm_blocks.erase(c->hmm_blocks.begin() + (ptrdiff_t)best);
            return d;
        }
    }
    uint8_t* d = nullptr;
    got = bytes;
    if (hipMalloc(reinterpret_cast<void**>(&d), bytes) == hipSuccess) return d;
    (void)hipGetLastError();
    std::vector<std::pair<uint8_t*, size_t>> drop;     // the kept ones that are too small make room
    {
        std::lock_guard<std::mutex> lock(c->hmm_mu);
        drop.swap(c->hmm_blocks);
    }
    for (auto& b : drop) (void)hipFree(b.first);
    if (hipMalloc(reinterpret_cast<void**>(&d), bytes) == hipSuccess) return d;
    (void)hipGetLastError();
    return nullptr;
}

void hmm_block_give(vgmi_ctx* c, uint8_t* d, size_t bytes)
{
    if (!d) return;
    std::lock_guard<std::mutex> lock(c->hmm_mu);
    c->hmm_blocks.emplace_back(d, bytes);
}

// recursion (+ posterior when gid is given) in one pass over device buffers: alpha / beta leave the device only if `out` asks.
// Every array is indexed by GLOBAL row / step; this call reads and writes rows [row_lo, row_hi) and steps [step_lo, step_hi) only
// (device buffers of that size, the kernels' pointers moved back by the range's start).  It works on a stream of its own and
// touches nothing of the context but its device and error text: calls on parts of the same arrays may run side by side.
int hmm_run(vgmi_ctx* c, uint32_t n_gt, uint32_t ploidy, const uint8_t* keep, uint32_t n_windows, const void* obs, uint64_t row_lo,
            uint64_t row_hi, const uint32_t* row, const uint8_t* restart, const void* pow, uint64_t step_lo, uint64_t step_hi,
            const void* uniform, const vgmi_hmm_chain* chains, uint32_t n_chains, void* out, const uint8_t* gid, const uint8_t* order,
            const uint64_t* fwd_step, const uint64_t* bwd_step, void* prob, uint32_t* winner, const uint8_t* dev_obs = nullptr)
{
    // dev_obs: the emission rows [row_lo, row_hi) are already on the device (vgmi_hmm_emissions); obs is then not read
    if (!c || !keep || (!obs && !dev_obs) || !row || !restart || !pow || !uniform || !chains) return VGMI_E_INVALID;
    if (n_gt < 1 || n_gt > VGMI_HMM_MAX_GT || ploidy < 1 || ploidy > 4) return fail(c, VGMI_E_INVALID, "HMM recursion: 1..2048 genotypes of 1..4 haplotypes");
    if (n_gt > 128)     // the many-genotype kernel reads keep[p][g] for keep[g][p]: what two genotypes share is symmetric
        for (uint32_t w = 0; w < n_windows; ++w) {
            const uint8_t* m = keep + (size_t)w * n_gt * n_gt;
            for (uint32_t i = 0; i < n_gt; ++i)
                for (uint32_t j = i + 1; j < n_gt; ++j)
                    if (m[(size_t)i * n_gt + j] != m[(size_t)j * n_gt + i]) return fail(c, VGMI_E_INVALID, "HMM recursion: keep matrix not symmetric");
        }
    if (row_lo > row_hi || step_lo > step_hi) return fail(c, VGMI_E_INVALID, "HMM recursion: an empty-handed range");
    const uint64_t n_rows = row_hi - row_lo, n_steps = step_hi - step_lo;
    for (uint32_t i = 0; i < n_chains; ++i)
        if (chains[i].keep_index >= n_windows || chains[i].first_step < step_lo || chains[i].first_step + chains[i].n_steps > step_hi)
            return fail(c, VGMI_E_INVALID, "HMM recursion: a chain points outside its arrays");
    for (uint64_t s = step_lo; s < step_hi; ++s)
        if (row[s] < row_lo || row[s] >= row_hi) return fail(c, VGMI_E_INVALID, "HMM recursion: a step points outside the emission rows");
    if (gid)
        for (uint64_t i = row_lo; i < row_hi; ++i)
            if (fwd_step[i] < step_lo || fwd_step[i] >= step_hi || bwd_step[i] < step_lo || bwd_step[i] >= step_hi)
                return fail(c, VGMI_E_INVALID, "HMM posterior: a row points outside the steps");
    if (n_steps == 0 || n_chains == 0) return VGMI_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t stride = ploidy + 1;
    const size_t b_keep = (size_t)n_windows * n_gt * n_gt, w_obs = (size_t)n_gt * 16, b_obs = dev_obs ? 0 : (size_t)n_rows * w_obs, b_row = (size_t)n_steps * 4,
                 w_pow = (size_t)2 * stride * 16, b_pow = (size_t)n_steps * w_pow, b_ch = (size_t)n_chains * sizeof(vgmi_hmm_chain),
                 b_out = (size_t)n_steps * w_obs, b_gid = gid ? (size_t)n_rows * n_gt : 0, b_fs = gid ? (size_t)n_rows * 8 : 0,
                 b_prob = gid ? (size_t)n_rows * 16 : 0, b_win = gid ? (size_t)n_rows * 4 : 0;
    static_assert(sizeof(vgmi_hmm_chain) == sizeof(HmmChain), "chain layout");
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_keep = 0, o_obs = up(o_keep + b_keep), o_row = up(o_obs + b_obs), o_rs = up(o_row + b_row), o_pow = up(o_rs + n_steps),
                 o_uni = up(o_pow + b_pow), o_ch = o_uni + 256, o_out = up(o_ch + b_ch), o_gid = up(o_out + b_out), o_ord = up(o_gid + b_gid),
                 o_fs = up(o_ord + b_gid), o_bs = up(o_fs + b_fs), o_prob = up(o_bs + b_fs), o_win = up(o_prob + b_prob), total = up(o_win + b_win);
    const auto h0 = std::chrono::steady_clock::now();
    uint8_t* d = nullptr;
    size_t d_bytes = 0;
    {
        // the smallest kept block that is large enough, else a new one (the kept ones that are too small make room first)
        std::lock_guard<std::mutex> lock(c->hmm_mu);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < c->hmm_blocks.size(); ++i)
            if (c->hmm_blocks[i].second >= total && (best == SIZE_MAX || c->hmm_blocks[i].second < c->hmm_blocks[best].second)) best = i;
        if (best != SIZE_MAX) {
            d = c->hmm_blocks[best].first;
            d_bytes = c->hmm_blocks[best].second;
            c->hmm_blocks.erase(c->hmm_blocks.begin() + (ptrdiff_t)best);
        }
    }
    hipError_t e = hipSuccess;
    if (!d) {
        d_bytes = total;
        e = hipMalloc(reinterpret_cast<void**>(&d), total);
        if (e != hipSuccess) {
            std::vector<std::pair<uint8_t*, size_t>> drop;
            {
                std::lock_guard<std::mutex> lock(c->hmm_mu);
                drop.swap(c->hmm_blocks);
            }
            for (auto& b : drop) (void)hipFree(b.first);
            e = hipMalloc(reinterpret_cast<void**>(&d), total);
        }
        if (e != hipSuccess) return fail(c, VGMI_E_NOMEM, "HMM recursion: not enough device memory");
    }
    auto keep_block = [&]() {
        std::lock_guard<std::mutex> lock(c->hmm_mu);
        c->hmm_blocks.emplace_back(d, d_bytes);
    };
    const auto h1 = std::chrono::steady_clock::now();
    hipStream_t st = nullptr;
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) {
        keep_block();
        HIPCHK(c, e);
    }
    // VGMI_HMM_TIMING=1: upload / recursion / posterior + download, milliseconds on stderr (diagnostics)
    const bool timing = getenv("VGMI_HMM_TIMING") != nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (timing)
        for (auto& x : ev) (void)hipEventCreate(&x);
    if (timing) (void)hipEventRecord(ev[0], st);
    const uint8_t* h_obs = dev_obs ? nullptr : static_cast<const uint8_t*>(obs) + row_lo * w_obs;
    const uint8_t* h_pow = static_cast<const uint8_t*>(pow) + step_lo * w_pow;
    e = hipMemcpyAsync(d + o_keep, keep, b_keep, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && !dev_obs) e = hipMemcpyAsync(d + o_obs, h_obs, b_obs, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_row, row + step_lo, b_row, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_rs, restart + step_lo, n_steps, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_pow, h_pow, b_pow, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_uni, uniform, 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_ch, chains, b_ch, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && gid) e = hipMemcpyAsync(d + o_gid, gid + row_lo * n_gt, b_gid, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && gid) e = hipMemcpyAsync(d + o_ord, order + row_lo * n_gt, b_gid, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && gid) e = hipMemcpyAsync(d + o_fs, fwd_step + row_lo, b_fs, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && gid) e = hipMemcpyAsync(d + o_bs, bwd_step + row_lo, b_fs, hipMemcpyHostToDevice, st);
    const auto h2 = std::chrono::steady_clock::now();
    // where global row / step 0 would lie (the kernels only touch the range)
    auto back = [](uint8_t* p, size_t bytes) { return reinterpret_cast<uint8_t*>(reinterpret_cast<uintptr_t>(p) - bytes); };
    if (e == hipSuccess) {
        HmmParams P{};
        P.n_gt = n_gt;
        P.ploidy = ploidy;
        P.keep = d + o_keep;
        P.obs = dev_obs ? back(const_cast<uint8_t*>(dev_obs), row_lo * w_obs) : back(d + o_obs, row_lo * w_obs);
        P.row = reinterpret_cast<const uint32_t*>(back(d + o_row, step_lo * 4));
        P.restart = back(d + o_rs, step_lo);
        P.pow = back(d + o_pow, step_lo * w_pow);
        P.uniform = d + o_uni;
        P.chains = reinterpret_cast<const HmmChain*>(d + o_ch);
        P.out = back(d + o_out, step_lo * w_obs);
        if (timing) (void)hipEventRecord(ev[1], st);
        e = launch_hmm_recursion(P, n_chains, st);
        if (timing) (void)hipEventRecord(ev[2], st);
    }
    if (e == hipSuccess && gid) {
        HmmPostParams Q{};
        Q.n_gt = n_gt;
        Q.row0 = row_lo;
        Q.ab = back(d + o_out, step_lo * w_obs);
        Q.fwd_step = reinterpret_cast<const uint64_t*>(back(d + o_fs, row_lo * 8));
        Q.bwd_step = reinterpret_cast<const uint64_t*>(back(d + o_bs, row_lo * 8));
        Q.gid = back(d + o_gid, row_lo * n_gt);
        Q.order = back(d + o_ord, row_lo * n_gt);
        Q.prob = back(d + o_prob, row_lo * 16);
        Q.winner = reinterpret_cast<uint32_t*>(back(d + o_win, row_lo * 4));
        e = launch_hmm_posterior(Q, n_rows, st);
        if (e == hipSuccess) e = hipMemcpyAsync(static_cast<uint8_t*>(prob) + row_lo * 16, d + o_prob, b_prob, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(winner + row_lo, d + o_win, b_win, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess && out) e = hipMemcpyAsync(static_cast<uint8_t*>(out) + step_lo * w_obs, d + o_out, b_out, hipMemcpyDeviceToHost, st);
    if (timing) (void)hipEventRecord(ev[3], st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (timing) {
        float a = 0, b = 0, g = 0;
        if (e == hipSuccess) {
            (void)hipEventElapsedTime(&a, ev[0], ev[1]);
            (void)hipEventElapsedTime(&b, ev[1], ev[2]);
            (void)hipEventElapsedTime(&g, ev[2], ev[3]);
        }
        auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
        fprintf(stderr, "[vgmi] HMM on the device: %u chains, %llu steps, upload %.1f ms (%.0f MB), recursion %.1f ms, posterior + download %.1f ms; "
                        "host: memory %.1f ms, copies issued in %.1f ms, whole call %.1f ms\n",
                n_chains, (unsigned long long)n_steps, a, (double)(b_keep + b_obs + b_row + b_pow + 2 * b_gid + 2 * b_fs) / 1e6, b, g, ms(h0, h1),
                ms(h1, h2), ms(h0, std::chrono::steady_clock::now()));
        for (auto& x : ev) (void)hipEventDestroy(x);
    }
    (void)hipStreamDestroy(st);
    keep_block();
    HIPCHK(c, e);
    return VGMI_OK;
}
}  // namespace

struct vgmi_hmm_part {
    vgmi_ctx* c = nullptr;
    uint8_t* d_obs = nullptr;
    size_t obs_bytes = 0;      // of the block d_obs came as
    uint64_t n_rows = 0;
    uint32_t n_gt = 0;
    // the emission launch's arguments and the block its row arrays and tables live in: vgmi_hmm_part_fix_rows scores rows again
    HmmEmitParams emit{};
    uint8_t* d_small = nullptr;
    size_t small_bytes = 0;
    std::vector<uint32_t> entry_count;      // (host copy: fix_j is checked against it)
};

int vgmi_hmm_entries_upload(vgmi_ctx* c, const uint64_t* entries, size_t n)
{
    if (!c || (n && !entries)) return VGMI_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->d_hmm_entries) (void)hipFree(c->d_hmm_entries);
    if (c->d_hmm_cov) (void)hipFree(c->d_hmm_cov);
    c->d_hmm_entries = nullptr;
    c->d_hmm_cov = nullptr;
    c->hmm_n_entries = n;
    if (hipMalloc(reinterpret_cast<void**>(&c->d_hmm_entries), (n ? n : 1) * 8) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_hmm_cov), n ? n : 1) != hipSuccess)
        return fail(c, VGMI_E_NOMEM, "HMM emissions: not enough device memory for the node-list entries");
    if (n) HIPCHK(c, hipMemcpy(c->d_hmm_entries, entries, n * 8, hipMemcpyHostToDevice));
    return VGMI_OK;
}

int vgmi_hmm_sample_upload(vgmi_ctx* c, const uint8_t* cov_node, size_t n)
{
    if (!c || (n && !cov_node)) return VGMI_E_INVALID;
    if (!c->d_hmm_cov || n != c->hmm_n_entries) return fail(c, VGMI_E_STATE, "HMM emissions: upload the entries first");
    HIPCHK(c, hipSetDevice(c->device));
    if (n) HIPCHK(c, hipMemcpy(c->d_hmm_cov, cov_node, n, hipMemcpyHostToDevice));
    return VGMI_OK;
}

int vgmi_hmm_emissions(vgmi_ctx* c, uint32_t n_gt, uint32_t n_used, const uint8_t* used, const uint8_t* pos_a, const uint8_t* pos_b,
                       uint64_t top_mask, uint32_t bit_len, float ave, double lower, double upper, const void* tables, uint64_t n_rows,
                       const uint64_t* entry_begin, const uint32_t* entry_count, const uint16_t* gt0, uint32_t* n_kept_out,
                       uint8_t* flags_out, vgmi_hmm_part** out)
{
    if (!pos_a || !pos_b || n_gt < 1 || n_gt > 128) return VGMI_E_INVALID;
    std::vector<uint8_t> pos(2 * (size_t)n_gt);
    for (uint32_t g = 0; g < n_gt; ++g) {
        pos[2 * g] = pos_a[g];
        pos[2 * g + 1] = pos_b[g];
    }
    return vgmi_hmm_emissions_ploidy(c, n_gt, 2, n_used, used, pos.data(), top_mask, bit_len, ave, lower, upper, tables, n_rows, entry_begin, entry_count, gt0,
                                     n_kept_out, flags_out, out);
}

// ... for genotypes of `ploidy` haplotypes (2 .. 4): pos[g * ploidy + q] = the place in `used` of genotype g's q-th haplotype; tables holds
// (ploidy + 1) x 256 terms (geometric for h = 0, Poisson(ave * h) for h = 1 .. ploidy)
int vgmi_hmm_emissions_ploidy(vgmi_ctx* c, uint32_t n_gt, uint32_t ploidy, uint32_t n_used, const uint8_t* used, const uint8_t* pos, uint64_t top_mask,
                              uint32_t bit_len, float ave, double lower, double upper, const void* tables, uint64_t n_rows, const uint64_t* entry_begin,
                              const uint32_t* entry_count, const uint16_t* gt0, uint32_t* n_kept_out, uint8_t* flags_out, vgmi_hmm_part** out)
{
    if (!c || !used || !pos || !tables || !out) return VGMI_E_INVALID;
    if (ploidy < 2 || ploidy > 4) return fail(c, VGMI_E_INVALID, "HMM emissions: genotypes of 2..4 haplotypes");
    if (n_gt < 1 || n_gt > 128) return fail(c, VGMI_E_INVALID, "HMM emissions: 1..128 genotypes");
    uint8_t pos_a_buf[128], pos_b_buf[128], pos_more_buf[2][128];
    memset(pos_more_buf, 0, sizeof pos_more_buf);
    for (uint32_t g = 0; g < n_gt; ++g) {
        pos_a_buf[g] = pos[(size_t)g * ploidy];
        pos_b_buf[g] = pos[(size_t)g * ploidy + 1];
        for (uint32_t q = 2; q < ploidy; ++q) pos_more_buf[q - 2][g] = pos[(size_t)g * ploidy + q];
        for (uint32_t q = 0; q < ploidy; ++q)
            if (pos[(size_t)g * ploidy + q] >= n_used) return fail(c, VGMI_E_INVALID, "HMM emissions: a genotype names a haplotype outside the list");
    }
    const uint8_t *pos_a = pos_a_buf, *pos_b = pos_b_buf;
    const size_t n_tab = (size_t)(ploidy + 1) * 256;
    if (n_rows && (!entry_begin || !entry_count || !gt0 || !n_kept_out || !flags_out)) return fail(c, VGMI_E_INVALID, "HMM emissions: rows without their arrays");
    if (n_used < 1 || n_used > 16 || bit_len < 1 || bit_len > 6) return fail(c, VGMI_E_INVALID, "HMM emissions: 1..128 genotypes over 1..16 haplotypes, 1..6 bytes of haplotype bits");
    if (!c->d_hmm_entries) return fail(c, VGMI_E_STATE, "HMM emissions: upload the entries first");
    for (uint64_t r = 0; r < n_rows; ++r)
        if (entry_begin[r] + entry_count[r] > c->hmm_n_entries) return fail(c, VGMI_E_INVALID, "HMM emissions: a row points outside the entries");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    auto* part = new vgmi_hmm_part;
    part->c = c;
    part->n_rows = n_rows;
    part->n_gt = n_gt;
    const size_t b_obs = (size_t)(n_rows ? n_rows : 1) * n_gt * 16;
    uint8_t* d_small = nullptr;     // entry_begin | entry_count | gt0 | tables | n_kept | flags
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_eb = 0, o_ec = up(o_eb + n_rows * 8), o_g0 = up(o_ec + n_rows * 4), o_tab = up(o_g0 + n_rows * 2), o_nk = up(o_tab + n_tab * 16),
                 o_fl = up(o_nk + n_rows * 4), total = up(o_fl + n_rows) + 256;
    hipStream_t st = nullptr;
    size_t small_bytes = 0;
    part->d_obs = hmm_block_take(c, b_obs, part->obs_bytes);
    d_small = hmm_block_take(c, total, small_bytes);
    if (!part->d_obs || !d_small) {
        hmm_block_give(c, part->d_obs, part->obs_bytes);
        hmm_block_give(c, d_small, small_bytes);
        delete part;
        return fail(c, VGMI_E_NOMEM, "HMM emissions: not enough device memory");
    }
    hipError_t e = hipSuccess;
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small + o_eb, entry_begin, n_rows * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small + o_ec, entry_count, n_rows * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small + o_g0, gt0, n_rows * 2, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_small + o_tab, tables, n_tab * 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        HmmEmitParams P{};
        P.packed = c->d_hmm_entries;
        P.cov = c->d_hmm_cov;
        P.entry_begin = reinterpret_cast<const uint64_t*>(d_small + o_eb);
        P.entry_count = reinterpret_cast<const uint32_t*>(d_small + o_ec);
        P.gt0 = reinterpret_cast<const uint16_t*>(d_small + o_g0);
        P.row_lo = 0;
        P.n_gt = n_gt;
        P.n_used = n_used;
        P.bl8 = 8 * bit_len;
        memcpy(P.used, used, n_used);
        memcpy(P.pos_a, pos_a, n_gt);
        memcpy(P.pos_b, pos_b, n_gt);
        memcpy(P.pos_more, pos_more_buf, sizeof pos_more_buf);
        P.ploidy = ploidy;
        P.top_mask = top_mask;
        P.ave = ave;
        P.lower = lower;
        P.upper = upper;
        P.tables = d_small + o_tab;
        P.obs = part->d_obs;
        P.n_kept = reinterpret_cast<uint32_t*>(d_small + o_nk);
        P.flags = d_small + o_fl;
        e = launch_hmm_emissions(P, n_rows, st);
        part->emit = P;
    }
    if (e == hipSuccess && n_rows) e = hipMemcpyAsync(n_kept_out, d_small + o_nk, n_rows * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && n_rows) e = hipMemcpyAsync(flags_out, d_small + o_fl, n_rows, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    if (e != hipSuccess) {
        hmm_block_give(c, d_small, small_bytes);
        hmm_block_give(c, part->d_obs, part->obs_bytes);
        delete part;
        HIPCHK(c, e);
    }
    part->d_small = d_small;
    part->small_bytes = small_bytes;
    part->entry_count.assign(entry_count, entry_count + n_rows);
    *out = part;
    return VGMI_OK;
}

// Rows the emission launch flagged (bit 0: an under-covered multi-copy k-mer that a haplotype of the window carries -- the reference
// then consults the haplotype's sequence, src/genotype.cpp:760-800), scored again with what the host found there: entry fix_j[i] of
// row rows[r] (fix_off[r] <= i < fix_off[r + 1], ascending) loses the haplotypes of fix_mask[i] (bits over the `used` list).  The
// sequences are strings on the host; the products stay on the device.
int vgmi_hmm_part_fix_rows(vgmi_hmm_part* part, uint64_t n, const uint64_t* rows, const uint32_t* fix_off, const uint32_t* fix_j, const uint16_t* fix_mask)
{
    if (!part || (n && (!rows || !fix_off))) return VGMI_E_INVALID;
    vgmi_ctx* c = part->c;
    if (n == 0) return VGMI_OK;
    const uint32_t n_fix = fix_off[n];
    if (n_fix && (!fix_j || !fix_mask)) return VGMI_E_INVALID;
    for (uint64_t r = 0; r < n; ++r) {
        if (rows[r] >= part->n_rows || fix_off[r] > fix_off[r + 1]) return fail(c, VGMI_E_INVALID, "HMM emissions: a fixed row outside the part");
        for (uint32_t i = fix_off[r]; i < fix_off[r + 1]; ++i)
            if (fix_j[i] >= part->entry_count[rows[r]] || (i > fix_off[r] && fix_j[i] <= fix_j[i - 1]))
                return fail(c, VGMI_E_INVALID, "HMM emissions: a row's fixes must name its entries in ascending order");
    }
    HIPCHK(c, hipSetDevice(c->device));
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_off = up(n * 8), o_j = up(o_off + (n + 1) * 4), o_m = up(o_j + (size_t)n_fix * 4), total = up(o_m + (size_t)n_fix * 2) + 256;
    size_t d_bytes = 0;
    uint8_t* d = hmm_block_take(c, total, d_bytes);
    if (!d) return fail(c, VGMI_E_NOMEM, "HMM emissions: not enough device memory");
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(d, rows, n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_off, fix_off, (n + 1) * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_fix) e = hipMemcpyAsync(d + o_j, fix_j, (size_t)n_fix * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_fix) e = hipMemcpyAsync(d + o_m, fix_mask, (size_t)n_fix * 2, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        HmmEmitParams P = part->emit;
        P.fix_rows = reinterpret_cast<const uint64_t*>(d);
        P.fix_off = reinterpret_cast<const uint32_t*>(d + o_off);
        P.fix_j = reinterpret_cast<const uint32_t*>(d + o_j);
        P.fix_mask = reinterpret_cast<const uint16_t*>(d + o_m);
        e = launch_hmm_emissions(P, n, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    hmm_block_give(c, d, d_bytes);
    HIPCHK(c, e);
    return VGMI_OK;
}

int vgmi_hmm_part_set_rows(vgmi_hmm_part* part, uint64_t n, const uint64_t* rows, const void* obs_rows)
{
    if (!part || (n && (!rows || !obs_rows))) return VGMI_E_INVALID;
    vgmi_ctx* c = part->c;
    for (uint64_t i = 0; i < n; ++i)
        if (rows[i] >= part->n_rows) return fail(c, VGMI_E_INVALID, "HMM emissions: a row outside the part");
    if (n == 0) return VGMI_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b_obs = (size_t)n * part->n_gt * 16, o_rows = (b_obs + 255) & ~(size_t)255;
    size_t d_bytes = 0;
    uint8_t* d = hmm_block_take(c, o_rows + n * 8, d_bytes);
    if (!d) return fail(c, VGMI_E_NOMEM, "HMM emissions: not enough device memory");
    hipStream_t st = nullptr;      // a stream of its own: other parts' work on this device is not waited for
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(d, obs_rows, b_obs, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_rows, rows, n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_hmm_scatter_rows(part->d_obs, reinterpret_cast<const uint64_t*>(d + o_rows), d, part->n_gt, n, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    hmm_block_give(c, d, d_bytes);
    HIPCHK(c, e);
    return VGMI_OK;
}

int vgmi_hmm_part_calls(vgmi_hmm_part* part, uint32_t ploidy, const uint8_t* keep, uint32_t n_windows, const uint32_t* row, const uint8_t* restart,
                        const void* pow, uint64_t n_steps, const void* uniform, const vgmi_hmm_chain* chains, uint32_t n_chains, const uint8_t* gid,
                        const uint8_t* order, const uint64_t* fwd_step, const uint64_t* bwd_step, void* prob, uint32_t* winner)
{
    if (!part || !gid || !order || !fwd_step || !bwd_step || !prob || !winner) return VGMI_E_INVALID;
    return hmm_run(part->c, part->n_gt, ploidy, keep, n_windows, nullptr, 0, part->n_rows, row, restart, pow, 0, n_steps, uniform, chains, n_chains,
                   nullptr, gid, order, fwd_step, bwd_step, prob, winner, part->d_obs);
}

// ---- a part's recursion inputs kept on the device (round 5).  Everything hmm_run uploads but the emission scores -- keep matrix, step
// tables (pow), rows, restarts, chains, genotype strings' ids and order, the rows' steps: 230 MB per chr20-scale sample -- is a
// function of the graph and the options, not of the sample: a plan holds it on the device, made once, used by every sample (and every
// context of the device: the block is plain device memory, not a context's pool).
struct vgmi_hmm_plan {
    int device = 0;
    uint8_t* d = nullptr;
    uint32_t n_gt = 0, ploidy = 0, n_chains = 0;
    uint64_t n_rows = 0, n_steps = 0;
    size_t o_keep = 0, o_row = 0, o_rs = 0, o_pow = 0, o_uni = 0, o_ch = 0, o_gid = 0, o_ord = 0, o_fs = 0, o_bs = 0, bytes = 0;
};

int vgmi_hmm_plan_create(vgmi_ctx* c, uint32_t n_gt, uint32_t ploidy, const uint8_t* keep, uint32_t n_windows, uint64_t n_rows, const uint32_t* row,
                         const uint8_t* restart, const void* pow, uint64_t n_steps, const void* uniform, const vgmi_hmm_chain* chains, uint32_t n_chains,
                         const uint8_t* gid, const uint8_t* order, const uint64_t* fwd_step, const uint64_t* bwd_step, vgmi_hmm_plan** out)
{
    if (!c || !out) return VGMI_E_INVALID;
    *out = nullptr;
    if (!keep || !row || !restart || !pow || !uniform || !chains || !gid || !order || !fwd_step || !bwd_step) return VGMI_E_INVALID;
    if (n_gt < 1 || n_gt > VGMI_HMM_MAX_GT || ploidy < 1 || ploidy > 4) return fail(c, VGMI_E_INVALID, "HMM plan: 1..2048 genotypes of 1..4 haplotypes");
    if (n_gt > 128)
        for (uint32_t w = 0; w < n_windows; ++w) {
            const uint8_t* m = keep + (size_t)w * n_gt * n_gt;
            for (uint32_t i = 0; i < n_gt; ++i)
                for (uint32_t j = i + 1; j < n_gt; ++j)
                    if (m[(size_t)i * n_gt + j] != m[(size_t)j * n_gt + i]) return fail(c, VGMI_E_INVALID, "HMM plan: keep matrix not symmetric");
        }
    if (n_steps == 0 || n_chains == 0 || n_rows == 0) return fail(c, VGMI_E_INVALID, "HMM plan: nothing to plan");
    for (uint32_t i = 0; i < n_chains; ++i)
        if (chains[i].keep_index >= n_windows || chains[i].first_step + chains[i].n_steps > n_steps) return fail(c, VGMI_E_INVALID, "HMM plan: a chain points outside its arrays");
    for (uint64_t s = 0; s < n_steps; ++s)
        if (row[s] >= n_rows) return fail(c, VGMI_E_INVALID, "HMM plan: a step points outside the emission rows");
    for (uint64_t i = 0; i < n_rows; ++i)
        if (fwd_step[i] >= n_steps || bwd_step[i] >= n_steps) return fail(c, VGMI_E_INVALID, "HMM plan: a row points outside the steps");
    HIPCHK(c, hipSetDevice(c->device));
    auto* pl = new vgmi_hmm_plan;
    pl->device = c->device;
    pl->n_gt = n_gt;
    pl->ploidy = ploidy;
    pl->n_chains = n_chains;
    pl->n_rows = n_rows;
    pl->n_steps = n_steps;
    const uint32_t stride = ploidy + 1;
    const size_t b_keep = (size_t)n_windows * n_gt * n_gt, b_row = (size_t)n_steps * 4, w_pow = (size_t)2 * stride * 16, b_pow = (size_t)n_steps * w_pow,
                 b_ch = (size_t)n_chains * sizeof(vgmi_hmm_chain), b_gid = (size_t)n_rows * n_gt, b_fs = (size_t)n_rows * 8;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    pl->o_keep = 0;
    pl->o_row = up(pl->o_keep + b_keep);
    pl->o_rs = up(pl->o_row + b_row);
    pl->o_pow = up(pl->o_rs + n_steps);
    pl->o_uni = up(pl->o_pow + b_pow);
    pl->o_ch = pl->o_uni + 256;
    pl->o_gid = up(pl->o_ch + b_ch);
    pl->o_ord = up(pl->o_gid + b_gid);
    pl->o_fs = up(pl->o_ord + b_gid);
    pl->o_bs = up(pl->o_fs + b_fs);
    pl->bytes = up(pl->o_bs + b_fs);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&pl->d), pl->bytes);
    if (e != hipSuccess) {
        delete pl;
        (void)hipGetLastError();
        return fail(c, VGMI_E_NOMEM, "HMM plan: not enough device memory");
    }
    e = hipMemcpy(pl->d + pl->o_keep, keep, b_keep, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_row, row, b_row, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_rs, restart, n_steps, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_pow, pow, b_pow, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_uni, uniform, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_ch, chains, b_ch, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_gid, gid, b_gid, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_ord, order, b_gid, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_fs, fwd_step, b_fs, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d + pl->o_bs, bwd_step, b_fs, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(pl->d);
        delete pl;
        HIPCHK(c, e);
    }
    *out = pl;
    return VGMI_OK;
}

void vgmi_hmm_plan_free(vgmi_hmm_plan* pl)
{
    if (!pl) return;
    if (pl->d && hipSetDevice(pl->device) == hipSuccess) (void)hipFree(pl->d);
    delete pl;
}

// recursion and posterior of a part on the inputs of a plan and the part's own emission scores: what comes back is the calls
int vgmi_hmm_part_calls_plan(vgmi_hmm_part* part, const vgmi_hmm_plan* pl, void* prob, uint32_t* winner)
{
    if (!part || !pl || !prob || !winner) return VGMI_E_INVALID;
    vgmi_ctx* c = part->c;
    if (pl->device != c->device || pl->n_gt != part->n_gt || pl->n_rows != part->n_rows) return fail(c, VGMI_E_INVALID, "HMM plan: made for another part");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t w_obs = (size_t)pl->n_gt * 16, b_out = (size_t)pl->n_steps * w_obs, b_prob = (size_t)pl->n_rows * 16, b_win = (size_t)pl->n_rows * 4;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_prob = up(b_out), o_win = up(o_prob + b_prob), total = up(o_win + b_win);
    size_t d_bytes = 0;
    uint8_t* d = hmm_block_take(c, total, d_bytes);
    if (!d) return fail(c, VGMI_E_NOMEM, "HMM recursion: not enough device memory");
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) {
        HmmParams P{};
        P.n_gt = pl->n_gt;
        P.ploidy = pl->ploidy;
        P.keep = pl->d + pl->o_keep;
        P.obs = part->d_obs;
        P.row = reinterpret_cast<const uint32_t*>(pl->d + pl->o_row);
        P.restart = pl->d + pl->o_rs;
        P.pow = pl->d + pl->o_pow;
        P.uniform = pl->d + pl->o_uni;
        P.chains = reinterpret_cast<const HmmChain*>(pl->d + pl->o_ch);
        P.out = d;
        e = launch_hmm_recursion(P, pl->n_chains, st);
    }
    if (e == hipSuccess) {
        HmmPostParams Q{};
        Q.n_gt = pl->n_gt;
        Q.row0 = 0;
        Q.ab = d;
        Q.fwd_step = reinterpret_cast<const uint64_t*>(pl->d + pl->o_fs);
        Q.bwd_step = reinterpret_cast<const uint64_t*>(pl->d + pl->o_bs);
        Q.gid = pl->d + pl->o_gid;
        Q.order = pl->d + pl->o_ord;
        Q.prob = d + o_prob;
        Q.winner = reinterpret_cast<uint32_t*>(d + o_win);
        e = launch_hmm_posterior(Q, pl->n_rows, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(prob, d + o_prob, b_prob, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(winner, d + o_win, b_win, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    hmm_block_give(c, d, d_bytes);
    HIPCHK(c, e);
    return VGMI_OK;
}

int vgmi_hmm_tallies(vgmi_ctx* c, uint64_t n_rows, const uint64_t* entry_begin, const uint32_t* entry_count, const uint32_t* winner, uint32_t n_gt,
                     const uint8_t* hap_ab, uint32_t n_hap, uint64_t sel_mask, uint32_t* out, uint8_t* unique_out)
{
    if (!c || (n_rows && (!entry_begin || !entry_count || !winner || !hap_ab || !out || !unique_out)) || n_gt > 128) return VGMI_E_INVALID;
    if (!c->d_hmm_entries || !c->d_hmm_cov) return fail(c, VGMI_E_STATE, "HMM tallies: upload the entries and the sample's coverage first");
    if (n_rows == 0) return VGMI_OK;
    for (uint64_t i = 0; i < n_rows; ++i)
        if (entry_begin[i] + entry_count[i] > c->hmm_n_entries) return fail(c, VGMI_E_INVALID, "HMM tallies: a row's entries lie outside the uploaded lists");
    HIPCHK(c, hipSetDevice(c->device));
    // one block: entry_begin | entry_count | winner | out | unique | hap_ab
    const size_t o_cnt = n_rows * 8, o_win = o_cnt + n_rows * 4, o_out = o_win + n_rows * 4, o_uni = o_out + n_rows * 16, o_hap = (o_uni + n_rows + 255) & ~(size_t)255;
    size_t d_bytes = 0;
    uint8_t* d = hmm_block_take(c, o_hap + 256, d_bytes);
    if (!d) return fail(c, VGMI_E_NOMEM, "HMM tallies: not enough device memory");
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(d, entry_begin, n_rows * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_cnt, entry_count, n_rows * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_win, winner, n_rows * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d + o_hap, hap_ab, 2 * (size_t)n_gt, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = launch_hmm_tally(reinterpret_cast<const unsigned long long*>(c->d_hmm_entries), c->d_hmm_cov, reinterpret_cast<const uint64_t*>(d),
                             reinterpret_cast<const uint32_t*>(d + o_cnt), reinterpret_cast<const uint32_t*>(d + o_win), d + o_hap, n_gt, n_hap, sel_mask, n_rows,
                             reinterpret_cast<uint32_t*>(d + o_out), d + o_uni, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + o_out, n_rows * 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(unique_out, d + o_uni, n_rows, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    hmm_block_give(c, d, d_bytes);
    HIPCHK(c, e);
    return VGMI_OK;
}

int vgmi_hmm_part_fetch(vgmi_hmm_part* part, void* obs_out)
{
    if (!part || !obs_out) return VGMI_E_INVALID;
    vgmi_ctx* c = part->c;
    HIPCHK(c, hipSetDevice(c->device));
    if (part->n_rows) HIPCHK(c, hipMemcpy(obs_out, part->d_obs, (size_t)part->n_rows * part->n_gt * 16, hipMemcpyDeviceToHost));
    return VGMI_OK;
}

void vgmi_hmm_part_free(vgmi_hmm_part* part)
{
    if (!part) return;
    hmm_block_give(part->c, part->d_obs, part->obs_bytes);      // kept for the next part / sample (hipFree would wait for every stream)
    hmm_block_give(part->c, part->d_small, part->small_bytes);
    delete part;
}

int vgmi_device_of(vgmi_ctx* c, int* device)
{
    if (!c || !device) return VGMI_E_INVALID;
    *device = c->device;
    return VGMI_OK;
}

int vgmi_device_memory(vgmi_ctx* c, size_t* free_bytes, size_t* total_bytes)
{
    if (!c) return VGMI_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(c, hipMemGetInfo(&f, &t));
    {   // blocks the context keeps between HMM calls and is not using: the next call takes them or frees them for a larger one
        std::lock_guard<std::mutex> lock(c->hmm_mu);
        for (const auto& b : c->hmm_blocks) f += b.second;
    }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return VGMI_OK;
}

int vgmi_hmm_recursion(vgmi_ctx* c, uint32_t n_gt, uint32_t ploidy, const uint8_t* keep, uint32_t n_windows, const void* obs,
                       uint64_t n_rows, const uint32_t* row, const uint8_t* restart, const void* pow, uint64_t n_steps,
                       const void* uniform, const vgmi_hmm_chain* chains, uint32_t n_chains, void* out)
{
    if (!out) return VGMI_E_INVALID;
    return hmm_run(c, n_gt, ploidy, keep, n_windows, obs, 0, n_rows, row, restart, pow, 0, n_steps, uniform, chains, n_chains, out, nullptr,
                   nullptr, nullptr, nullptr, nullptr, nullptr);
}

int vgmi_hmm_calls(vgmi_ctx* c, uint32_t n_gt, uint32_t ploidy, const uint8_t* keep, uint32_t n_windows, const void* obs, uint64_t n_rows,
                   const uint32_t* row, const uint8_t* restart, const void* pow, uint64_t n_steps, const void* uniform,
                   const vgmi_hmm_chain* chains, uint32_t n_chains, const uint8_t* gid, const uint8_t* order, const uint64_t* fwd_step,
                   const uint64_t* bwd_step, void* prob, uint32_t* winner, void* alpha_beta_or_null)
{
    if (!gid || !order || !fwd_step || !bwd_step || !prob || !winner) return VGMI_E_INVALID;
    return hmm_run(c, n_gt, ploidy, keep, n_windows, obs, 0, n_rows, row, restart, pow, 0, n_steps, uniform, chains, n_chains, alpha_beta_or_null,
                   gid, order, fwd_step, bwd_step, prob, winner);
}

int vgmi_hmm_calls_part(vgmi_ctx* c, uint32_t n_gt, uint32_t ploidy, const uint8_t* keep, uint32_t n_windows, const void* obs, uint64_t row_lo,
                        uint64_t row_hi, const uint32_t* row, const uint8_t* restart, const void* pow, uint64_t step_lo, uint64_t step_hi,
                        const void* uniform, const vgmi_hmm_chain* chains, uint32_t n_chains, const uint8_t* gid, const uint8_t* order,
                        const uint64_t* fwd_step, const uint64_t* bwd_step, void* prob, uint32_t* winner)
{
    if (!gid || !order || !fwd_step || !bwd_step || !prob || !winner) return VGMI_E_INVALID;
    return hmm_run(c, n_gt, ploidy, keep, n_windows, obs, row_lo, row_hi, row, restart, pow, step_lo, step_hi, uniform, chains, n_chains, nullptr,
                   gid, order, fwd_step, bwd_step, prob, winner);
}

// BloomFilter::save / load (src/counting_bloom_filter.cpp:126-190): u64 size | u32 numHashes | numHashes x u64 seed | size bytes
int vgmi_bloom_save_file(vgmi_ctx* c, const char* path)
{
    if (!c || !path) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    std::vector<uint8_t> filt(c->bv.m);
    int rc = vgmi_bloom_fetch(c, filt.data());
    if (rc != VGMI_OK) return rc;
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail(c, VGMI_E_INVALID, std::string("'") + path + "': No such file or directory.");
    const uint64_t m = c->bv.m;
    const uint32_t nh = c->bv.n_hash;
    bool ok = fwrite(&m, 8, 1, fp) == 1 && fwrite(&nh, 4, 1, fp) == 1 && fwrite(c->bloom_seeds64, 8, nh, fp) == nh &&
              fwrite(filt.data(), 1, filt.size(), fp) == filt.size();
    if (fclose(fp) != 0) ok = false;
    return ok ? VGMI_OK : fail(c, VGMI_E_INVALID, std::string("'") + path + "': write error.");
}

int vgmi_bloom_load_file(vgmi_ctx* c, const char* path)
{
    if (!c || !path) return VGMI_E_INVALID;
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(c, VGMI_E_INVALID, std::string("'") + path + "': No such file or directory.");
    uint64_t m = 0, seeds[VG_BLOOM_MAX_HASH];
    uint32_t nh = 0;
    bool ok = fread(&m, 8, 1, fp) == 1 && fread(&nh, 4, 1, fp) == 1 && nh >= 1 && nh <= VG_BLOOM_MAX_HASH && m > 0 &&
              fread(seeds, 8, nh, fp) == nh;
    std::vector<uint8_t> filt;
    if (ok) {
        const long at = ftell(fp);
        ok = at >= 0 && fseek(fp, 0, SEEK_END) == 0 && (uint64_t)(ftell(fp) - at) == m && fseek(fp, at, SEEK_SET) == 0;   // sized by the file, not by its header
        if (ok) {
            filt.resize(m);
            ok = fread(filt.data(), 1, filt.size(), fp) == filt.size();
        }
    }
    fclose(fp);
    if (!ok) return fail(c, VGMI_E_INVALID, std::string("'") + path + "': not a counting Bloom filter file.");
    int rc = vgmi_bloom_create(c, m, nh, seeds);
    if (rc != VGMI_OK) return rc;
    return vgmi_bloom_load(c, filt.data());
}

int vgmi_bloom_fetch(vgmi_ctx* c, uint8_t* out)
{
    if (!c || !out) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->bv.filter, c->bv.m, hipMemcpyDeviceToHost));
    return VGMI_OK;
}

int vgmi_bloom_load(vgmi_ctx* c, const uint8_t* in)
{
    if (!c || !in) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->bv.filter, in, c->bv.m, hipMemcpyHostToDevice));
    return VGMI_OK;
}

int vgmi_bloom_query(vgmi_ctx* c, const uint64_t* keys, size_t n, uint8_t* min_out, uint8_t* nz_out)
{
    if (!c) return VGMI_E_INVALID;
    if (!c->has_bloom) return fail(c, VGMI_E_STATE, "no Bloom filter");
    if (n == 0) return VGMI_OK;
    if (!keys) return fail(c, VGMI_E_INVALID, "keys is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t* d_k = nullptr;
    uint8_t* d_o = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_k), n * 8));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), 2 * n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_k, keys, n * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_bloom_query(c->bv, d_k, n, d_o, d_o + n, c->stream);
    if (e == hipSuccess && min_out) e = hipMemcpyAsync(min_out, d_o, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && nz_out) e = hipMemcpyAsync(nz_out, d_o + n, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_k);
    if (d_o) (void)hipFree(d_o);
    HIPCHK(c, e);
    return VGMI_OK;
}

/* ---------------------------------------------------------------- tooling */

int vgmi_synth_reads_device(vgmi_ctx* c, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                            const char* dev_hap_cat, const uint64_t* hap_off, uint32_t n_hap, char* dev_out)
{
    if (!c || !dev_hap_cat || !hap_off || !dev_out) return VGMI_E_INVALID;
    if (n_hap < 1 || n_hap > VG_SYNTH_MAX_HAPS) return fail(c, VGMI_E_INVALID, "1..8 haplotypes");
    if (read_len < 1 || read_len > VGS_INSERT) return fail(c, VGMI_E_INVALID, "read_len must be in 1..350");
    SynthHaps h{};
    h.n = n_hap;
    for (uint32_t i = 0; i < n_hap; ++i) {
        h.off[i] = hap_off[i];
        h.len[i] = hap_off[i + 1] - hap_off[i];
        if (h.len[i] < VGS_INSERT) return fail(c, VGMI_E_INVALID, "haplotype shorter than the insert size");
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_synth_reads(seed, first_read, n_reads, read_len, dev_hap_cat, h, dev_out, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VGMI_OK;
}

int vgmi_synth_reads_host(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len, const char* hap_cat,
                          const uint64_t* hap_off, uint32_t n_hap, char* out)
{
    if (!hap_cat || !hap_off || !out || n_hap < 1 || n_hap > VG_SYNTH_MAX_HAPS) return VGMI_E_INVALID;
    if (read_len < 1 || read_len > VGS_INSERT) return VGMI_E_INVALID;
    const char* hp[VG_SYNTH_MAX_HAPS];
    uint64_t hl[VG_SYNTH_MAX_HAPS];
    for (uint32_t i = 0; i < n_hap; ++i) {
        hp[i] = hap_cat + hap_off[i];
        hl[i] = hap_off[i + 1] - hap_off[i];
        if (hl[i] < VGS_INSERT) return VGMI_E_INVALID;
    }
    for (uint64_t r = 0; r < n_reads; ++r) {
        char* o = out + r * (read_len + 1);
        for (uint32_t j = 0; j < read_len; ++j) o[j] = vgs_read_base(seed, first_read + r, j, read_len, hp, hl, n_hap);
        o[read_len] = '\n';
    }
    return VGMI_OK;
}

int vgmi_synth_snp_keys_host(const char* ref, uint64_t ref_len, const uint64_t* pos, const char* alts, uint64_t n_sites,
                             uint32_t k, uint64_t* keys_out)
{
    if (!ref || !pos || !alts || !keys_out || k < 1 || k > 28) return VGMI_E_INVALID;
    const uint64_t mask = k == 32 ? ~0ULL : (1ULL << (2 * k)) - 1;
    for (uint64_t i = 0; i < n_sites; ++i) {
        const uint64_t p = pos[i];
        if (p < k - 1 || p + k > ref_len) return VGMI_E_INVALID;
        for (uint32_t allele = 0; allele < 2; ++allele) {
            uint64_t fwd = 0, rc = 0;
            uint64_t* out = keys_out + (2 * i + allele) * k;
            for (uint64_t q = p - (k - 1); q <= p + (k - 1); ++q) {
                const uint32_t c = vg_nt4((unsigned char)(allele && q == p ? alts[i] : ref[q]));
                if (c > 3) return VGMI_E_INVALID;
                fwd = (fwd << 2 | c) & mask;
                rc = (rc >> 2) | (uint64_t)(3u ^ c) << (2 * (k - 1));
                if (q >= p) out[q - p] = vg_hash64(fwd < rc ? fwd : rc, mask) << 8 | k;
            }
        }
    }
    return VGMI_OK;
}

int vgmi_synth_reference_host(uint64_t seed, uint64_t len, char* out)
{
    if (!out) return VGMI_E_INVALID;
    for (uint64_t i = 0; i < len; ++i) out[i] = vgs_ref_base(seed, i);
    return VGMI_OK;
}

}  // extern "C"
