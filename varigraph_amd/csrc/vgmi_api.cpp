// vgmi_api.cpp -- the C ABI of include/vgmi.h over the gfx950 kernels (vgmi_kernels.hip): contexts, read counting, read-out.
//
// Host-side plumbing only: contexts, device memory, pinned double-buffered staging, streams and
// events.  There is no CPU implementation of any compute path in here -- without a HIP device
// vgmi_create fails with VGMI_E_NO_DEVICE.  The other parts of the ABI: vgmi_ctx.h.
#include "vgmi_ctx.h"

namespace vgapi {

thread_local std::string g_create_error;

// The last error of a context is read by the thread that got the failing return code -- and a context serves several threads at once
// (FASTQ streams, HMM parts, the --procs broadcast thread): the message is kept per THREAD as well as in the context (the
// broadcast thread's text no longer lands where a counting thread's vgmi_last_error may be reading).
thread_local uint64_t t_err_ctx = 0;      // vgmi_ctx::id (not the address: a later context may be allocated where an earlier one was)
thread_local std::string t_err_msg;

int fail(vgmi_ctx* c, int code, const std::string& msg)
{
    static std::mutex mu;       // vgmi_hmm_calls_part may fail on several threads of one context
    std::lock_guard<std::mutex> lock(mu);
    if (c) {
        c->err = msg;
        t_err_ctx = c->id;
        t_err_msg = msg;
    } else g_create_error = msg;
    return code;
}

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
// VGMI_CT_DEFER=0|1 (A/B), VGMI_CT_DEFER_MIN: the smallest block in bytes that defers (default 512 MiB: below).
int ctd_prepare(vgmi_ctx* c, size_t n_bytes, hipStream_t st, CtDefer* d)
{
    // (read per launch: the test matrix switches them inside one process)
    const char* const e_on = getenv("VGMI_CT_DEFER");
    const int on = e_on ? atoi(e_on) : VGMI_CT_DEFER_DEFAULT;
    const char* const e_min = getenv("VGMI_CT_DEFER_MIN");
    // the smallest block that defers: the second pass costs ~70 us whatever the block holds (three launches; 1 280 regions zeroed, added up and
    // scanned) and saves a tenth of the row loop's time, 0.03 us per thousand reads -- even at ~2.6e6 reads = 0.4 GB of packed reads.  Measured
    // through the CLI (tools/history/gpu_r6_d.sh, eight chr20 samples, ~100 MiB pieces on two streams): count passes of a sample 0.020-0.027 s in the
    // row loop, 0.045-0.056 s deferred; the command's wall is the same (counting is ingest-bound).  So pieces of a FASTQ stream keep their
    // atomics and blocks of half a gigabyte or more -- a sample resident in HBM -- defer them.
    const size_t min_bytes = e_min ? (size_t)atoll(e_min) : (size_t)512 << 20;
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

}  // namespace vgapi


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

namespace vgapi {
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
                 hipStream_t st, const unsigned long long* n_bytes_dev)
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

}  // namespace vgapi

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
    static std::atomic<uint64_t> next_id{1};
    c->id = next_id.fetch_add(1);
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

const char* vgmi_last_error(const vgmi_ctx* c)
{
    if (!c) return g_create_error.c_str();
    if (t_err_ctx == c->id && !t_err_msg.empty()) return t_err_msg.c_str();      // this thread's own last failure on this context
    return c->err.c_str();
}

void* vgmi_stream(vgmi_ctx* c) { return c ? (void*)c->stream : nullptr; }

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

}  // extern "C"
