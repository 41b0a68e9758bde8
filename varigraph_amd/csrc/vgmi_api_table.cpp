// vgmi_api_table.cpp -- the table image (include/vgmi.h: vgmi_table_*): its layout, upload / import / export / snapshot / clone, and the tables
// the count kernels walk, built from it on every device: path table (graphs of <= 65 536 k-mers), context table, grid-16-mer table.
#include "vgmi_ctx.h"

namespace vgapi {

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
    // ... and k = 26 at any size: its runs of k + 7 bases do not fit the path-table kernel's two words, the context table's flanks of 10 do;
    // k = 28 (round 6) likewise: flanks of 11, eleven windows an entry (vgmi_ctable.h)
    const bool largek = ((k >= 19 && k <= 25 && n_keys > VG_GRID_LDS_MAX_KEYS) || ((k == 26 || k == 28) && n_keys > 0)) && n_keys < (1ULL << 31) - 16 && !off_k;
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
    if ((e && e[0] == '0') || h.slot_bytes != 8 || (h.n_keys <= VG_GRID_LDS_MAX_KEYS && !((h.k == 26 || h.k == 28) && h.n_keys > 0))) return false;
    // k = 19 .. 25, odd (round 5): the context table only (flanks of k - 16 bases, vgmi_ctable.h); VGMI_CTABLE_K=0 keeps them on the generic kernel (A/B)
    if ((h.k >= 19 && h.k <= 26) || h.k == 28) {      // (even k too: the pass that takes back what the reference's run counter suppresses runs ahead of the kernel)
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
    // a unitig of L k-mers holds L + k - 16 occurrences (palindromic 16-mers: two entries, rare); k = 28: its first and its last one have no window an entry can hold
    c->ct_entries = n + (h.k == 28 ? 10 : h.k - 16) * cur[1];
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

}  // namespace vgapi

extern "C" {

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

}  // extern "C"
