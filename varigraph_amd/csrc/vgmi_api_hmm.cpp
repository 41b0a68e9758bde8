// vgmi_api_hmm.cpp -- the HMM on the device (vgmi_hmm_*): emission scores, recursion, posterior, tallies (kernels: vgmi_hmm.hip)
#include "vgmi_ctx.h"

extern "C" {

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

}  // extern "C"
