// vgmi_api_fastq.cpp -- FASTQ text on the device (vgmi_fastq_*, vgmi_gunzip_buffer): records found by vgmi_fastq.hip, block-gzip members
// inflated by vgmi_inflate.hip, ordinary gzip by vgmi_gunzip.hip
#include "vgmi_ctx.h"

extern "C" {

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
}
extern "C++" void vgapi::fastq_free(vgmi_fastq* f)
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

}  // extern "C"
