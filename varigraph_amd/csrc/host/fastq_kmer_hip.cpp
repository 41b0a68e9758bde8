#include "fastq_kmer_hip.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "fast_inflate.hpp"
#include "fastx_reader.hpp"
#include "vgmi.h"

namespace vgh {

namespace {

struct Block {
    std::vector<char> bytes;
    size_t n_reads = 0;
    uint64_t read_base = 0;
};

struct Channel {  // parser threads -> submitting thread
    std::mutex mu;
    std::condition_variable cv_full, cv_free;
    std::deque<std::unique_ptr<Block>> full;
    std::vector<std::unique_ptr<Block>> free_list;
    size_t producers = 0;
    std::string error;
};

void parse_file(const std::string& path, Channel& ch, size_t block_bytes, unsigned decode_threads)
{
    auto grab = [&]() {
        std::unique_lock<std::mutex> lk(ch.mu);
        ch.cv_free.wait(lk, [&] { return !ch.free_list.empty(); });
        auto b = std::move(ch.free_list.back());
        ch.free_list.pop_back();
        b->bytes.clear();
        b->n_reads = 0;
        b->read_base = 0;
        return b;
    };
    auto push = [&](std::unique_ptr<Block> b) {
        std::lock_guard<std::mutex> lk(ch.mu);
        ch.full.push_back(std::move(b));
        ch.cv_full.notify_one();
    };
    try {
        FastxReader rd(path, decode_threads);
        std::unique_ptr<Block> cur = grab();
        while (rd.next() >= 0) {  // stops at EOF (-1) and at the first truncated record (-2)
            const std::string& s = rd.seq();
            // `string sequence = ks->seq.s` (src/fastq_kmer.cpp:101) ends at the first NUL, while
            // mReadBase adds the full ks->seq.l (:105)
            const size_t len = strnlen(s.data(), s.size());
            if (len == 0) throw std::runtime_error("'" + path + "': empty read sequence (the reference aborts on assert(len > 0), kmer.cpp:124)");
            if (cur->bytes.size() + len + 1 > block_bytes && cur->n_reads) {
                push(std::move(cur));
                cur = grab();
            }
            cur->bytes.insert(cur->bytes.end(), s.data(), s.data() + len);
            cur->bytes.push_back('\n');
            cur->n_reads++;
            cur->read_base += s.size();
        }
        if (cur->n_reads) push(std::move(cur));
        else {
            std::lock_guard<std::mutex> lk(ch.mu);
            ch.free_list.push_back(std::move(cur));
        }
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(ch.mu);
        if (ch.error.empty()) ch.error = e.what();
    }
    std::lock_guard<std::mutex> lk(ch.mu);
    ch.producers--;
    ch.cv_full.notify_all();
}

// ---- device-side parsing (vgmi_fastq_*): the host moves file text, the device finds the records ----------------------

// records of `rd` (from a record boundary on) through the host reader into read blocks and vgmi_reads_submit; the
// context's host-block path is single-threaded, hence the mutex shared by the files of one sample
void host_leg(vgmi_ctx* ctx, FastxReader& rd, const std::string& path, size_t block_bytes, std::mutex& submit_mu, uint64_t& n_reads,
              uint64_t& read_base)
{
    std::vector<char> block;
    block.reserve(block_bytes + 1024);
    size_t in_block = 0;
    auto flush = [&]() {
        if (!in_block) return;
        std::lock_guard<std::mutex> lk(submit_mu);
        if (vgmi_reads_submit(ctx, block.data(), block.size(), nullptr, in_block) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
        block.clear();
        in_block = 0;
    };
    while (rd.next() >= 0) {
        const std::string& s = rd.seq();
        const size_t len = strnlen(s.data(), s.size());   // src/fastq_kmer.cpp:101 vs :105, see parse_file
        if (len == 0) throw std::runtime_error("'" + path + "': empty read sequence (the reference aborts on assert(len > 0), kmer.cpp:124)");
        if (block.size() + len + 1 > block_bytes && in_block) flush();
        block.insert(block.end(), s.data(), s.data() + len);
        block.push_back('\n');
        ++in_block;
        ++n_reads;
        read_base += s.size();
    }
    flush();
}

// plain file: `threads` readers fill one pinned buffer side by side (a single read(2) stream moves 3-6 GB/s, the link
// to the device ten times that)
size_t fill_plain(int fd, uint64_t offset, uint64_t file_size, char* buf, size_t cap, unsigned threads)
{
    const size_t want = (size_t)std::min<uint64_t>(cap, file_size - offset);
    if (want == 0) return 0;
    const unsigned n_thr = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, want >> 22));   // >= 4 MiB per reader
    std::atomic<bool> bad{false};
    auto part = [&](unsigned t) {
        size_t b = want / n_thr * t, e = t + 1 == n_thr ? want : want / n_thr * (t + 1);
        while (b < e) {
            const ssize_t r = ::pread(fd, buf + b, e - b, (off_t)(offset + b));
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) { bad = true; return; }
            b += (size_t)r;
        }
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < n_thr; ++t) ts.emplace_back(part, t);
    part(0);
    for (auto& t : ts) t.join();
    if (bad) throw std::runtime_error("read error");
    return want;
}

// What the host decoder (fast_inflate.cpp: gzread's semantics) makes of the whole file: how it ends and how much text it yields
GunzipEnd host_gunzip_verdict(const std::string& path, uint64_t& n_text)
{
    n_text = 0;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return GunzipEnd::Truncated;
    std::vector<unsigned char> out((size_t)32768 + ((size_t)4 << 20) + 512);      // window in front, slack behind (as byte_source.cpp's chunks)
    GunzipIO io;
    io.read = [&](unsigned char* dst, size_t n) -> size_t {
        for (;;) {
            const ssize_t r = ::read(fd, dst, n);
            if (r < 0 && errno == EINTR) continue;
            return r < 0 ? 0 : (size_t)r;
        }
    };
    io.next_buffer = [&](size_t, size_t) { return out.data() + 32768; };
    io.commit = [&](size_t n) { n_text += n; };
    const GunzipEnd end = fast_gunzip(io, (size_t)4 << 20);
    ::close(fd);
    return end;
}

struct DeviceFileResult {
    uint64_t n_reads = 0, read_base = 0;
};

DeviceFileResult device_file(vgmi_ctx* ctx, const std::string& path, size_t block_bytes, unsigned threads, std::mutex& submit_mu)
{
    DeviceFileResult res;
    // plain or compressed?  (gzopen's transparent mode: anything that does not start with the gzip magic is plain)
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("'" + path + "': No such file or directory.");
    unsigned char magic[2] = {0, 0};
    const ssize_t got = ::pread(fd, magic, 2, 0);
    const bool plain = !(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { ::close(fd); throw std::runtime_error("'" + path + "': cannot stat"); }
    const uint64_t file_size = (uint64_t)sb.st_size;
    struct FdGuard { int fd; ~FdGuard() { if (fd >= 0) ::close(fd); } } guard{fd};
    // A named pipe or `<(zcat ...)` (the reference reads whatever gzopen opens, include/kseq.h:59-72): one pass, no offsets, so
    // neither the side-by-side readers nor a host reader resuming in the middle apply -- the host reader takes the whole stream.
    if (!S_ISREG(sb.st_mode)) {
        ::close(guard.fd);
        guard.fd = -1;
        FastxReader rd(ByteSource::open(path, threads));
        host_leg(ctx, rd, path, block_bytes, submit_mu, res.n_reads, res.read_base);
        return res;
    }

    // block gzip?  (the first member carries the 'BC' extra subfield: bgzip, htslib)
    bool bgzf = false;
    if (!plain) {
        unsigned char h[18] = {0};
        if (::pread(fd, h, 18, 0) == 18 && h[2] == 8 && h[3] == 4 && (h[10] | h[11] << 8) >= 6 && h[12] == 'B' && h[13] == 'C') bgzf = true;
        const char* off = std::getenv("VGH_HOST_INFLATE");   // A/B: block gzip through the host's inflate workers
        if (off && off[0] == '1') bgzf = false;
    }

    // ordinary gzip: inflated on the device too (vgmi_gunzip.hip; VGH_DEVICE_GUNZIP=0: by the host's inflate threads, the A/B and
    // the path that takes over wherever the device gives a stream up)
    bool dev_gunzip = !plain && !bgzf;
    if (const char* e = std::getenv("VGH_DEVICE_GUNZIP"))
        if (e[0] == '0') dev_gunzip = false;
    bool gz_gave_up = false;

    vgmi_fastq* fq = nullptr;
    if (vgmi_fastq_open(ctx, &fq) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
    uint64_t n_rec = 0, n_bases = 0, consumed = 0;
    int stopped = 0;
    std::vector<char> tail(1u << 20);
    size_t tail_len = 0;
    bool closed = false;
    int inflate_failed = 0;
    uint64_t comp_taken = 0;          // block gzip: compressed bytes handed to the device as whole members
    uint64_t comp_good = 0;           // ... of which the device vouches for (everything unless a member failed)
    auto close_stream = [&]() {
        if (bgzf && vgmi_fastq_bgzf_status(fq, &inflate_failed, &comp_good, nullptr) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
        closed = true;
        if (vgmi_fastq_close(fq, &n_rec, &n_bases, &consumed, &stopped, tail.data(), tail.size(), &tail_len) != VGMI_OK)
            throw std::runtime_error(vgmi_last_error(ctx));
    };
    try {
        std::unique_ptr<ByteSource> src;
        if (!plain && !bgzf && !dev_gunzip) src = ByteSource::open(path, threads);
        uint64_t offset = 0;
        const unsigned char* left_p = nullptr;   // rest of a decoded chunk that did not fit the previous buffer
        size_t left_n = 0;
        std::vector<char> carry;                 // block gzip: the bytes behind the last whole member of the previous buffer
        double ratio = 5.0;                      // text bytes per compressed byte, tracked
        size_t text_cap = 0;
        if ((bgzf || dev_gunzip) && vgmi_fastq_text_capacity(fq, &text_cap) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
        for (;;) {
            char* buf = nullptr;
            size_t cap = 0;
            if (vgmi_fastq_acquire(fq, &buf, &cap) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
            size_t n = 0;
            if (plain) {
                n = fill_plain(fd, offset, file_size, buf, cap, threads);
                offset += n;
                if (vgmi_fastq_commit(fq, n) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
                if (n < cap) break;   // end of the data
            } else if (bgzf) {
                // compressed bytes whose text fills about nine tenths of a chunk
                size_t want = (size_t)std::min<double>((double)cap, 0.9 * (double)text_cap / ratio), round_want = 0;
                if (vgmi_fastq_bgzf_want(fq, &round_want) == VGMI_OK && round_want) want = std::min(cap, round_want);   // whole rounds of wavefronts
                want = std::max<size_t>(want, std::min<size_t>(cap, carry.size() + (256u << 10)));
                memcpy(buf, carry.data(), carry.size());
                n = carry.size();
                const size_t got = fill_plain(fd, offset, file_size, buf + n, want - n, threads);
                offset += got;
                n += got;
                size_t taken = 0, n_text = 0;
                int not_bgzf = 0;
                if (vgmi_fastq_commit_bgzf(fq, n, &taken, &n_text, &not_bgzf) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
                comp_taken += taken;
                carry.assign(buf + taken, buf + n);
                if (taken && n_text) ratio = std::max(1.0, (double)n_text / (double)taken);
                if (not_bgzf) break;                                  // the host decoder goes on at comp_taken
                if (offset >= file_size && (taken == 0 || carry.empty())) break;   // end of the file (a cut-off member stays for the host)
                if (taken == 0 && want >= cap) break;                 // cannot happen with 64 KiB members; never spin
            } else if (dev_gunzip) {
                // compressed bytes whose text fills about nine tenths of a chunk; what the device leaves untaken (the stretch that
                // runs on into the bytes to come) is presented again in front of them
                size_t want = (size_t)std::min<double>((double)cap, 0.9 * (double)text_cap / ratio);
                want = std::max<size_t>(want, std::min<size_t>(cap, carry.size() + (1u << 20)));
                memcpy(buf, carry.data(), carry.size());
                n = carry.size();
                const size_t got = fill_plain(fd, offset, file_size, buf + n, want - n, threads);
                offset += got;
                n += got;
                const bool at_eof = offset >= file_size;
                size_t taken = 0, n_text = 0;
                int stop = 0;
                if (vgmi_fastq_commit_gzip(fq, n, at_eof ? 1 : 0, &taken, &n_text, &stop) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
                carry.assign(buf + taken, buf + n);
                if (taken && n_text) ratio = std::max(1.0, (double)n_text / (double)taken);
                if (stop == 2) {
                    // reason 11: a member's text as the device resolved it fails the trailer's CRC-32 / ISIZE.  zlib delivers such a
                    // member's text all the same and reports the damage at its end, which include/kseq.h:112 reads as the end of the
                    // data -- what has happened here too, PROVIDED the fault is the file's: the host decoder says whether it is
                    uint32_t why = 0;
                    uint64_t dev_text = 0;
                    if (vgmi_fastq_gzip_status(fq, &dev_text, &why) == VGMI_OK && why == 11) {
                        uint64_t host_text = 0;
                        const GunzipEnd end = host_gunzip_verdict(path, host_text);
                        if (end != GunzipEnd::Corrupt || host_text != dev_text)
                            throw std::runtime_error("'" + path + "': the device's gzip decoder produced text that fails the member's CRC-32 / length while the "
                                                     "host decoder disagrees (internal error; VGH_DEVICE_GUNZIP=0 decodes on the host)");
                        // (Where this differs from the reference on a DAMAGED file: zlib's gzread hands kseq nothing of its last call once the trailer
                        // check fails -- up to 16 KiB of the member's text, include/kseq.h's buffer -- while the member's whole text is counted here.
                        // Files that pass their own checks are byte-identical; ADVICE r5 #3, DESIGN_INGEST_HMM.md 4.8.)
                        std::fprintf(stderr, "[varigraph-mi] warning: '%s': gzip stream is damaged (CRC-32 or length): the damaged member's text is used up to its "
                                             "end (the reference drops the last <= 16 KiB of it)\n", path.c_str());
                        break;
                    }
                }
                if (stop == 2 || (taken == 0 && (at_eof || want >= cap))) {
                    gz_gave_up = true;
                    uint32_t why = 0;
                    uint64_t dev_text = 0;
                    // (the hand-over is said, not silent -- reason 12 is a file of many small members, e.g. pigz -i or per-lane files joined:
                    // a piece per member is launch-bound on the device, and the stream stays with the host decoder from here on)
                    if (vgmi_fastq_gzip_status(fq, &dev_text, &why) == VGMI_OK && why == 12)
                        std::fprintf(stderr, "[varigraph-mi] note: '%s': gzip members of under 1 MiB of text each: decoded by the host's threads from byte %llu of the text on\n",
                                     path.c_str(), (unsigned long long)dev_text);
                    break;
                }
                if (stop == 1 || (at_eof && carry.empty())) break;
            } else {
                for (;;) {
                    if (!left_n && !src->next_chunk(left_p, left_n)) break;
                    const size_t take = std::min(left_n, cap - n);
                    memcpy(buf + n, left_p, take);
                    n += take;
                    left_p += take;
                    left_n -= take;
                    if (n == cap) break;
                }
                if (vgmi_fastq_commit(fq, n) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx));
                if (n < cap) break;   // end of the data
            }
        }
        close_stream();
    } catch (...) {
        if (!closed) (void)vgmi_fastq_close(fq, nullptr, nullptr, nullptr, nullptr, tail.data(), tail.size(), &tail_len);
        throw;
    }
    res.n_reads = n_rec;
    res.read_base = n_bases;
    // what the device did not take: the text after the last complete record (an unterminated last line, or nothing), or
    // -- once it met a record that is not a regular four-line one -- the rest of the stream from that record on; for
    // block gzip also the file from the first member the device did not take or could not vouch for
    if (stopped || gz_gave_up) {       // (the host decodes the file from its start and passes over the text the device has counted)
        FastxReader rd(ByteSource::skip(ByteSource::open(path, threads), consumed));
        host_leg(ctx, rd, path, block_bytes, submit_mu, res.n_reads, res.read_base);
    } else if (bgzf && (inflate_failed ? comp_good : comp_taken) < file_size) {
        const uint64_t at = inflate_failed ? comp_good : comp_taken;
        FastxReader rd(ByteSource::concat(ByteSource::from_memory(tail.data(), tail_len), ByteSource::open_at(path, at, threads)));
        host_leg(ctx, rd, path, block_bytes, submit_mu, res.n_reads, res.read_base);
    } else if (tail_len) {
        FastxReader rd(ByteSource::from_memory(tail.data(), tail_len));
        host_leg(ctx, rd, path, block_bytes, submit_mu, res.n_reads, res.read_base);
    }
    return res;
}

}  // namespace

FastqKmerHip::FastqKmerHip(vgmi_ctx* ctx, const std::vector<std::string>& fastqFileNameVec, uint32_t kmerLen,
                           uint32_t threads, size_t block_bytes)
    : ctx_(ctx), files_(fastqFileNameVec), k_(kmerLen), threads_(threads ? threads : 1), block_bytes_(block_bytes)
{
}

void FastqKmerHip::build_fastq_index()
{
    if (files_.empty()) throw std::runtime_error("Parameter error: -f");  // src/fastq_kmer.cpp:42-45
    uint32_t k_tab = 0;
    if (vgmi_table_info(ctx_, nullptr, &k_tab, nullptr, nullptr) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx_));
    if (k_tab != k_) throw std::runtime_error("k-mer length differs from the uploaded graph table");
    if (vgmi_counts_reset(ctx_) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx_));
    mReadBase = 0;
    mReadNum = 0;

    const char* force_host = std::getenv("VGH_HOST_PARSE");
    if ((k_ & 1) && !(force_host && force_host[0] == '1')) {
        // device-side record parsing: one worker per file (two at a time: the files of a pair), each moving file text
        // through its own pinned buffers and HIP stream
        const size_t n_par = std::min<size_t>(std::min<size_t>(threads_, 2), files_.size());
        const unsigned io_threads = std::max<unsigned>(1u, (unsigned)(threads_ / n_par));
        std::mutex submit_mu, res_mu;
        std::atomic<size_t> next{0};
        std::string err;
        auto worker = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= files_.size()) return;
                try {
                    const DeviceFileResult r = device_file(ctx_, files_[i], block_bytes_, io_threads, submit_mu);
                    std::lock_guard<std::mutex> lk(res_mu);
                    mReadBase += r.read_base;
                    mReadNum += r.n_reads;
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lk(res_mu);
                    if (err.empty()) err = e.what();
                    next = files_.size();
                    return;
                }
            }
        };
        std::vector<std::thread> ws;
        for (size_t t = 1; t < n_par; ++t) ws.emplace_back(worker);
        worker();
        for (auto& t : ws) t.join();
        if (!err.empty()) throw std::runtime_error(err);
        return;
    }

    Channel ch;
    const size_t n_par = std::min<size_t>(threads_, files_.size());
    // block-gzip inputs: the threads not parsing inflate blocks (byte_source.hpp); at least 2 per file
    const unsigned decode_threads = std::max<unsigned>(2u, (unsigned)(threads_ / n_par));
    for (size_t i = 0; i < 2 * n_par + 1; ++i) {
        auto b = std::make_unique<Block>();
        b->bytes.reserve(block_bytes_ + 1024);
        ch.free_list.push_back(std::move(b));
    }
    size_t next_file = 0;
    std::vector<std::thread> workers;
    auto start_more = [&]() {  // called with ch.mu held
        while (ch.producers < n_par && next_file < files_.size()) {
            ch.producers++;
            workers.emplace_back(parse_file, files_[next_file++], std::ref(ch), block_bytes_, decode_threads);
        }
    };
    std::string err;
    {
        std::unique_lock<std::mutex> lk(ch.mu);
        start_more();
        for (;;) {
            ch.cv_full.wait(lk, [&] { return !ch.full.empty() || ch.producers == 0; });
            if (!ch.error.empty() && err.empty()) err = ch.error;
            if (ch.full.empty()) {
                if (next_file < files_.size() && err.empty()) { start_more(); continue; }
                if (ch.producers == 0) break;
                continue;
            }
            auto b = std::move(ch.full.front());
            ch.full.pop_front();
            lk.unlock();
            if (err.empty()) {
                if (vgmi_reads_submit(ctx_, b->bytes.data(), b->bytes.size(), nullptr, b->n_reads) != VGMI_OK)
                    err = vgmi_last_error(ctx_);
                mReadBase += b->read_base;
                mReadNum += b->n_reads;
            }
            lk.lock();
            ch.free_list.push_back(std::move(b));
            ch.cv_free.notify_one();
            start_more();
        }
    }
    for (auto& t : workers) t.join();
    if (!err.empty()) throw std::runtime_error(err);
}

void FastqKmerHip::fetch(uint8_t* cov, uint8_t* cov_node, uint64_t* hist256)
{
    if (vgmi_counts_finish(ctx_, cov, cov_node, hist256) != VGMI_OK) throw std::runtime_error(vgmi_last_error(ctx_));
    float ms = 0;
    uint64_t n = 0;
    vgmi_count_kernel_ms(ctx_, &ms, &n);
    kernel_s_ = ms / 1e3;
}

// src/varigraph.cpp:308-348 (get_hom_kmer_c) over the histogram of get_hom_kmer (:253-296), then
// :230-232 (--use-depth) and :360-362 (cal_hap_kmer_cov)
bool coverage_stats(const uint64_t hist[256], uint64_t read_base, uint64_t genome_size, uint32_t sample_ploidy,
                    bool use_depth, CoverageStats& out)
{
    out.read_depth = read_base / (float)genome_size;  // :198
    // the reference iterates a std::map holding only the coverages that occur
    uint8_t cov[256];
    uint64_t fre[256];
    size_t nb = 0;
    int index = -1, max_index = -1;
    uint8_t max_cov = 0, hom_cov = 0;
    uint64_t max_fre = 0;
    for (int v = 0; v < 256; ++v) {
        if (!hist[v]) continue;
        cov[nb] = (uint8_t)v;
        fre[nb] = hist[v];
        ++nb;
        ++index;
        if (v > 1 && hist[v] >= max_fre && v < UINT8_MAX) {
            max_index = index;
            max_cov = (uint8_t)v;
            max_fre = hist[v];
            hom_cov = (uint8_t)v;
        }
    }
    if (max_index == -1) return false;
    for (size_t i = (size_t)max_index + 1; i < nb - 1; i++) {
        if (cov[i] > out.read_depth) break;
        if (fre[i] >= fre[i - 1] && fre[i] >= fre[i + 1]) hom_cov = cov[i];
    }
    if (use_depth) hom_cov = out.read_depth * 0.8;
    out.max_coverage = max_cov;
    out.hom_coverage = hom_cov;
    out.hap_kmer_coverage = (hom_cov > 0 && sample_ploidy > 0)
                                ? static_cast<float>(hom_cov) / static_cast<float>(sample_ploidy)
                                : out.read_depth / static_cast<float>(sample_ploidy);
    return true;
}

}  // namespace vgh
