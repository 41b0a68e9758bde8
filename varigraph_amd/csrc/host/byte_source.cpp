#include "byte_source.hpp"

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "fast_inflate.hpp"
#include "par_gunzip.hpp"

namespace vgh {

namespace {

constexpr size_t kChunk = 1u << 20;      // decoded bytes handed over at a time
constexpr size_t kPlain = 256u << 10;    // plain files: read(2) size (stays cache resident for the parser)
constexpr size_t kInBuf = 1u << 20;      // compressed bytes read at a time
constexpr size_t kTaskOut = 2u << 20;    // BGZF: decoded bytes per worker task (about 32 blocks)

// file descriptor with a push-back area (the bytes looked at to tell the formats apart)
class FileIn {
public:
    explicit FileIn(const std::string& path, uint64_t offset = 0) : path_(path), fd_(::open(path.c_str(), O_RDONLY))
    {
        if (fd_ < 0) throw std::runtime_error("'" + path + "': No such file or directory.");
        if (offset && ::lseek(fd_, (off_t)offset, SEEK_SET) < 0) throw std::runtime_error("'" + path + "': cannot seek.");
#ifdef POSIX_FADV_SEQUENTIAL
        (void)posix_fadvise(fd_, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    }
    ~FileIn() { if (fd_ >= 0) ::close(fd_); }
    FileIn(const FileIn&) = delete;
    FileIn& operator=(const FileIn&) = delete;

    size_t read(unsigned char* dst, size_t n)   // short only at the end of the file
    {
        size_t got = 0;
        if (back_pos_ < back_.size()) {
            got = std::min(n, back_.size() - back_pos_);
            std::memcpy(dst, back_.data() + back_pos_, got);
            back_pos_ += got;
        }
        while (got < n) {
            const ssize_t r = ::read(fd_, dst + got, n - got);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) {   // an I/O error ends the data like gzread's -1 ends the reference's loop -- but not silently
                std::fprintf(stderr, "[varigraph-mi] warning: '%s': read error (%s); the data ends here\n", path_.c_str(),
                             std::strerror(errno));
                break;
            }
            if (r == 0) break;
            got += (size_t)r;
        }
        return got;
    }
    void unread(const unsigned char* p, size_t n)   // in front of whatever is still pushed back
    {
        std::vector<unsigned char> nb(p, p + n);
        nb.insert(nb.end(), back_.begin() + (long)back_pos_, back_.end());
        back_.swap(nb);
        back_pos_ = 0;
    }

    const std::string& path() const { return path_; }

private:
    std::string path_;
    int fd_;
    std::vector<unsigned char> back_;
    size_t back_pos_ = 0;
};

constexpr size_t kHistory = 32768;       // room in front of a chunk's payload for the decoder's window (fast_inflate.hpp)
constexpr size_t kSlack = 512;

struct Chunk {
    std::vector<unsigned char> data;     // kHistory + payload capacity + kSlack
    size_t n = 0;                        // payload bytes
    unsigned char* payload() { return data.data() + kHistory; }
};

// filled chunks in order from one producer to the consumer, empty ones back
class Pipe {
public:
    explicit Pipe(size_t n_chunks, size_t bytes)
    {
        for (size_t i = 0; i < n_chunks; ++i) {
            auto c = std::make_unique<Chunk>();
            c->data.resize(kHistory + bytes + kSlack);
            free_.push_back(std::move(c));
        }
    }
    std::unique_ptr<Chunk> get_free()   // null once cancelled
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_free_.wait(lk, [&] { return cancel_ || !free_.empty(); });
        if (cancel_) return nullptr;
        auto c = std::move(free_.back());
        free_.pop_back();
        c->n = 0;
        return c;
    }
    void put_full(std::unique_ptr<Chunk> c)
    {
        std::lock_guard<std::mutex> lk(mu_);
        full_.push_back(std::move(c));
        cv_full_.notify_one();
    }
    void finish()
    {
        std::lock_guard<std::mutex> lk(mu_);
        done_ = true;
        cv_full_.notify_all();
    }
    std::unique_ptr<Chunk> take()   // null at the end
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_full_.wait(lk, [&] { return done_ || !full_.empty(); });
        if (full_.empty()) return nullptr;
        auto c = std::move(full_.front());
        full_.pop_front();
        return c;
    }
    void recycle(std::unique_ptr<Chunk> c)
    {
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(std::move(c));
        cv_free_.notify_one();
    }
    void cancel()
    {
        std::lock_guard<std::mutex> lk(mu_);
        cancel_ = true;
        cv_free_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_full_, cv_free_;
    std::deque<std::unique_ptr<Chunk>> full_;
    std::vector<std::unique_ptr<Chunk>> free_;
    bool done_ = false, cancel_ = false;
};

// zlib inflate over (concatenated) gzip members from the current position of `in` into `pipe`; returns when the
// data ends (cleanly or not) or the pipe is cancelled.  Kept as the cross-check of the decoder below (VGH_ZLIB_INFLATE=1).
void inflate_members_zlib(FileIn& in, Pipe& pipe)
{
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return;
    std::vector<unsigned char> inbuf(kInBuf);
    std::unique_ptr<Chunk> cur;
    for (;;) {
        if (zs.avail_in == 0) {
            const size_t n = in.read(inbuf.data(), inbuf.size());
            if (n == 0) break;   // end of file: clean after a member, truncated inside one -- the data ends either way
            zs.next_in = inbuf.data();
            zs.avail_in = (uInt)n;
        }
        if (!cur) {
            cur = pipe.get_free();
            if (!cur) break;
            zs.next_out = cur->payload();
            zs.avail_out = (uInt)kChunk;
        }
        const int ret = inflate(&zs, Z_NO_FLUSH);
        cur->n = kChunk - zs.avail_out;
        if (ret == Z_STREAM_END) {
            // another member?  (gz_look: fewer than two bytes left, or no gzip magic = trailing garbage, ignored)
            if (zs.avail_in < 2) {
                unsigned char tmp[2];
                const size_t have = zs.avail_in;
                if (have) tmp[0] = *zs.next_in;
                const size_t n = in.read(inbuf.data() + have, inbuf.size() - have);
                if (have) inbuf[0] = tmp[0];
                zs.next_in = inbuf.data();
                zs.avail_in = (uInt)(have + n);
            }
            if (zs.avail_in < 2 || zs.next_in[0] != 0x1f || zs.next_in[1] != 0x8b) break;
            if (inflateReset(&zs) != Z_OK) break;
        } else if (ret != Z_OK && ret != Z_BUF_ERROR) {
            break;   // corrupt stream: what decoded so far is delivered, then the data ends
        }
        if (zs.avail_out == 0) {
            pipe.put_full(std::move(cur));
            cur.reset();
        }
    }
    if (cur) {
        if (cur->n) pipe.put_full(std::move(cur));
        else pipe.recycle(std::move(cur));
    }
    inflateEnd(&zs);
}

// the same through this repo's decoder (fast_inflate.hpp): about twice zlib's rate
void inflate_members(FileIn& in, Pipe& pipe)
{
    static const bool use_zlib = [] {
        const char* e = getenv("VGH_ZLIB_INFLATE");
        return e && e[0] == '1';
    }();
    if (use_zlib) return inflate_members_zlib(in, pipe);
    std::unique_ptr<Chunk> cur;
    GunzipIO io;
    io.read = [&](unsigned char* dst, size_t n) { return in.read(dst, n); };
    io.next_buffer = [&](size_t, size_t) -> unsigned char* {
        cur = pipe.get_free();
        return cur ? cur->payload() : nullptr;
    };
    io.commit = [&](size_t n) {
        cur->n = n;
        pipe.put_full(std::move(cur));
    };
    const GunzipEnd end = fast_gunzip(io, kChunk);
    // the reference's gzread loop ends the same way at a damaged or short stream; the decoder here knows why
    if (end == GunzipEnd::Corrupt || end == GunzipEnd::Truncated)
        std::fprintf(stderr, "[varigraph-mi] warning: '%s': gzip stream %s; only what decoded cleanly is used\n", in.path().c_str(),
                     end == GunzipEnd::Corrupt ? "is damaged (bad block, CRC-32 or length)" : "ends inside a member");
    if (cur) pipe.recycle(std::move(cur));
}

class PlainSource final : public ByteSource {
public:
    explicit PlainSource(std::unique_ptr<FileIn> in) : in_(std::move(in)), buf_(kPlain) {}
    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        n = in_->read(buf_.data(), buf_.size());
        p = buf_.data();
        return n != 0;
    }
    const char* kind() const override { return "plain"; }

private:
    std::unique_ptr<FileIn> in_;
    std::vector<unsigned char> buf_;
};

class GzipSource final : public ByteSource {
public:
    explicit GzipSource(std::unique_ptr<FileIn> in) : in_(std::move(in)), pipe_(4, kChunk)
    {
        th_ = std::thread([this] {
            inflate_members(*in_, pipe_);
            pipe_.finish();
        });
    }
    ~GzipSource() override
    {
        pipe_.cancel();
        th_.join();
    }
    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        if (held_) pipe_.recycle(std::move(held_));
        held_ = pipe_.take();
        if (!held_) return false;
        p = held_->payload();
        n = held_->n;
        return true;
    }
    const char* kind() const override { return "gzip"; }

private:
    std::unique_ptr<FileIn> in_;
    Pipe pipe_;
    std::thread th_;
    std::unique_ptr<Chunk> held_;
};

// ---- BGZF ----
struct BgzfHeader {
    bool is_bgzf = false;
    uint32_t block_size = 0;   // whole member, header to ISIZE
    uint32_t header_len = 0;   // 12 + XLEN
};

// `h` holds the first 12 bytes of a member and its extra field (12 + XLEN bytes)
BgzfHeader parse_bgzf_extra(const unsigned char* h, size_t n)
{
    BgzfHeader r;
    if (n < 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return r;   // FLG must be FEXTRA only
    const uint32_t xlen = h[10] | (h[11] << 8);
    if (n < 12 + (size_t)xlen) return r;
    size_t o = 12;
    while (o + 4 <= 12 + (size_t)xlen) {
        const uint32_t slen = h[o + 2] | (h[o + 3] << 8);
        if (h[o] == 'B' && h[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + (size_t)xlen) {
            r.is_bgzf = true;
            r.block_size = (uint32_t)(h[o + 4] | (h[o + 5] << 8)) + 1u;
            r.header_len = 12 + xlen;
            return r;
        }
        o += 4 + slen;
    }
    return r;
}

class BgzfSource final : public ByteSource {
    struct Block { uint32_t in_off, in_len, out_off, out_len, crc; };
    struct Task {
        std::vector<unsigned char> in, out;
        std::vector<Block> blocks;
        size_t out_n = 0;
        bool decoded = false;   // guarded by mu_
        bool failed = false;    // a block did not inflate: out_n stops in front of it, the data ends there
        bool last = false;      // produced by the stream fallback's end / end of file
        void reset() { in.clear(); blocks.clear(); out_n = 0; decoded = false; failed = false; last = false; }
    };

public:
    BgzfSource(std::unique_ptr<FileIn> in, unsigned threads) : in_(std::move(in)), n_workers_(threads ? threads : 1)
    {
        max_inflight_ = 2 * n_workers_ + 2;
        for (unsigned i = 0; i < n_workers_; ++i) workers_.emplace_back([this] { work(); });
        reader_ = std::thread([this] { produce(); });
    }
    ~BgzfSource() override
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            cancel_ = true;
        }
        cv_work_.notify_all();
        cv_order_.notify_all();
        cv_room_.notify_all();
        reader_.join();
        for (auto& t : workers_) t.join();
    }
    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            if (held_) {
                spare_.push_back(std::move(held_));
                cv_room_.notify_one();
            }
            if (ended_) return false;
            cv_order_.wait(lk, [&] { return (!order_.empty() && order_.front()->decoded) || (order_.empty() && eof_); });
            if (order_.empty()) { ended_ = true; return false; }
            held_ = std::move(order_.front());
            order_.pop_front();
            cv_room_.notify_one();
            if (held_->failed) {   // nothing after the damage is delivered
                std::fprintf(stderr, "[varigraph-mi] warning: '%s': a block-gzip member does not inflate; the data ends in front of it\n",
                             in_->path().c_str());
                ended_ = true;
                cancel_ = true;
                cv_work_.notify_all();
                cv_room_.notify_all();
            }
            if (held_->out_n == 0) {
                if (ended_) return false;
                continue;
            }
            p = held_->out.data();
            n = held_->out_n;
            return true;
        }
    }
    const char* kind() const override { return "bgzf"; }

private:
    std::shared_ptr<Task> fresh_task()   // null once cancelled; waits while too many tasks are in flight
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_room_.wait(lk, [&] { return cancel_ || order_.size() + (held_ ? 1 : 0) < max_inflight_; });
        if (cancel_) return nullptr;
        std::shared_ptr<Task> t;
        if (!spare_.empty()) {
            t = std::move(spare_.back());
            spare_.pop_back();
        } else {
            t = std::make_shared<Task>();
        }
        t->reset();
        return t;
    }
    void dispatch(std::shared_ptr<Task> t, bool already_decoded)
    {
        std::lock_guard<std::mutex> lk(mu_);
        t->decoded = already_decoded;
        order_.push_back(t);
        if (already_decoded) cv_order_.notify_all();
        else {
            work_.push_back(std::move(t));
            cv_work_.notify_one();
        }
    }
    void produce()
    {
        std::shared_ptr<Task> cur;
        size_t cur_out = 0;
        std::vector<unsigned char> hdr, body_buf;
        for (;;) {
            hdr.resize(12);
            size_t n = in_->read(hdr.data(), 12);
            if (n == 0) break;
            BgzfHeader bh;
            if (n == 12 && hdr[0] == 0x1f && hdr[1] == 0x8b && hdr[2] == 8 && hdr[3] == 4) {
                const uint32_t xlen = hdr[10] | (hdr[11] << 8);
                hdr.resize(12 + xlen);
                n += in_->read(hdr.data() + 12, xlen);
                if (n == 12 + (size_t)xlen) bh = parse_bgzf_extra(hdr.data(), n);
            }
            if (!bh.is_bgzf || bh.block_size < bh.header_len + 8) {
                // not a BGZF member (or not gzip at all): the stream decoder takes over from here, in this thread
                if (n < 2 || hdr[0] != 0x1f || hdr[1] != 0x8b) break;   // trailing garbage after the last member
                in_->unread(hdr.data(), n);
                if (cur) { dispatch(std::move(cur), false); cur.reset(); }
                stream_rest();
                break;
            }
            const uint32_t body = bh.block_size - bh.header_len;   // deflate data + CRC32 + ISIZE
            body_buf.resize(body);
            if (in_->read(body_buf.data(), body) != body) break;   // truncated file: the blocks before it still count
            const unsigned char* tr = body_buf.data() + body - 8;
            const uint32_t crc = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24);
            const uint32_t isize = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
            if (isize > (1u << 16)) {   // not a block bgzip would write: this member goes to the stream decoder
                in_->unread(body_buf.data(), body);
                in_->unread(hdr.data(), hdr.size());
                if (cur) { dispatch(std::move(cur), false); cur.reset(); }
                stream_rest();
                break;
            }
            if (!cur) {
                cur = fresh_task();
                if (!cur) return;
                cur_out = 0;
            }
            const size_t off = cur->in.size();
            cur->in.insert(cur->in.end(), body_buf.begin(), body_buf.end());
            cur->blocks.push_back(Block{(uint32_t)off, body - 8, (uint32_t)cur_out, isize, crc});
            cur_out += isize;
            if (cur_out >= kTaskOut) {
                dispatch(std::move(cur), false);
                cur.reset();
            }
        }
        if (cur) {
            if (!cur->blocks.empty()) dispatch(std::move(cur), false);
        }
        std::lock_guard<std::mutex> lk(mu_);
        eof_ = true;
        cv_order_.notify_all();
    }
    // ordinary gzip members in the middle of the file: decoded sequentially into tasks that are complete on arrival
    void stream_rest()
    {
        Pipe pipe(2, kChunk);
        std::thread dec([&] {
            inflate_members(*in_, pipe);
            pipe.finish();
        });
        for (;;) {
            auto c = pipe.take();
            if (!c) break;
            auto t = fresh_task();
            if (!t) { pipe.cancel(); break; }
            t->out.assign(c->payload(), c->payload() + c->n);
            t->out_n = c->n;
            pipe.recycle(std::move(c));
            dispatch(std::move(t), true);
        }
        pipe.cancel();
        dec.join();
    }
    void work()
    {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return;
        for (;;) {
            std::shared_ptr<Task> t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return cancel_ || !work_.empty(); });
                if (cancel_) break;
                t = std::move(work_.front());
                work_.pop_front();
            }
            size_t total = 0;
            for (const Block& b : t->blocks) total += b.out_len;
            if (t->out.size() < total) t->out.resize(total);
            size_t good = 0;
            bool failed = false;
            for (const Block& b : t->blocks) {
                inflateReset(&zs);
                zs.next_in = t->in.data() + b.in_off;
                zs.avail_in = b.in_len;
                zs.next_out = t->out.data() + b.out_off;
                zs.avail_out = b.out_len;
                const int ret = inflate(&zs, Z_FINISH);
                if (ret != Z_STREAM_END || zs.avail_out != 0 ||
                    crc32_fast(0, t->out.data() + b.out_off, b.out_len) != b.crc) {
                    failed = true;
                    break;
                }
                good = b.out_off + b.out_len;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                t->out_n = good;
                t->failed = failed;
                t->decoded = true;
            }
            cv_order_.notify_all();
        }
        inflateEnd(&zs);
    }

    std::unique_ptr<FileIn> in_;
    unsigned n_workers_;
    size_t max_inflight_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_order_, cv_room_;
    std::deque<std::shared_ptr<Task>> order_, work_;
    std::vector<std::shared_ptr<Task>> spare_;
    std::shared_ptr<Task> held_;
    bool eof_ = false, cancel_ = false, ended_ = false;
    std::vector<std::thread> workers_;
    std::thread reader_;
};

}  // namespace

std::unique_ptr<ByteSource> ByteSource::open(const std::string& path, unsigned decode_threads) { return open_at(path, 0, decode_threads); }

std::unique_ptr<ByteSource> ByteSource::open_at(const std::string& path, uint64_t offset, unsigned decode_threads)
{
    auto in = std::make_unique<FileIn>(path, offset);
    // enough of the first member to see a BGZF extra field (bgzip writes XLEN = 6)
    unsigned char head[64];
    const size_t n = in->read(head, sizeof head);
    in->unread(head, n);
    if (n < 2 || head[0] != 0x1f || head[1] != 0x8b) {
        // Transparent (plain text) mode exists at the START of a file only.  At an offset the caller continues a compressed
        // stream (the device took the block-gzip members in front): bytes there that are no gzip header are trailing
        // garbage, which gzread -- and BgzfSource / GzipSource above a whole file -- ignore.  Delivering them as text would
        // turn garbage holding '@' or '>' into reads.
        if (offset > 0) return from_memory(nullptr, 0);
        return std::make_unique<PlainSource>(std::move(in));
    }
    if (parse_bgzf_extra(head, n).is_bgzf) return std::make_unique<BgzfSource>(std::move(in), decode_threads);
    // an ordinary gzip file read from its start with threads to spare: several of them decode it (par_gunzip.hpp);
    // VGH_PAR_GUNZIP=0 keeps the one-thread decoder
    if (offset == 0 && decode_threads >= 2) {
        const char* e = getenv("VGH_PAR_GUNZIP");
        if (!(e && e[0] == '0'))
            if (auto par = open_parallel_gunzip(path, decode_threads)) return par;
    }
    return std::make_unique<GzipSource>(std::move(in));
}

namespace {
class MemorySource final : public ByteSource {
public:
    MemorySource(const void* d, size_t n) : p_(static_cast<const unsigned char*>(d)), n_(n) {}
    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        if (!n_) return false;
        p = p_;
        n = n_;
        n_ = 0;
        return true;
    }
    const char* kind() const override { return "memory"; }

private:
    const unsigned char* p_;
    size_t n_;
};

class SkipSource final : public ByteSource {
public:
    SkipSource(std::unique_ptr<ByteSource> in, uint64_t n) : in_(std::move(in)), left_(n) {}
    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        for (;;) {
            if (!in_->next_chunk(p, n)) return false;
            if (left_ >= n) { left_ -= n; continue; }
            p += left_;
            n -= (size_t)left_;
            left_ = 0;
            if (n) return true;
        }
    }
    const char* kind() const override { return in_->kind(); }

private:
    std::unique_ptr<ByteSource> in_;
    uint64_t left_;
};
}  // namespace

namespace {
class ConcatSource final : public ByteSource {
public:
    ConcatSource(std::unique_ptr<ByteSource> a, std::unique_ptr<ByteSource> b) : a_(std::move(a)), b_(std::move(b)) {}
    bool next_chunk(const unsigned char*& p, size_t& n) override
    {
        if (a_) {
            if (a_->next_chunk(p, n)) return true;
            a_.reset();
        }
        return b_->next_chunk(p, n);
    }
    const char* kind() const override { return b_->kind(); }

private:
    std::unique_ptr<ByteSource> a_, b_;
};
}  // namespace

std::unique_ptr<ByteSource> ByteSource::concat(std::unique_ptr<ByteSource> a, std::unique_ptr<ByteSource> b)
{
    return std::make_unique<ConcatSource>(std::move(a), std::move(b));
}

std::unique_ptr<ByteSource> ByteSource::from_memory(const void* data, size_t n) { return std::make_unique<MemorySource>(data, n); }
std::unique_ptr<ByteSource> ByteSource::skip(std::unique_ptr<ByteSource> inner, uint64_t n)
{
    return n ? std::make_unique<SkipSource>(std::move(inner), n) : std::move(inner);
}

}  // namespace vgh
