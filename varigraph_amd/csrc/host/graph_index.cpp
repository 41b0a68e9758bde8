#include "graph_index.hpp"

#include "mem_advice.hpp"
#include "stl_order_map.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <iterator>
#include <memory>
#include <zlib.h>

#include <chrono>
#include <exception>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

#include "vgmi.h"

namespace vgh {

namespace {

// the bytes of the file: plain files (what `construct --save-graph` writes) are mapped, gzip'd ones inflated through
// zlib (transparent like the gzopen the reference uses for its other inputs)
struct Bytes {
    const uint8_t* data = nullptr;
    size_t size = 0;
    void* map = nullptr;
    std::vector<uint8_t> own;
    Bytes() = default;
    Bytes(const Bytes&) = delete;
    Bytes& operator=(const Bytes&) = delete;
    ~Bytes() { if (map) munmap(map, size); }
};

void slurp(const std::string& path, Bytes& out)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("'" + path + "': No such file or directory.");
    unsigned char magic[2] = {0, 0};
    const ssize_t got = ::pread(fd, magic, 2, 0);
    struct stat st;
    if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m != MAP_FAILED) {
            ::close(fd);
            (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            out.map = m;
            out.data = static_cast<const uint8_t*>(m);
            out.size = (size_t)st.st_size;
            return;
        }
    }
    ::close(fd);
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("'" + path + "': No such file or directory.");
    gzbuffer(fp, 1 << 20);
    std::vector<uint8_t>& buf = out.own;
    buf.resize(1 << 22);
    size_t n = 0;
    for (;;) {
        if (n == buf.size()) buf.resize(buf.size() * 2);
        size_t want = buf.size() - n;
        if (want > (1u << 30)) want = 1u << 30;
        int r = gzread(fp, buf.data() + n, (unsigned)want);
        if (r < 0) {
            gzclose(fp);
            throw std::runtime_error("'" + path + "': read error");
        }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(fp);
    buf.resize(n);
    out.data = buf.data();
    out.size = n;
}

template <typename F>
void parallel_chunks(size_t n, unsigned threads, F&& body)   // body(begin, end, thread)
{
    if (threads <= 1 || n < 4096) {
        body((size_t)0, n, 0u);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (n + threads - 1) / threads;
    for (unsigned t = 0; t < threads; ++t) {
        const size_t b = std::min(n, (size_t)t * per), e = std::min(n, b + per);
        if (b == e) break;
        pool.emplace_back([&body, b, e, t] { body(b, e, t); });
    }
    for (auto& th : pool) th.join();
}

// key -> record index of the k-mer table: open addressing, multiplicative hash (the reference's unordered_map is
// needed only for membership + position here; this is ~5x faster to build and probe at 1e7..1e8 keys)
struct KeyIndex {
    // zero pages straight from the kernel are the empty table.  A cell is tag << 32 | ~index (0 = empty): 32 hash bits
    // select the candidates, the record's key decides; 8 bytes per cell and a load factor of 1/3..2/3 keep the table --
    // whose pages are touched in random order -- small
    const uint64_t* keys = nullptr;
    uint64_t* cell = nullptr;
    size_t bytes = 0;
    uint64_t mask = 0;
    KeyIndex(const KeyIndex&) = delete;
    KeyIndex& operator=(const KeyIndex&) = delete;
    ~KeyIndex() { if (cell) munmap(cell, bytes); }
    KeyIndex(const std::vector<uint64_t>& key_vec, unsigned threads) : keys(key_vec.data())
    {
        uint64_t cap = 16;
        while (2 * cap < 3 * key_vec.size()) cap <<= 1;
        mask = cap - 1;
        bytes = cap * sizeof(uint64_t);
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::runtime_error("graph index: out of memory");
#ifdef MADV_HUGEPAGE
        (void)madvise(m, bytes, MADV_HUGEPAGE);   // touched at random: 2 MiB pages spare most of the TLB misses and page faults
#endif
        cell = static_cast<uint64_t*>(m);
        uint64_t* cp = cell;
        // cells are claimed with a compare-and-swap, so the threads fill the table side by side; the FIRST record of a
        // key wins like unordered_map::emplace (smallest index = largest ~index, whatever the arrival order)
        parallel_chunks(key_vec.size(), threads, [&](size_t b, size_t e, unsigned) {
            for (size_t i = b; i < e; ++i) {
                if (i + 12 < e) __builtin_prefetch(&cp[(hash(keys[i + 12]) >> 8) & mask], 1);
                const uint64_t k = keys[i];
                const uint64_t h = hash(k);
                const uint64_t mine = (h & 0xFFFFFFFF00000000ULL) | (uint32_t) ~(uint32_t)i;
                uint64_t s = (h >> 8) & mask;
                for (;;) {
                    uint64_t cur = __atomic_load_n(&cp[s], __ATOMIC_RELAXED);
                    if (cur == 0 && __atomic_compare_exchange_n(&cp[s], &cur, mine, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                    if ((cur >> 32) == (mine >> 32) && keys[(uint32_t) ~(uint32_t)cur] == k) {   // the same key again
                        while (mine > cur && !__atomic_compare_exchange_n(&cp[s], &cur, mine, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                        break;
                    }
                    s = (s + 1) & mask;
                }
            }
        });
    }
    static uint64_t hash(uint64_t k)
    {
        k *= 0x9E3779B97F4A7C15ULL;
        return (k ^ (k >> 29)) | (1ULL << 63);   // tag never 0: an occupied cell is never 0
    }
    void prefetch(uint64_t k) const { __builtin_prefetch(&cell[(hash(k) >> 8) & mask]); }
    // second stage, for a key whose cell was asked for a while ago: the record its first candidate points at
    void prefetch_record(uint64_t k) const
    {
        const uint64_t h = hash(k), cur = cell[(h >> 8) & mask];
        if (cur != 0 && (cur >> 32) == (h >> 32)) __builtin_prefetch(&keys[(uint32_t) ~(uint32_t)cur]);
    }
    bool find(uint64_t k, uint32_t& out) const
    {
        const uint64_t h = hash(k);
        for (uint64_t s = (h >> 8) & mask;; s = (s + 1) & mask) {
            const uint64_t cur = cell[s];
            if (cur == 0) return false;
            if ((cur >> 32) == (h >> 32)) {
                const uint32_t i = ~(uint32_t)cur;
                if (keys[i] == k) {
                    out = i;
                    return true;
                }
            }
        }
    }
};

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    template <typename T>
    T get()
    {
        if ((size_t)(end - p) < sizeof(T)) throw std::runtime_error("graph index truncated");
        T v;
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str()
    {
        uint32_t n = get<uint32_t>();
        if ((size_t)(end - p) < n) throw std::runtime_error("graph index truncated");
        std::string s(reinterpret_cast<const char*>(p), n);
        p += n;
        return s;
    }
    // a count read from the file must be backed by that many bytes BEFORE anything is sized by it
    void need(size_t n) const
    {
        if ((size_t)(end - p) < n) throw std::runtime_error("graph index corrupt: a list is longer than the file");
    }
    void bytes(void* dst, size_t n)
    {
        if ((size_t)(end - p) < n) throw std::runtime_error("graph index truncated");
        if (n) memcpy(dst, p, n);   // an empty list's data() may be null
        p += n;
    }
};

}  // namespace

void GraphIndex::load(const std::string& path)
{
    const bool timing = getenv("VGH_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) {
        if (timing) std::fprintf(stderr, "[graph_index] %-22s %.3f s\n", what, now() - t0);
        t0 = now();
    };
    Bytes buf;
    slurp(path, buf);
    lap("file read");
    Cursor c{buf.data, buf.data + buf.size};

    graph_base_num = c.get<uint64_t>();
    k = c.get<uint32_t>();
    vcf_ploidy = c.get<uint32_t>();
    if (k < 1 || k > 28) throw std::runtime_error("graph index: bad k-mer length");
    vcf_head = c.str();

    // The three sections -- VCF lines, nodes, k-mer records -- are independent; a first pass over the length fields finds
    // where the second and the third begin, then the sections are read side by side (the first two on a thread of their
    // own each, the fixed-size k-mer records by the rest).
    auto skip_str = [](Cursor& w) {
        const uint32_t n = w.get<uint32_t>();
        w.need(n);
        w.p += n;
    };
    Cursor c_vcf = c;
    {
        const uint32_t n_info = c.get<uint32_t>();
        for (uint32_t i = 0; i < n_info; ++i) {
            skip_str(c);
            (void)c.get<uint32_t>();
            const uint32_t n_sites = c.get<uint32_t>();
            for (uint32_t j = 0; j < n_sites; ++j) {
                (void)c.get<uint32_t>();
                const uint32_t n_fields = c.get<uint32_t>();
                for (uint32_t q = 0; q < n_fields; ++q) skip_str(c);
            }
        }
    }
    Cursor c_nodes = c;
    {
        const uint16_t n_hap = c.get<uint16_t>();
        hap_num = n_hap;          // the k-mer records are checked against it while the names are still being read
        for (uint16_t i = 0; i < n_hap; ++i) {
            (void)c.get<uint16_t>();
            skip_str(c);
        }
        const uint32_t n_chr = c.get<uint32_t>();
        for (uint32_t i = 0; i < n_chr; ++i) {
            skip_str(c);
            const uint32_t n_nodes = c.get<uint32_t>();
            for (uint32_t j = 0; j < n_nodes; ++j) {
                (void)c.get<uint32_t>();
                const uint32_t n_seq = c.get<uint32_t>();
                c.need((size_t)n_seq * 4);
                for (uint32_t q = 0; q < n_seq; ++q) skip_str(c);
                const uint32_t n_gt = c.get<uint32_t>();
                c.need(sizeof(uint16_t) * (size_t)n_gt);
                c.p += sizeof(uint16_t) * (size_t)n_gt;
                const uint32_t n_km = c.get<uint32_t>();
                c.need(sizeof(uint64_t) * (size_t)n_km);
                c.p += sizeof(uint64_t) * (size_t)n_km;
            }
        }
    }
    lap("section offsets");
    std::exception_ptr err_vcf, err_nodes;
    auto read_vcf = [&]() {
        try {
            Cursor c = c_vcf;
            const uint32_t n_info = c.get<uint32_t>();
            for (uint32_t i = 0; i < n_info; ++i) {
                std::string chr = c.str();
                const uint32_t len = c.get<uint32_t>();
                chr_len[chr] = len;
                genome_size += len;  // construct_index.cpp:959-960
                auto& sites = vcf_info[chr];
                const uint32_t n_sites = c.get<uint32_t>();
                for (uint32_t j = 0; j < n_sites; ++j) {
                    const uint32_t start = c.get<uint32_t>();
                    const uint32_t n_fields = c.get<uint32_t>();
                    // CHROM .. INFO: what the VCF writer and --sv read (src/genotype.cpp:1579-1696, :600-610); the FORMAT column and the
                    // cohort's genotype strings behind it are hopped over, not turned into strings (half of a site's sixteen)
                    std::vector<std::string> fields;
                    const uint32_t keep = std::min<uint32_t>(n_fields, kVcfFieldsKept);
                    fields.reserve(keep);
                    for (uint32_t q = 0; q < keep; ++q) fields.push_back(c.str());
                    for (uint32_t q = keep; q < n_fields; ++q) skip_str(c);
                    if (sites.empty() || std::prev(sites.end())->first < start) sites.emplace_hint(sites.end(), start, std::move(fields));
                    else sites[start] = std::move(fields);
                }
            }

        } catch (...) {
            err_vcf = std::current_exception();
        }
    };
    bool seq_in_order = true;
    graph_seq.clear();
    auto read_nodes = [&]() {
        try {
            Cursor c = c_nodes;
            const uint16_t n_names = c.get<uint16_t>();
            for (uint16_t i = 0; i < n_names; ++i) {
                const uint16_t idx = c.get<uint16_t>();
                hap_names.emplace(idx, c.str());
            }

            const uint32_t n_chr = c.get<uint32_t>();
            for (uint32_t i = 0; i < n_chr; ++i) {
                std::string chr = c.str();
                auto& nodes = graph[chr];
                auto& seq = graph_seq[chr];
                const uint32_t n_nodes = c.get<uint32_t>();
                seq.reserve(seq.size() + n_nodes);
                for (uint32_t j = 0; j < n_nodes; ++j) {
                    GraphNode nd;
                    nd.start = c.get<uint32_t>();
                    const uint32_t n_seq = c.get<uint32_t>();
                    c.need((size_t)n_seq * 4);   // every sequence carries at least its length field
                    nd.seqs.reserve(n_seq);
                    for (uint32_t q = 0; q < n_seq; ++q) nd.seqs.push_back(c.str());
                    const uint32_t n_gt = c.get<uint32_t>();
                    c.need(sizeof(uint16_t) * (size_t)n_gt);
                    nd.hap_gt.resize(n_gt);
                    c.bytes(nd.hap_gt.data(), sizeof(uint16_t) * n_gt);
                    const uint32_t n_km = c.get<uint32_t>();
                    c.need(sizeof(uint64_t) * (size_t)n_km);
                    nd.kmer_file = n_km ? c.p : nullptr;        // read in place by graph2node (below, before the buffer goes)
                    nd.kmer_file_n = n_km;
                    c.p += sizeof(uint64_t) * (size_t)n_km;
                    const uint32_t st = nd.start;
                    // the writer walks its map: starts ascend, and a node goes behind the last one without a search of the tree
                    if (nodes.empty() || std::prev(nodes.end())->first < st) {
                        seq.push_back(&nodes.emplace_hint(nodes.end(), st, std::move(nd))->second);
                    } else {
                        nodes[st] = std::move(nd);
                        seq_in_order = false;
                    }
                }
            }

        } catch (...) {
            err_nodes = std::current_exception();
        }
    };
    std::thread th_vcf, th_nodes;
    if (threads >= 3) {
        th_vcf = std::thread(read_vcf);
        th_nodes = std::thread(read_nodes);
    } else {
        read_vcf();
        read_nodes();
    }
    struct Joiner {
        std::thread &a, &b;
        ~Joiner() { if (a.joinable()) a.join(); if (b.joinable()) b.join(); }
    } joiner{th_vcf, th_nodes};
    (void)c.get<uint64_t>();  // ReadBase (always 0 in a graph index)

    // k-mer records until EOF: u64 key | u8 c | u8 f | u64 bitLen | i8[bitLen].  bitLen = floor(#haplotypes / 8) + 1 in every
    // record (construct_index.cpp:1206-1215; the bitmap readers -- hom flags, HMM -- index it by haplotype number), so the
    // records have one size and the threads take a stretch of them each; every record's bitLen is still checked.
    keys.clear(); f.clear(); bitvec.clear();
    bitlen = 0;
    if (c.p < c.end) {
        const size_t left = (size_t)(c.end - c.p);
        if (left < 18) throw std::runtime_error("graph index truncated");
        uint64_t bl0;
        memcpy(&bl0, c.p + 10, 8);
        if (bl0 != (uint64_t)(hap_num >> 3) + 1) throw std::runtime_error("graph index corrupt: k-mer bitmap length does not match the haplotype count");
        bitlen = bl0;
        const size_t rec = 18 + (size_t)bl0;
        if (left % rec != 0) {
            // a short last record, or records of another size somewhere: the serial walk names the reason
            Cursor w = c;
            while (w.p < w.end) {
                (void)w.get<uint64_t>();
                (void)w.get<uint8_t>();
                (void)w.get<uint8_t>();
                const uint64_t bl = w.get<uint64_t>();
                if (bl != bitlen) throw std::runtime_error("graph index: k-mer records with different bitmap lengths");
                w.need((size_t)bl);
                w.p += bl;
            }
        }
        const size_t n_rec = left / rec;
        keys.reserve(n_rec);
        f.reserve(n_rec);
        bitvec.reserve(n_rec * bl0);
        advise_huge_pages(keys.data(), n_rec * sizeof(uint64_t));   // before the first touch
        advise_huge_pages(f.data(), n_rec);
        advise_huge_pages(bitvec.data(), n_rec * bl0);
        keys.resize(n_rec);
        f.resize(n_rec);
        bitvec.resize(n_rec * bl0);
        const uint8_t* const base = c.p;
        std::atomic<bool> mixed{false};
        parallel_chunks(n_rec, threads >= 3 ? threads - 2 : threads, [&](size_t rb, size_t re, unsigned) {
            for (size_t i = rb; i < re; ++i) {
                const uint8_t* r = base + i * rec;
                uint64_t bl;
                memcpy(&keys[i], r, 8);              // key; r[8] = c: per-sample, zero in the index
                f[i] = r[9];
                memcpy(&bl, r + 10, 8);
                if (bl != bl0) mixed = true;
                memcpy(bitvec.data() + i * bl0, r + 18, bl0);
            }
        });
        if (mixed) throw std::runtime_error("graph index: k-mer records with different bitmap lengths");
        c.p = c.end;
    }
    lap("k-mer records");
    if (on_keys) on_keys();
    if (th_vcf.joinable()) th_vcf.join();
    if (th_nodes.joinable()) th_nodes.join();
    if (err_vcf) std::rethrow_exception(err_vcf);
    if (err_nodes) std::rethrow_exception(err_nodes);
    if (!seq_in_order) index_nodes();
    lap("VCF lines, nodes");
    graph2node();        // (forgets the nodes' places in the file buffer, which ends with this function)
    lap("graph2node");
    compute_hom_flags();
    lap("hom flags");
}

// src/construct_index.cpp:710-751 + :1572-1603: per variant node, resolve kmerHashVec against the
// table (drop absent keys), and when more than 128 remain, std::sort by frequency ascending and
// keep the first 128.  The sort runs on a vector with the same length, initial order and
// comparison outcomes as the reference's vector of map iterators, so libstdc++'s introsort
// produces the same permutation.
void GraphIndex::build_entry_words()
{
    entry_words.clear();
    if (bitlen > 6) return;
    const size_t n = node_key_index.size(), bl = bitlen;
    entry_words.reserve(n);
    advise_huge_pages(entry_words.data(), n * sizeof(uint64_t));
    entry_words.resize(n);
    parallel_chunks(n, threads, [&](size_t b, size_t e, unsigned) {
        for (size_t j = b; j < e; ++j) {
            const size_t key = node_key_index[j];
            uint64_t bits = 0;
            std::memcpy(&bits, &bitvec[key * bl], bl);
            entry_words[j] = (uint64_t)(uint8_t)f[key] << 8 | bits << 16;
        }
    });
}

void GraphIndex::index_nodes()
{
    graph_seq.clear();
    for (const auto& [chr, nodes] : graph) {
        auto& seq = graph_seq[chr];
        seq.reserve(nodes.size());
        for (const auto& [start, nd] : nodes) seq.push_back(&nd);
    }
}

void GraphIndex::graph2node()
{
    const bool timing = getenv("VGH_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) {
        if (timing) std::fprintf(stderr, "[graph_index]   %-20s %.3f s\n", what, now() - t0);
        t0 = now();
    };
    chr_names.clear(); node_chr.clear(); node_start.clear(); node_key_index.clear();
    node_off.assign(1, 0);
    // variant nodes in mGraphMap order, resolved side by side, appended in order
    std::vector<const GraphNode*> vnodes;
    {
        size_t have = 0, want = 0;
        for (const auto& kv : graph_seq) have += kv.second.size();
        for (const auto& kv : graph) want += kv.second.size();
        if (have != want || graph_seq.size() != graph.size()) index_nodes();     // a graph that was not read by load()
    }
    for (const auto& [chr, seq] : graph_seq) {
        const uint32_t chr_id = (uint32_t)chr_names.size();
        chr_names.push_back(chr);
        for (const GraphNode* nd : seq) {
            if (nd->hap_gt.size() == 1) continue;
            node_chr.push_back(chr_id);
            node_start.push_back(nd->start);
            vnodes.push_back(nd);
        }
    }
    std::vector<std::vector<uint32_t>> kept_all(vnodes.size());
    const uint8_t* fp = f.data();
    // a graph that load() is reading keeps its nodes' hashes in the file buffer: whichever way the lists are resolved, the
    // nodes forget those places before this function returns (the buffer does not outlive load())
    // (EVERY node: the ones with a single allele are not in vnodes, and their place would dangle just the same)
    auto forget_file_places = [&]() {
        for (auto& kv : graph_seq) {
            auto& seq = kv.second;
            parallel_chunks(seq.size(), threads, [&](size_t b, size_t e, unsigned) {
                for (size_t v = b; v < e; ++v) {
                    GraphNode* nd = const_cast<GraphNode*>(seq[v]);
                    nd->kmer_file = nullptr;
                    nd->kmer_file_n = 0;
                }
            });
        }
    };
    if (batched_find) {
        // every node's k-mers in one array, one batched lookup (the device holds the table already), the answers written per node
        std::vector<uint64_t> q_off(vnodes.size() + 1, 0);
        for (size_t v = 0; v < vnodes.size(); ++v) q_off[v + 1] = q_off[v] + vnodes[v]->n_kmers();
        const size_t total_q = q_off.back();
        std::unique_ptr<uint64_t[]> flat(new uint64_t[total_q ? total_q : 1]);        // (no zero fill: every word is written below)
        std::unique_ptr<uint32_t[]> found(new uint32_t[total_q ? total_q : 1]);
        parallel_chunks(vnodes.size(), threads, [&](size_t b, size_t e, unsigned) {
            for (size_t v = b; v < e; ++v)
                if (vnodes[v]->n_kmers())
                    std::memcpy(&flat[q_off[v]], vnodes[v]->kmer_file ? (const void*)vnodes[v]->kmer_file : (const void*)vnodes[v]->kmer_hash.data(),
                                vnodes[v]->n_kmers() * 8);
        });
        lap("node k-mers listed");
        if (batched_find(flat.get(), total_q, found.get())) {
            lap("batched lookup");
            node_off.assign(vnodes.size() + 1, 0);
            parallel_chunks(vnodes.size(), threads, [&](size_t b, size_t e, unsigned) {
                for (size_t v = b; v < e; ++v) {
                    size_t n_kept = 0;
                    for (size_t j = q_off[v]; j < q_off[v + 1]; ++j) n_kept += found[j] != 0xFFFFFFFFu;
                    node_off[v + 1] = std::min<size_t>(n_kept, 128);
                }
            });
            for (size_t v = 0; v < vnodes.size(); ++v) node_off[v + 1] += node_off[v];
            node_key_index.resize(node_off.back());
            parallel_chunks(vnodes.size(), threads, [&](size_t b, size_t e, unsigned) {
                std::vector<uint32_t> kept;
                for (size_t v = b; v < e; ++v) {
                    uint32_t* dst = node_key_index.data() + node_off[v];
                    size_t n_kept = 0;
                    for (size_t j = q_off[v]; j < q_off[v + 1]; ++j) n_kept += found[j] != 0xFFFFFFFFu;
                    if (n_kept <= 128) {
                        for (size_t j = q_off[v]; j < q_off[v + 1]; ++j)
                            if (found[j] != 0xFFFFFFFFu) *dst++ = found[j];
                        continue;
                    }
                    kept.clear();
                    for (size_t j = q_off[v]; j < q_off[v + 1]; ++j)
                        if (found[j] != 0xFFFFFFFFu) kept.push_back(found[j]);
                    std::sort(kept.begin(), kept.end(), [fp](uint32_t a, uint32_t b2) { return fp[a] < fp[b2]; });
                    std::memcpy(dst, kept.data(), 128 * 4);
                }
            });
            lap("node lists written");
            forget_file_places();
            return;
        }
    }
    {
    const KeyIndex index(keys, threads);
    lap("key index");
    parallel_chunks(vnodes.size(), threads, [&](size_t b, size_t e, unsigned) {
        for (size_t v = b; v < e; ++v) {
            const GraphNode& nd = *vnodes[v];
            std::vector<uint32_t>& kept = kept_all[v];
            kept.reserve(nd.n_kmers());
            const size_t nk = nd.n_kmers();
            for (size_t j = 0; j < nk && j < 16; ++j) index.prefetch(nd.kmer(j));
            for (size_t j = 0; j < nk; ++j) {
                if (j + 16 < nk) index.prefetch(nd.kmer(j + 16));
                if (j + 8 < nk) index.prefetch_record(nd.kmer(j + 8));
                uint32_t at;
                if (index.find(nd.kmer(j), at)) kept.push_back(at);
            }
            if (kept.size() > 128) {
                std::sort(kept.begin(), kept.end(), [fp](uint32_t a, uint32_t b2) { return fp[a] < fp[b2]; });
                kept.resize(128);
            }
        }
    });
    }
    lap("node lookups");
    node_off.resize(vnodes.size() + 1);
    for (size_t v = 0; v < vnodes.size(); ++v) node_off[v + 1] = node_off[v] + kept_all[v].size();
    node_key_index.resize(node_off.back());
    parallel_chunks(vnodes.size(), threads, [&](size_t b, size_t e, unsigned) {
        for (size_t v = b; v < e; ++v)
            if (!kept_all[v].empty()) std::memcpy(&node_key_index[node_off[v]], kept_all[v].data(), kept_all[v].size() * 4);
    });
    lap("node lists joined");
    forget_file_places();
}

// src/varigraph.cpp:263-287 without the per-sample `c == 0` test: f <= 1 and, for some VCF sample,
// all vcf_ploidy consecutive haplotypes (1-based hap index, bit i%8 of byte i/8) carry the k-mer.
void GraphIndex::compute_hom_flags()
{
    hom_flag.assign(keys.size(), 0);
    parallel_chunks(keys.size(), threads, [&](size_t rb, size_t re, unsigned) {
        for (size_t r = rb; r < re; ++r) {
            if (f[r] > 1) continue;
            const int8_t* bv = bitvec.data() + r * bitlen;
            uint32_t index = 0, sample_count = 0;
            for (uint32_t i = 1; i < hap_num; ++i) {
                index++;
                if ((bv[i >> 3] >> (i & 7)) & 1) sample_count++;
                if (index == vcf_ploidy) {
                    index = 0;
                    if (sample_count == vcf_ploidy) { hom_flag[r] = 1; break; }
                    sample_count = 0;
                }
            }
        }
    });
}

void GraphIndex::save_reads_index(const std::string& path, const uint8_t* cov, uint64_t read_base) const
{
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("'" + path + "': No such file or directory.");
    const size_t rec = 18 + (size_t)bitlen, n = keys.size();
    // The reference writes its unordered_map in iteration order, and that map was filled by load_index with operator[] in
    // graph.bin's record order (src/construct_index.cpp:1067-1100): the order libstdc++ gives that insert sequence
    StlOrderMap seq(0);
    for (size_t i = 0; i < n; ++i) seq.emplace(keys[i]);
    const std::vector<uint32_t> order = seq.order();
    std::vector<uint8_t> buf;
    buf.reserve((size_t)1 << 22);
    bool ok = std::fwrite(&read_base, 8, 1, fp) == 1;
    const uint64_t bl = bitlen;
    for (size_t at = 0; at < n && ok; ++at) {
        const size_t i = order[at];
        const size_t o = buf.size();
        buf.resize(o + rec);
        uint8_t* r = buf.data() + o;
        memcpy(r, &keys[i], 8);
        r[8] = cov[i];
        r[9] = f[i];
        memcpy(r + 10, &bl, 8);
        memcpy(r + 18, bitvec.data() + i * bitlen, bitlen);
        if (buf.size() + rec > buf.capacity() || at + 1 == n) {
            ok = std::fwrite(buf.data(), 1, buf.size(), fp) == buf.size();
            buf.clear();
        }
    }
    if (std::fclose(fp) != 0) ok = false;
    if (!ok) throw std::runtime_error("'" + path + "': write error.");
}

void GraphIndex::load_reads_index(const std::string& path, uint8_t* cov, uint64_t& read_base) const
{
    Bytes in;
    slurp(path, in);
    Cursor c{in.data, in.data + in.size};
    read_base = c.get<uint64_t>();
    std::memset(cov, 0, keys.size());
    std::unique_ptr<KeyIndex> index;      // only for a file whose records are not in this graph's order
    size_t i = 0;
    while (c.p < c.end) {
        const uint64_t key = c.get<uint64_t>();
        const uint8_t cv = c.get<uint8_t>();
        (void)c.get<uint8_t>();           // f: the graph's own
        const uint64_t bl = c.get<uint64_t>();
        c.need(bl);
        c.p += bl;
        if (i < keys.size() && keys[i] == key) {
            cov[i++] = cv;
            continue;
        }
        if (!index) index.reset(new KeyIndex(keys, threads));
        uint32_t at;
        if (!index->find(key, at)) throw std::runtime_error("'" + path + "': a k-mer that is not in the graph");
        cov[at] = cv;
        ++i;
    }
}

int GraphIndex::upload(vgmi_ctx* ctx) const
{
    int rc = vgmi_table_upload(ctx, keys.data(), keys.size(), k);
    if (rc) return rc;
    return upload_nodes(ctx);
}

// the per-node lists and the hom flags alone, for a context that received its table image from another device
int GraphIndex::upload_nodes(vgmi_ctx* ctx) const
{
    int rc = vgmi_nodes_upload(ctx, node_off.data(), node_key_index.data(), node_off.size() - 1);
    if (rc) return rc;
    return vgmi_flags_upload(ctx, hom_flag.data());
}

}  // namespace vgh
