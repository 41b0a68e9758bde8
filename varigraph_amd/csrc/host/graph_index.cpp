#include "graph_index.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

#include "vgmi.h"

namespace vgh {

namespace {

// whole file into memory through zlib (transparent for plain files, like the gzopen the
// reference uses for its other inputs); graph.bin itself is written uncompressed
std::vector<uint8_t> slurp(const std::string& path)
{
    // plain file (what `construct --save-graph` writes): one bulk read
    if (FILE* f = fopen(path.c_str(), "rb")) {
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, f);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
            std::vector<uint8_t> buf;
            if (fseeko(f, 0, SEEK_END) == 0) {
                const off_t size = ftello(f);
                if (size >= 0 && fseeko(f, 0, SEEK_SET) == 0) {
                    buf.resize((size_t)size);
                    size_t n = 0;
                    while (n < buf.size()) {
                        const size_t r = fread(buf.data() + n, 1, buf.size() - n, f);
                        if (r == 0) break;
                        n += r;
                    }
                    fclose(f);
                    if (n != buf.size()) throw std::runtime_error("'" + path + "': read error");
                    return buf;
                }
            }
        }
        fclose(f);
    }
    gzFile fp = gzopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("'" + path + "': No such file or directory.");
    gzbuffer(fp, 1 << 20);
    std::vector<uint8_t> buf;
    size_t cap = 1 << 22;
    buf.resize(cap);
    size_t n = 0;
    for (;;) {
        if (n == buf.size()) buf.resize(buf.size() * 2);
        size_t want = buf.size() - n;
        if (want > (1u << 30)) want = 1u << 30;
        int got = gzread(fp, buf.data() + n, (unsigned)want);
        if (got < 0) {
            gzclose(fp);
            throw std::runtime_error("'" + path + "': read error");
        }
        if (got == 0) break;
        n += (size_t)got;
    }
    gzclose(fp);
    buf.resize(n);
    return buf;
}

// key -> record index of the k-mer table: open addressing, multiplicative hash (the reference's unordered_map is
// needed only for membership + position here; this is ~5x faster to build and probe at 1e7..1e8 keys)
struct KeyIndex {
    std::vector<uint64_t> key;
    std::vector<uint32_t> idx;
    uint64_t mask = 0;
    explicit KeyIndex(const std::vector<uint64_t>& keys)
    {
        uint64_t cap = 16;
        while (cap < 2 * keys.size()) cap <<= 1;
        mask = cap - 1;
        key.assign(cap, ~0ULL);
        idx.assign(cap, 0);
        for (size_t i = 0; i < keys.size(); ++i) {
            uint64_t s = slot(keys[i]);
            while (key[s] != ~0ULL && key[s] != keys[i]) s = (s + 1) & mask;
            if (key[s] == ~0ULL) {   // the first record of a key wins, like unordered_map::emplace
                key[s] = keys[i];
                idx[s] = (uint32_t)i;
            }
        }
    }
    uint64_t slot(uint64_t k) const { return ((k * 0x9E3779B97F4A7C15ULL) >> 20) & mask; }
    bool find(uint64_t k, uint32_t& out) const
    {
        for (uint64_t s = slot(k);; s = (s + 1) & mask) {
            if (key[s] == k) {
                out = idx[s];
                return true;
            }
            if (key[s] == ~0ULL) return false;
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
    void bytes(void* dst, size_t n)
    {
        if ((size_t)(end - p) < n) throw std::runtime_error("graph index truncated");
        memcpy(dst, p, n);
        p += n;
    }
};

}  // namespace

void GraphIndex::load(const std::string& path)
{
    std::vector<uint8_t> buf = slurp(path);
    Cursor c{buf.data(), buf.data() + buf.size()};

    graph_base_num = c.get<uint64_t>();
    k = c.get<uint32_t>();
    vcf_ploidy = c.get<uint32_t>();
    if (k < 1 || k > 28) throw std::runtime_error("graph index: bad k-mer length");
    vcf_head = c.str();

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
            std::vector<std::string> fields;
            fields.reserve(n_fields);
            for (uint32_t q = 0; q < n_fields; ++q) fields.push_back(c.str());
            sites[start] = std::move(fields);
        }
    }

    hap_num = c.get<uint16_t>();
    for (uint16_t i = 0; i < hap_num; ++i) {
        const uint16_t idx = c.get<uint16_t>();
        hap_names.emplace(idx, c.str());
    }

    const uint32_t n_chr = c.get<uint32_t>();
    for (uint32_t i = 0; i < n_chr; ++i) {
        std::string chr = c.str();
        auto& nodes = graph[chr];
        const uint32_t n_nodes = c.get<uint32_t>();
        for (uint32_t j = 0; j < n_nodes; ++j) {
            GraphNode nd;
            nd.start = c.get<uint32_t>();
            const uint32_t n_seq = c.get<uint32_t>();
            nd.seqs.reserve(n_seq);
            for (uint32_t q = 0; q < n_seq; ++q) nd.seqs.push_back(c.str());
            const uint32_t n_gt = c.get<uint32_t>();
            nd.hap_gt.resize(n_gt);
            c.bytes(nd.hap_gt.data(), sizeof(uint16_t) * n_gt);
            const uint32_t n_km = c.get<uint32_t>();
            nd.kmer_hash.resize(n_km);
            c.bytes(nd.kmer_hash.data(), sizeof(uint64_t) * n_km);
            const uint32_t st = nd.start;
            nodes[st] = std::move(nd);
        }
    }

    (void)c.get<uint64_t>();  // ReadBase (always 0 in a graph index)

    // k-mer records until EOF: u64 key | u8 c | u8 f | u64 bitLen | i8[bitLen]
    keys.clear(); f.clear(); bitvec.clear();
    bitlen = 0;
    if ((size_t)(c.end - c.p) >= 26) {   // all records have the same size: reserve once
        uint64_t bl0;
        memcpy(&bl0, c.p + 10, 8);
        const size_t n_rec = (size_t)(c.end - c.p) / (18 + bl0) + 1;
        keys.reserve(n_rec);
        f.reserve(n_rec);
        bitvec.reserve(n_rec * bl0);
    }
    while (c.p < c.end) {
        const uint64_t key = c.get<uint64_t>();
        (void)c.get<uint8_t>();  // c: per-sample, zero in the index
        const uint8_t fv = c.get<uint8_t>();
        const uint64_t bl = c.get<uint64_t>();
        if (keys.empty()) bitlen = bl;
        if (bl != bitlen) throw std::runtime_error("graph index: k-mer records with different bitmap lengths");
        keys.push_back(key);
        f.push_back(fv);
        const size_t o = bitvec.size();
        bitvec.resize(o + bl);
        c.bytes(bitvec.data() + o, bl);
    }
    graph2node();
    compute_hom_flags();
}

// src/construct_index.cpp:710-751 + :1572-1603: per variant node, resolve kmerHashVec against the
// table (drop absent keys), and when more than 128 remain, std::sort by frequency ascending and
// keep the first 128.  The sort runs on a vector with the same length, initial order and
// comparison outcomes as the reference's vector of map iterators, so libstdc++'s introsort
// produces the same permutation.
void GraphIndex::graph2node()
{
    const KeyIndex index(keys);

    chr_names.clear(); node_chr.clear(); node_start.clear(); node_key_index.clear();
    node_off.assign(1, 0);
    for (const auto& [chr, nodes] : graph) {
        const uint32_t chr_id = (uint32_t)chr_names.size();
        chr_names.push_back(chr);
        for (const auto& [start, nd] : nodes) {
            if (nd.hap_gt.size() == 1) continue;
            std::vector<uint32_t> kept;
            kept.reserve(nd.kmer_hash.size());
            for (uint64_t h : nd.kmer_hash) {
                uint32_t at;
                if (index.find(h, at)) kept.push_back(at);
            }
            if (kept.size() > 128) {
                const uint8_t* fp = f.data();
                std::sort(kept.begin(), kept.end(), [fp](uint32_t a, uint32_t b) { return fp[a] < fp[b]; });
                kept.resize(128);
            }
            node_chr.push_back(chr_id);
            node_start.push_back(start);
            node_key_index.insert(node_key_index.end(), kept.begin(), kept.end());
            node_off.push_back(node_key_index.size());
        }
    }
}

// src/varigraph.cpp:263-287 without the per-sample `c == 0` test: f <= 1 and, for some VCF sample,
// all vcf_ploidy consecutive haplotypes (1-based hap index, bit i%8 of byte i/8) carry the k-mer.
void GraphIndex::compute_hom_flags()
{
    hom_flag.assign(keys.size(), 0);
    for (size_t r = 0; r < keys.size(); ++r) {
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
}

int GraphIndex::upload(vgmi_ctx* ctx) const
{
    int rc = vgmi_table_upload(ctx, keys.data(), keys.size(), k);
    if (rc) return rc;
    rc = vgmi_nodes_upload(ctx, node_off.data(), node_key_index.data(), node_off.size() - 1);
    if (rc) return rc;
    return vgmi_flags_upload(ctx, hom_flag.data());
}

}  // namespace vgh
