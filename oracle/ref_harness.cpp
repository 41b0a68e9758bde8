// ref_harness.cpp -- drives the UNMODIFIED reference (/root/reference) so its outputs can be
// dumped as golden vectors and timed as the CPU baseline.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ref_harness by
// compiling the reference sources where they lie; no reference source is copied here.
// This file only *calls* the reference's public API:
//   kmerBit::kmer_sketch_fastq / kmer_sketch_genotype / kmer_sketch_bf   (include/kmer.hpp)
//   BloomFilter (subclassed to reach the protected seeds/filter)        (include/counting_bloom_filter.hpp)
//   ConstructIndex::load_index / graph2node                              (include/construct_index.hpp)
//   FastqKmer::build_fastq_index                                         (include/fastq_kmer.hpp)
//   Varigraph::get_hom_kmer / get_hom_kmer_c / cal_hap_kmer_cov          (include/varigraph.hpp)
//
// Sub-commands (all output is little-endian binary or plain text on stdout):
//   hash64 K                      stdin: one hex canonical k-mer value per line -> hash64 hex
//   sketch K                      stdin: one sequence per line -> "n key0 key1 ..." (hex, ordered, dups)
//   bloomsize N P                 -> "m nh"
//   murmur                        stdin: "key_hex seed_hex" -> sum hex
//   bloom K N P seed0,seed1,.. OUT  stdin: sequences -> BloomFilter::save() dump in OUT; then for
//                                 each stdin line after a line "Q": key_hex -> "count find"
//   mbf FASTA K OUT               ConstructIndex::build_fasta_index + make_mbf, then BloomFilter::save(OUT)
//                                 (seeds are random unless this is the det_shim build, ref_harness_det)
//   count GRAPH THREADS OUT FQ...  load_index + build_fastq_index; writes OUT (see below); prints timing
//   sample GRAPH THREADS PLOIDY USEDEPTH OUT FQ...   count + graph2node + hom-kmer statistics
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "include/varigraph.hpp"

using namespace std;

struct OpenBF : public BloomFilter {
    OpenBF(uint64_t n, double p) : BloomFilter(n, p) {}
    void set_seeds(const vector<uint64_t>& s) { _seeds = s; }
    const vector<uint64_t>& seeds() const { return _seeds; }
    uint64_t murmur(uint64_t key, uint64_t seed) {
        return _murmur_hash(static_cast<void*>(&key), sizeof(key), seed);
    }
    static uint64_t size_for(uint64_t n, double p) { return _calculate_size(n, p); }
    static uint32_t hashes_for(uint64_t n, uint64_t m) { return _calculate_num_hashes(n, m); }
};

struct OpenVG : public Varigraph {
    OpenVG(const VarigraphConfig& c) : Varigraph(c) {}
    ConstructIndex* ci() { return ConstructIndexClassPtr_; }
    void set_depth(float d) { ReadDepth_ = d; }
    float depth() const { return ReadDepth_; }
    float hapcov() const { return hapKmerCoverage_; }
    bool use_depth() const { return useDepth_; }
};

template <typename T> static void put(ostream& o, const T& v) { o.write(reinterpret_cast<const char*>(&v), sizeof(T)); }

static int cmd_hash64(int argc, char** argv) {
    uint32_t k = atoi(argv[2]);
    uint64_t mask = (1ULL << 2 * k) - 1;
    string line;
    while (getline(cin, line)) {
        if (line.empty()) continue;
        uint64_t v = strtoull(line.c_str(), nullptr, 16);
        printf("%llx\n", (unsigned long long)hash64(v, mask));
    }
    return 0;
}

static int cmd_sketch(int argc, char** argv) {
    uint32_t k = atoi(argv[2]);
    string line;
    while (getline(cin, line)) {
        // the table holds every key the sequence emits, so kmer_sketch_fastq returns the full
        // ordered trace (src/kmer.cpp:140-142)
        unordered_map<uint64_t, kmerCovFreBitVec> table;
        for (auto h : kmerBit::kmer_sketch_genotype(line, k)) table[h];
        vector<uint64_t> keys = kmerBit::kmer_sketch_fastq(line, k, table);
        printf("%zu", keys.size());
        for (auto h : keys) printf(" %llx", (unsigned long long)h);
        printf("\n");
    }
    return 0;
}

static int cmd_bloomsize(int argc, char** argv) {
    uint64_t n = strtoull(argv[2], nullptr, 10);
    double p = atof(argv[3]);
    uint64_t m = OpenBF::size_for(n, p);
    printf("%llu %u\n", (unsigned long long)m, OpenBF::hashes_for(n, m));
    return 0;
}

static int cmd_murmur(int argc, char** argv) {
    OpenBF bf(100, 0.01);
    string line;
    while (getline(cin, line)) {
        if (line.empty()) continue;
        unsigned long long key, seed;
        if (sscanf(line.c_str(), "%llx %llx", &key, &seed) != 2) continue;
        printf("%llx\n", (unsigned long long)bf.murmur(key, seed));
    }
    return 0;
}

static int cmd_bloom(int argc, char** argv) {
    uint32_t k = atoi(argv[2]);
    uint64_t n = strtoull(argv[3], nullptr, 10);
    double p = atof(argv[4]);
    vector<uint64_t> seeds;
    {
        stringstream ss(argv[5]);
        string tok;
        while (getline(ss, tok, ',')) seeds.push_back(strtoull(tok.c_str(), nullptr, 16));
    }
    string out = argv[6];
    OpenBF bf(n, p);
    if (seeds.size() != bf.get_num()) {
        fprintf(stderr, "need %u seeds\n", bf.get_num());
        return 2;
    }
    bf.set_seeds(seeds);
    string line;
    bool query = false;
    while (getline(cin, line)) {
        if (line == "Q") { query = true; bf.save(out); continue; }
        if (line.empty()) continue;
        if (!query) {
            kmerBit::kmer_sketch_bf(line, k, &bf);
        } else {
            uint64_t key = strtoull(line.c_str(), nullptr, 16);
            printf("%u %d\n", (unsigned)bf.count(key), bf.find(key) ? 1 : 0);
        }
    }
    if (!query) bf.save(out);
    return 0;
}

static int cmd_mbf(int argc, char** argv) {
    string fasta = argv[2], none, out = argv[4];
    uint32_t k = atoi(argv[3]);
    ConstructIndex ci(fasta, none, none, none, false, false, k, 2, false, 1);
    ci.build_fasta_index();
    ci.make_mbf();
    ci.mbf->save(out);
    printf("genome_size %llu\nm %llu\nn_hash %u\n", (unsigned long long)ci.mGenomeSize,
           (unsigned long long)ci.mbf->get_size(), ci.mbf->get_num());
    return 0;
}

// OUT layout for count/sample:
//   u64 readBase | u64 genomeSize | u64 n | n * { u64 key | u8 c | u8 f }   (unordered_map iteration order)
static void dump_counts(const string& out, uint64_t readBase, ConstructIndex* ci) {
    ofstream o(out, ios::binary);
    put<uint64_t>(o, readBase);
    put<uint64_t>(o, ci->mGenomeSize);
    put<uint64_t>(o, (uint64_t)ci->mGraphKmerHashHapStrMap.size());
    for (const auto& kv : ci->mGraphKmerHashHapStrMap) {
        put<uint64_t>(o, kv.first);
        put<uint8_t>(o, kv.second.c);
        put<uint8_t>(o, kv.second.f);
    }
}

static int cmd_count(int argc, char** argv, bool sample) {
    VarigraphConfig cfg;
    cfg.inputGraphFileName = argv[2];
    cfg.threads = atoi(argv[3]);
    int a = 4;
    if (sample) {
        cfg.samplePloidy = atoi(argv[a++]);
        cfg.useDepth = atoi(argv[a++]) != 0;
    }
    string out = argv[a++];
    vector<string> fqs;
    for (; a < argc; ++a) fqs.push_back(argv[a]);

    OpenVG vg(cfg);
    auto t0 = chrono::steady_clock::now();
    vg.load();
    auto t1 = chrono::steady_clock::now();
    ConstructIndex* ci = vg.ci();
    if (sample) ci->graph2node();
    auto t2 = chrono::steady_clock::now();
    FastqKmer fk(ci->mGraphKmerHashHapStrMap, fqs, ci->mKmerLen, cfg.threads);
    fk.build_fastq_index();
    auto t3 = chrono::steady_clock::now();
    dump_counts(out, fk.mReadBase, ci);
    if (const char* idx = getenv("VG_SAVE_READS_INDEX")) fk.save_index(idx);   // FastqKmer's own dump (src/fastq_kmer.cpp:200-238)

    auto sec = [](auto a, auto b) { return chrono::duration<double>(b - a).count(); };
    printf("load_s %.6f\ngraph2node_s %.6f\nbuild_fastq_index_s %.6f\nread_base %llu\nn_keys %zu\nk %u\nthreads %u\n",
           sec(t0, t1), sec(t1, t2), sec(t2, t3), (unsigned long long)fk.mReadBase,
           ci->mGraphKmerHashHapStrMap.size(), ci->mKmerLen, cfg.threads);

    if (sample) {
        // src/varigraph.cpp:198, 220-243
        vg.set_depth(fk.mReadBase / (float)ci->mGenomeSize);
        map<uint8_t, uint64_t> hist = vg.get_hom_kmer();
        uint8_t maxc, homc;
        tie(maxc, homc) = vg.get_hom_kmer_c(hist);
        if (vg.use_depth()) homc = vg.depth() * 0.8;
        vg.cal_hap_kmer_cov(homc);
        float d = vg.depth(), h = vg.hapcov();
        uint32_t db, hb;
        memcpy(&db, &d, 4);
        memcpy(&hb, &h, 4);
        printf("read_depth_bits %08x\nmax_cov %u\nhom_cov %u\nhap_kmer_cov_bits %08x\n", db, (unsigned)maxc, (unsigned)homc, hb);
        printf("hist");
        for (int v = 0; v < 256; ++v) {
            auto it = hist.find((uint8_t)v);
            printf(" %llu", (unsigned long long)(it == hist.end() ? 0 : it->second));
        }
        printf("\n");
        // node -> k-mer order after graph2node (src/construct_index.cpp:710-751,1572-1603)
        ofstream o(out + ".nodes", ios::binary);
        for (auto& [chr, m] : ci->mGraphMap) {
            for (auto& [start, node] : m) {
                if (node.hapGtVec.size() == 1) continue;
                put<uint32_t>(o, (uint32_t)chr.size());
                o.write(chr.data(), chr.size());
                put<uint32_t>(o, start);
                put<uint32_t>(o, (uint32_t)node.GraphKmerHashHapStrMapIterVec.size());
                for (auto& it : node.GraphKmerHashHapStrMapIterVec) put<uint64_t>(o, it->first);
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_harness <cmd> ...\n"); return 2; }
    string c = argv[1];
    if (c == "hash64" && argc >= 3) return cmd_hash64(argc, argv);
    if (c == "sketch" && argc >= 3) return cmd_sketch(argc, argv);
    if (c == "bloomsize" && argc >= 4) return cmd_bloomsize(argc, argv);
    if (c == "murmur") return cmd_murmur(argc, argv);
    if (c == "bloom" && argc >= 7) return cmd_bloom(argc, argv);
    if (c == "mbf" && argc >= 5) return cmd_mbf(argc, argv);
    if (c == "count" && argc >= 6) return cmd_count(argc, argv, false);
    if (c == "sample" && argc >= 8) return cmd_count(argc, argv, true);
    fprintf(stderr, "bad command line\n");
    return 2;
}
