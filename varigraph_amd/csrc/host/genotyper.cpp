// genotyper.cpp -- see genotyper.hpp.  Floating-point note: every score is an x87 `long double`; the order of the
// operations and the std:: overload each one resolves to (double vs long double) are part of the contract, because
// the VCF prints GPP with one decimal and GQ from log10(1 - p).
#include "genotyper.hpp"

#include <immintrin.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <iomanip>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <set>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "../vgmi_device.h"
#include "fixed1.hpp"
#include "mem_advice.hpp"
#include "node_flanks.hpp"

namespace vgh {

namespace {
std::mutex g_cpu_mu;
std::condition_variable g_cpu_cv;
unsigned g_cpu_limit = 0, g_cpu_used = 0;
std::atomic<long long> g_cpu_wait_ns{0};        // VGH_TIMING: how long helpers stood in line for a token
const bool g_cpu_timing = getenv("VGH_TIMING") != nullptr;
}  // namespace

void CpuBudget::set(unsigned tokens)
{
    std::lock_guard<std::mutex> lk(g_cpu_mu);
    g_cpu_limit = tokens;
    g_cpu_cv.notify_all();
}

CpuBudget::Hold::Hold()
{
    std::unique_lock<std::mutex> lk(g_cpu_mu);
    if (g_cpu_timing && g_cpu_limit != 0 && g_cpu_used >= g_cpu_limit) {
        const auto t0 = std::chrono::steady_clock::now();
        g_cpu_cv.wait(lk, [] { return g_cpu_limit == 0 || g_cpu_used < g_cpu_limit; });
        g_cpu_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    } else {
        g_cpu_cv.wait(lk, [] { return g_cpu_limit == 0 || g_cpu_used < g_cpu_limit; });
    }
    ++g_cpu_used;
}

CpuBudget::Hold::~Hold()
{
    {
        std::lock_guard<std::mutex> lk(g_cpu_mu);
        --g_cpu_used;
    }
    g_cpu_cv.notify_one();
}

namespace {

// ---------------------------------------------------------------- small numeric pieces (src/genotype.cpp:930-1150)
void poisson_interval(const double& lambda, double& lower, double& upper)
{
    const double sd = std::sqrt(lambda);
    upper = lambda + 1.96 * sd;
    lower = lambda - 1.96 * sd;
}

// recombination / no-recombination probabilities for a gap of `distance_bp` (Li-Stephens style, :940-950)
std::pair<long double, long double> transition_probabilities(uint32_t distance_bp, uint16_t population)
{
    const double effective_population_size = 1e-05;
    const double recomb_rate = 1.26;
    const long double d = distance_bp * 0.000004L * ((long double)recomb_rate) * effective_population_size;
    const long double recomb = (1.0L - std::exp(-d / (long double)population)) * (1.0L / (long double)population);
    const long double no_recomb = std::exp(-d / (long double)population) + recomb;
    return {recomb, no_recomb};
}

long double poisson_pmf(long double mean, uint8_t value)
{
    long double sum = 0.0L;
    const int v = (int)value;
    for (size_t i = 1; i <= value; ++i) sum += std::log(i);   // double log of an integer, accumulated in long double
    const long double log_val = -mean + v * std::log(mean) - sum;
    return std::exp(log_val);
}

double error_param(double ave_cov)
{
    if (ave_cov < 10.0) return 0.99;
    if (ave_cov < 20) return 0.95;
    if (ave_cov < 40) return 0.9;
    return 0.8;
}

long double geometric_prior(long double p)
{
    const long double mean = 0.5;
    const long double variance = 0.05;
    return (1 / (std::sqrt(2 * M_PI * variance))) * std::exp(-std::pow(p - mean, 2) / (2 * variance));
}

long double geometric_likelihood(long double p, uint8_t value)
{
    const long double q = 1.0 - p;
    return (std::pow(q, value)) * (std::pow(p, (1 - value)));
}

long double geometric(long double p, uint8_t value) { return geometric_likelihood(p, value) * geometric_prior(p); }

// h/c/f adjustment before scoring (:1118-1145)
void most_likely_depth(uint8_t h, uint8_t& c, uint8_t f, float ave_cov, double upper)
{
    if (f == 1) return;
    if (h > 0 && c > (ave_cov * h)) {
        c = ave_cov * h;
    } else if (h == 0 && c > ave_cov) {
        c = (f > ((float)c / upper)) ? 0 : c / (float)f;
    } else if (h == 0 && c <= ave_cov) {
        c /= (float)f;
    }
}

double phred_scaled(long double value) { return (value >= 1.0) ? 99 : (-10 * std::log10(1.0 - value)); }

// every k-mer key of a sequence (kmerBit::kmer_sketch_genotype, src/kmer.cpp:150-190)
std::unordered_set<uint64_t> sequence_keys(const std::string& s, uint32_t k)
{
    std::unordered_set<uint64_t> out;
    out.reserve(s.size());
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t fwd = 0, rev = 0;
    int l = 0, span = 0;
    const unsigned int len = (unsigned int)s.size();
    for (unsigned int i = 0; i < len; ++i) {
        const uint32_t c = vg_nt4((uint8_t)s[i]);
        if (c < 4) {
            span = l + 1 < (int)k ? l + 1 : (int)k;
            fwd = (fwd << 2 | c) & mask;
            rev = (rev >> 2) | (3ULL ^ c) << shift1;
            if (fwd == rev) continue;
            const uint64_t canon = fwd < rev ? fwd : rev;
            ++l;
            if (l >= (int)k && span < 256) out.insert(vg_hash64(canon, mask) << 8 | (uint64_t)span);
        } else {
            l = 0;
            span = 0;
        }
    }
    return out;
}

// seed source of the haplotype sampler: std::random_device like the reference, or a fixed value for reproducible
// runs (VGH_RANDOM_DEVICE_VALUE; the reference's deterministic test build pins random_device the same way)
uint32_t random_device_value()
{
    if (const char* e = std::getenv("VGH_RANDOM_DEVICE_VALUE")) return (uint32_t)std::strtoul(e, nullptr, 10);
    std::random_device rd;
    return rd();
}

// Dirichlet-style haplotype sampling (src/haplotype_select.cpp): gamma draws weighted by the k-mer support of each
// haplotype, then the `n` largest
struct HaplotypeSampler {
    std::vector<uint16_t> top;                      // ascending draw (order the reference's min-heap pops them)
    std::unordered_map<uint16_t, double> score;     // normalised over the selected ones

    HaplotypeSampler(const std::vector<uint32_t>& support, int n)
    {
        std::mt19937 prng(random_device_value());
        const size_t hap_num = support.size();
        std::vector<double> freq(hap_num, 0.0);
        double total = 0;
        for (size_t i = 0; i < hap_num; ++i) {
            if (support[i] == 0) continue;
            freq[i] = std::gamma_distribution<double>(support[i] + 1.0, 1)(prng);
            total += freq[i];
        }
        if (total > 0)
            for (auto& f : freq) f /= total;
        struct Greater {
            bool operator()(const std::pair<double, uint16_t>& a, const std::pair<double, uint16_t>& b) { return a.first > b.first; }
        };
        std::priority_queue<std::pair<double, uint16_t>, std::vector<std::pair<double, uint16_t>>, Greater> pq;
        double sum = 0.0;
        for (uint16_t i = 0; i < freq.size(); i++) {
            pq.push(std::make_pair(freq[i], i));
            sum += freq[i];
            if (pq.size() > (size_t)n) {
                sum -= pq.top().first;
                pq.pop();
            }
        }
        while (!pq.empty()) {
            top.push_back(pq.top().second);
            score[pq.top().second] = pq.top().first / sum;
            pq.pop();
        }
    }
};

// every genotype the HMM considers: multisets of `ploidy` selected haplotypes (diploid), or the blocks of `ploidy`
// consecutive haplotype indices that make one polyploid VCF sample (src/genotype.cpp:835-915)
std::vector<std::vector<uint16_t>> haplotype_combinations(const std::vector<uint16_t>& haps, const std::string& sample_type,
                                                          uint32_t ploidy, uint16_t max_hap_idx)
{
    std::vector<std::vector<uint16_t>> out;
    if (ploidy > 2) {
        for (const auto& hap : haps) {
            std::vector<uint16_t> v(ploidy);
            if (hap == 0) {
                v.assign(ploidy, 0);
            } else {
                const int32_t quotient = std::ceil(hap / (float)ploidy);
                const uint16_t first = (quotient - 1) * ploidy + 1;
                std::iota(v.begin(), v.end(), first);
                for (auto& x : v)
                    if (x > max_hap_idx) x = 0;
            }
            out.push_back(std::move(v));
        }
        std::set<std::vector<uint16_t>> uniq(out.begin(), out.end());
        out.assign(uniq.begin(), uniq.end());
        return out;
    }
    const uint32_t last = (uint32_t)haps.size() - 1;
    std::vector<std::vector<uint32_t>> idx;
    for (uint32_t i = 0; i < haps.size(); i++) {
        std::vector<uint32_t> v(ploidy, i);
        idx.push_back(v);
        if (sample_type == "hom" || ploidy < 2) continue;
        auto mn = std::min_element(v.begin() + 1, v.end());
        while (*mn < last) {
            uint32_t j = (uint32_t)v.size() - 1;
            while (v[j] == last) {
                v[j] = *mn + 1;
                j--;
            }
            v[j]++;
            idx.push_back(v);
            mn = std::min_element(v.begin() + 1, v.end());
        }
    }
    out.reserve(idx.size());
    for (const auto& v : idx) {
        std::vector<uint16_t> h;
        h.reserve(v.size());
        for (uint32_t i : v) h.push_back(haps[i]);
        out.push_back(std::move(h));
    }
    return out;
}

template <typename T>
std::string join_numbers(const std::vector<T>& v, const char* delim)
{
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) {
        s += std::to_string(v[i]);
        if (i + 1 != v.size()) s += delim;
    }
    return s;
}

std::string strip_newlines(const std::string& s)   // strip(str, '\n') of src/strip_split_join.cpp
{
    size_t i = 0, j = s.size();
    while (i < j && s[i] == '\n') ++i;
    while (j > i && s[j - 1] == '\n') --j;
    return s.substr(i, j - i);
}

}  // namespace

struct Genotyper::Run {
    // Everything per sample is in NODE ORDER: entry j belongs to place j of the graph2node lists (GraphIndex::node_key_index[j] is
    // its key) -- the order the device's per-node gather (K5, vgmi_counts_finish's cov_node) delivers and the order the windows
    // walk, so a node's k-mers are neighbours in memory instead of gathers over the key arrays.
    const uint8_t* cov;      // cov_node
    // coverage | multiplicity << 8 | haplotype bits << 16 of every entry in one word, when the bits fit; else nullptr
    const uint64_t* packed = nullptr;
    float hap_cov;
    const GenotypeConfig* cfg;
    uint32_t haploid_num;   // min(-n, #haplotypes)
};

// A window prepared for the device recursion: its genotypes, the emission scores of its nodes and the tables of powers of
// both directions; window_finish() turns the device's alpha / beta into the nodes' calls.
struct Genotyper::WindowWork {
    Chrom* chr = nullptr;
    bool on_device = false;
    std::vector<uint16_t> top;
    std::vector<std::vector<uint16_t>> genotypes;
    std::vector<uint8_t> keep_mat;
    size_t n_gt = 0;
    uint32_t ploidy = 0;
    std::vector<uint32_t> nodes;                  // nodes with emission scores, in position order
    // where this window's rows and steps go in the run's arrays (room for every node the HMM works on; the tail stays unused)
    size_t row0 = 0, step0 = 0, room = 0;
    long double* obs = nullptr;                   // row0 on: nodes.size() x n_gt emission scores
    long double* pw = nullptr;                    // step0 on: per step no_recomb^0..ploidy, recomb^0..ploidy
    uint32_t* row = nullptr;                      // step0 on: forward chain (the nodes in order), then backward chain (from the last)
    uint8_t* restart = nullptr;
    uint8_t *gid = nullptr, *order = nullptr;     // row0 on: per node the genotype string of every entry / the strings in string order
    uint64_t *fwd_step = nullptr, *bwd_step = nullptr;   // row0 on: the steps that hold the node's alpha / beta
};

Genotyper::Genotyper(const GraphIndex& g, unsigned threads) : g_(g)
{
    n_hap_ = (uint32_t)g.hap_names.size();
    for (const auto& kv : g.hap_names) hap_ids_.push_back(kv.first);
    // variant nodes in mGraphMap order carry the graph2node k-mer lists (CSR over key indices).  The maps are walked once for
    // the nodes' places; the half million small k-mer lists are then copied by `threads` workers.
    struct Place { uint32_t start; const GraphNode* gn; size_t v; };
    size_t v = 0;
    const auto t_ctor = std::chrono::steady_clock::now();
    kmer_pool_.reset(new uint32_t[g.node_off.empty() || g.node_off.back() == 0 ? 1 : g.node_off.back()]);
    double s_walk = 0, s_alloc = 0, s_fill = 0;
    auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    for (const auto& [chr, nodes] : g.graph_seq) {
        auto t_a = std::chrono::steady_clock::now();
        Chrom c;
        c.name = chr;
        auto it = g.chr_len.find(chr);
        c.len = it == g.chr_len.end() ? 0 : it->second;
        std::vector<Place> places;
        places.reserve(nodes.size());
        for (const GraphNode* gn : nodes) {
            size_t mine = SIZE_MAX;
            if (gn->hap_gt.size() != 1) {
                if (v + 1 >= g.node_off.size()) throw std::runtime_error("graph index: node list shorter than the graph");
                mine = v++;
            }
            places.push_back({gn->start, gn, mine});
        }
        s_walk += since(t_a);
        t_a = std::chrono::steady_clock::now();
        c.nodes.resize(places.size());
        s_alloc += since(t_a);
        t_a = std::chrono::steady_clock::now();
        auto fill = [&](size_t lo, size_t hi) {
            CpuBudget::Hold cpu;
            for (size_t i = lo; i < hi; ++i) {
                Node& n = c.nodes[i];
                n.start = places[i].start;
                n.gn = places[i].gn;
                if (places[i].v != SIZE_MAX) {      // places of the node's k-mers in the node-ordered arrays (not key indices)
                    n.kmers.p = kmer_pool_.get() + g.node_off[places[i].v];
                    n.kmers.n = (uint32_t)(g.node_off[places[i].v + 1] - g.node_off[places[i].v]);
                    std::iota(n.kmers.p, n.kmers.p + n.kmers.n, (uint32_t)g.node_off[places[i].v]);
                }
            }
        };
        const size_t nt = std::max<size_t>(1, std::min<size_t>(threads, places.size() / 4096 + 1));
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t) pool.emplace_back(fill, places.size() * t / nt, places.size() * (t + 1) / nt);
        fill(0, places.size() / nt);
        for (auto& th : pool) th.join();
        s_fill += since(t_a);
        chroms_.push_back(std::move(c));
    }
    if (getenv("VGH_TIMING"))
        std::fprintf(stderr, "[varigraph-mi] genotyper set-up %.3f s: nodes walked %.3f, allocated %.3f, k-mer lists %.3f\n", since(t_ctor), s_walk, s_alloc, s_fill);
}

// flanking sequence of a haplotype around a node: node_flanks.hpp (shared with `construct`)
std::pair<std::string, std::string> Genotyper::flanks(const Chrom& chr, uint32_t node_i, uint16_t hap, uint16_t alt_gt,
                                                      std::string& alt_seq, uint32_t want) const
{
    return node_flanks(chr.nodes, node_i, hap, alt_gt, alt_seq, want);
}

// ---------------------------------------------------------------- hidden states of one node (src/genotype.cpp:640-830)
// `genotypes` = the window's haplotype combinations (the same for every node of a window), `used` = the haplotypes
// occurring in them.  The per-haplotype term of a k-mer does not depend on the genotype, so it is evaluated once
// per (k-mer, haplotype) and summed per genotype.
// phase times summed over the pool's threads (VGH_TIMING): where the windows spend their time
namespace {
struct HmmPhases {
    std::atomic<long long> select{0}, states{0}, emit{0}, fwd{0}, bwd{0}, post{0};
    std::atomic<long long> list{0}, pass_a{0}, pass_b{0}, pass_c{0}, fill{0}, text{0};     // the device-emission path's host work, thread-seconds
};
HmmPhases g_phase;
const bool g_phase_on = getenv("VGH_TIMING") != nullptr;
struct PhaseTimer {
    std::atomic<long long>& acc;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(std::atomic<long long>& a) : acc(a) { if (g_phase_on) t0 = std::chrono::steady_clock::now(); }
    ~PhaseTimer() { if (g_phase_on) acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

namespace {
// h[g] = one16[pos_a[g]] + one16[pos_b[g]] for 16 genotypes per step (the copy count of every genotype of a window for one
// k-mer: the same byte sums as the scalar loop)
__attribute__((target("ssse3"))) void hrow_pairs_ssse3(const uint8_t* one16, const uint8_t* pos_a, const uint8_t* pos_b, uint8_t* out, size_t n16)
{
    const __m128i tab = _mm_loadu_si128(reinterpret_cast<const __m128i*>(one16));
    for (size_t g = 0; g < n16; g += 16) {
        const __m128i a = _mm_shuffle_epi8(tab, _mm_loadu_si128(reinterpret_cast<const __m128i*>(pos_a + g)));
        const __m128i b = _mm_shuffle_epi8(tab, _mm_loadu_si128(reinterpret_cast<const __m128i*>(pos_b + g)));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out + g), _mm_add_epi8(a, b));
    }
}
const bool g_have_ssse3 = [] {
    __builtin_cpu_init();
    return __builtin_cpu_supports("ssse3") != 0;
}();
}  // namespace

Genotyper::NodeStates Genotyper::hidden_states(Chrom& chr, uint32_t node_i, const std::vector<uint16_t>& top,
                                               const std::vector<std::vector<uint16_t>>& genotypes,
                                               const std::vector<uint16_t>& used, const GenotypeList& gl, double lower,
                                               double upper, bool filter, const Run& r, NodeStates&& recycled, const Node* ahead)
{
    Node& node = chr.nodes[node_i];
    const std::vector<uint16_t>& hap_gt = node.gn->hap_gt;
    const uint64_t bl = g_.bitlen;
    const uint32_t* const key_of = g_.node_key_index.data();      // place in the node-ordered arrays -> key
    auto hap_bit = [&](uint32_t key, uint16_t hap) -> uint8_t { return ((uint8_t)g_.bitvec[(size_t)key * bl + (hap >> 3)] >> (hap & 7)) & 1u; };
    auto last_bit = [&](uint32_t key) -> int { return ((uint8_t)g_.bitvec[(size_t)key * bl + bl - 1] >> 7) & 1; };

    const size_t n_gt = genotypes.size();
    NodeStates ns = std::move(recycled);   // the previous node's buffers
    ns.n_genotypes = n_gt;
    ns.c.clear();
    ns.f.clear();
    ns.h.clear();
    ns.c.reserve(node.kmers.size());
    ns.f.reserve(node.kmers.size());
    ns.h.resize(node.kmers.size() * n_gt);
    const std::vector<uint16_t>& flat = gl.flat;
    const std::vector<uint32_t>& flat_off = gl.off;
    const bool pairs = gl.pairs;
    const bool shuffle = pairs && !gl.pos_a.empty() && g_have_ssse3;
    size_t n_kept = 0;

    std::vector<uint32_t>& kept = ns.kept;        // the node's k-mers that take part (all of them unless `filter`)
    kept.clear();
    kept.reserve(node.kmers.size());
    std::map<uint16_t, uint32_t> need_sequence;   // haplotypes with multi-copy, under-covered k-mers: check their sequence
    std::vector<uint8_t>& one = ns.one;
    one.assign(n_hap_, 0);
    // the per-key arrays are indexed at random: the keys of the node ahead are asked for one per k-mer of this node (all at
    // once they would overrun the core's miss buffers and most of the requests would be dropped)
    const uint32_t* pf = ahead ? ahead->kmers.data() : nullptr;
    const uint32_t* const pf_end = ahead ? pf + ahead->kmers.size() : nullptr;
    auto prefetch_one = [&]() {
        if (pf != pf_end) {
            const uint32_t k2 = *pf++;
            if (r.packed) {
                __builtin_prefetch(&r.packed[k2]);
                return;
            }
            __builtin_prefetch(&r.cov[k2]);
            __builtin_prefetch(&g_.f[key_of[k2]]);
            __builtin_prefetch(&g_.bitvec[(size_t)key_of[k2] * bl]);
        }
    };
    // Fast path (haplotype bits of a k-mer fit one 64-bit word, genotypes are pairs over <= 16 haplotypes): the word is
    // read once per k-mer and every test is a mask or a shift of it
    const bool fast = shuffle && r.packed != nullptr;
    ns.cls.clear();
    ns.rep.clear();
    // haplotypes (positions in `used`) that carried the same k-mers so far: a partition refined k-mer by k-mer
    uint16_t part[16];
    uint32_t n_part = 1;
    part[0] = (uint16_t)((1u << used.size()) - 1u);
    uint64_t top_mask = 0;
    uint8_t gt0[16] = {0};      // haplotype used[p] carries the reference allele at this node
    if (fast) {
        for (uint16_t hap : top) top_mask |= 1ULL << hap;
        for (size_t p = 0; p < used.size(); ++p) gt0[p] = hap_gt[used[p]] == 0;
    }
    for (uint32_t pos : node.kmers) {
        prefetch_one();
        uint8_t* hrow = &ns.h[n_kept * n_gt];
        if (fast) {
            const uint64_t w = r.packed[pos];
            const uint8_t c = (uint8_t)w, f = (uint8_t)(w >> 8);
            const uint64_t bits = w >> 16;
            const int lb = (int)((bits >> (8 * bl - 1)) & 1u);
            if (filter && (bits & top_mask) == 0) continue;
            kept.push_back(pos);
            const bool in_interval = lb == 1 && c >= lower && c <= upper;
            uint8_t one16[16] = {0};
            uint32_t carried_mask = 0;
            for (size_t p = 0; p < used.size(); ++p) {
                one16[p] = (in_interval && gt0[p]) ? 1 : (uint8_t)((bits >> used[p]) & 1u);
                carried_mask |= (uint32_t)one16[p] << p;
            }
            for (uint32_t q = 0, nq = n_part; q < nq; ++q) {
                const uint16_t in = (uint16_t)(part[q] & carried_mask);
                if (in && in != part[q]) {
                    part[n_part++] = (uint16_t)(part[q] & ~carried_mask);
                    part[q] = in;
                }
            }
            if (c < lower && f >= 2)
                for (size_t p = 0; p < used.size(); ++p)
                    if (one16[p]) need_sequence.emplace(used[p], 0);
            ns.c.push_back(c);
            ns.f.push_back((lb == 1 && f == 1) ? (uint8_t)(f + 1) : f);
            const size_t g16 = n_gt & ~(size_t)15;
            hrow_pairs_ssse3(one16, gl.pos_a.data(), gl.pos_b.data(), hrow, g16);
            for (size_t gi = g16; gi < n_gt; ++gi) hrow[gi] = (uint8_t)(one16[gl.pos_a[gi]] + one16[gl.pos_b[gi]]);
            ++n_kept;
            continue;
        }
        const uint32_t key = key_of[pos];
        const uint8_t c = r.cov[pos];
        const uint8_t f = g_.f[key];
        const int lb = last_bit(key);
        if (filter) {
            uint64_t carried = 0;
            for (uint16_t hap : top) carried += hap_bit(key, hap);
            if (carried == 0) continue;
        }
        kept.push_back(pos);
        const bool in_interval = lb == 1 && c >= lower && c <= upper;
        for (uint16_t hap : used) {
            one[hap] = (in_interval && hap_gt[hap] == 0) ? 1 : hap_bit(key, hap);
            if (one[hap] > 0 && c < lower && f >= 2) need_sequence.emplace(hap, 0);
        }
        ns.c.push_back(c);
        ns.f.push_back((lb == 1 && f == 1) ? (uint8_t)(f + 1) : f);
        if (pairs) {
            for (size_t gi = 0; gi < n_gt; ++gi) hrow[gi] = (uint8_t)(one[flat[2 * gi]] + one[flat[2 * gi + 1]]);
        } else {
            for (size_t gi = 0; gi < n_gt; ++gi) {
                uint8_t h = 0;
                for (uint32_t t = flat_off[gi]; t < flat_off[gi + 1]; ++t) h += one[flat[t]];
                hrow[gi] = h;
            }
        }
        ++n_kept;
    }
    ns.h.resize(n_kept * n_gt);
    while (pf != pf_end) prefetch_one();

    if (fast && need_sequence.empty() && n_kept) {
        uint8_t group[16] = {0};
        for (uint32_t q = 0; q < n_part; ++q)
            for (size_t p = 0; p < used.size(); ++p)
                if ((part[q] >> p) & 1u) group[p] = (uint8_t)q;
        int16_t id_of[256];
        std::memset(id_of, 0xFF, sizeof id_of);
        ns.cls.resize(n_gt);
        for (size_t gi = 0; gi < n_gt; ++gi) {
            const uint8_t a = group[gl.pos_a[gi]], b = group[gl.pos_b[gi]];
            int16_t& id = id_of[a < b ? a * 16 + b : b * 16 + a];
            if (id < 0) {
                id = (int16_t)ns.rep.size();
                ns.rep.push_back((uint16_t)gi);
            }
            ns.cls[gi] = (uint16_t)id;
        }
    }
    if (!need_sequence.empty()) {
        std::unordered_map<uint16_t, std::unordered_set<uint64_t>> hap_keys;
        for (const auto& [hap, unused] : need_sequence) {
            (void)unused;
            const uint16_t gt = hap_gt[hap];
            if (gt >= node.gn->seqs.size())
                throw std::runtime_error("Node '" + chr.name + "-" + std::to_string(node.start) +
                                         "' does not contain sequence information for haplotype " + std::to_string(gt) + ".");
            std::string seq = node.gn->seqs[gt];
            const auto fl = flanks(chr, node_i, hap, gt, seq, g_.k - 1);
            seq = fl.first + seq + fl.second;
            hap_keys[hap] = sequence_keys(seq, g_.k);
        }
        uint32_t si = 0;
        for (uint32_t pos : kept) {
            const uint32_t key = key_of[pos];
            const uint8_t c = r.cov[pos];
            const uint8_t f = g_.f[key];
            if (c > lower || f <= 1) {
                si++;
                continue;
            }
            const int lb = last_bit(key);
            const uint64_t key_hash = g_.keys[key];
            for (size_t gi = 0; gi < n_gt; ++gi) {
                uint8_t& h = ns.h[(size_t)si * n_gt + gi];
                for (uint16_t hap : genotypes[gi]) {
                    const uint8_t o1 = (lb == 1 && hap_gt[hap] == 0 && c >= lower && c <= upper) ? 1 : hap_bit(key, hap);
                    auto it = hap_keys.find(hap);
                    if (it == hap_keys.end() || o1 == 0) continue;
                    if (o1 == 1 && it->second.find(key_hash) == it->second.end()) {
                        if (h >= 1) h--;
                    }
                }
            }
            si++;
        }
    }
    if (filter) {
        if (kept.size() != node.kmers.size()) lists_whole_.store(false, std::memory_order_relaxed);     // the device's emission path needs whole lists
        node.kmers.keep(kept);
    }
    return ns;
}

Genotyper::EmitPartPlan::~EmitPartPlan() { vgmi_hmm_plan_free(plan); }

// The sequence check of hidden_states() on its own (same conditions, same sets, same strings), for a node whose products stay on the
// device: which entries of the node's list lose which haplotypes.  The fast path's reading of an entry (one 64-bit word: coverage,
// multiplicity, haplotype bits) -- the device's emission kernel reads the same word the same way.
void Genotyper::sequence_fixes(const Chrom& chr, uint32_t node_i, const std::vector<uint16_t>& used, uint16_t gt0_mask, double lower, double upper,
                               const Run& r, std::vector<uint32_t>& fix_j, std::vector<uint16_t>& fix_mask) const
{
    const Node& node = chr.nodes[node_i];
    const std::vector<uint16_t>& hap_gt = node.gn->hap_gt;
    const uint64_t bl = g_.bitlen;
    const uint32_t* const key_of = g_.node_key_index.data();
    auto carried = [&](uint64_t w, uint32_t& low_multi) -> uint32_t {      // bits over `used`: the haplotypes that count as carrying the k-mer
        const uint8_t c = (uint8_t)w, f = (uint8_t)(w >> 8);
        const uint64_t bits = w >> 16;
        const int lb = (int)((bits >> (8 * bl - 1)) & 1u);
        const bool in_interval = lb == 1 && c >= lower && c <= upper;
        uint32_t om = 0;
        for (size_t p = 0; p < used.size(); ++p) om |= (uint32_t)((in_interval && ((gt0_mask >> p) & 1u)) ? 1u : (uint32_t)((bits >> used[p]) & 1u)) << p;
        low_multi = (c < lower && f >= 2) ? 2u : (!(c > lower || f <= 1)) ? 1u : 0u;      // 2: asks for the sequences; 1: is checked once they are there
        return om;
    };
    uint32_t need = 0;
    for (uint32_t pos : node.kmers) {
        uint32_t lm;
        const uint32_t om = carried(r.packed[pos], lm);
        if (lm == 2u) need |= om;
    }
    if (!need) return;
    // the needed haplotypes' sequences; haplotypes with the same allele and the same flanks share one (two sequences at most nodes)
    std::vector<std::pair<std::string, std::unordered_set<uint64_t>>> seqs;
    uint8_t which[16] = {0};
    for (size_t p = 0; p < used.size(); ++p) {
        if (!((need >> p) & 1u)) continue;
        const uint16_t hap = used[p], gt = hap_gt[hap];
        if (gt >= node.gn->seqs.size())
            throw std::runtime_error("Node '" + chr.name + "-" + std::to_string(node.start) + "' does not contain sequence information for haplotype " +
                                     std::to_string(gt) + ".");
        std::string seq = node.gn->seqs[gt];
        const auto fl = flanks(chr, node_i, hap, gt, seq, g_.k - 1);
        seq = fl.first + seq + fl.second;
        size_t at = 0;
        while (at < seqs.size() && seqs[at].first != seq) ++at;
        if (at == seqs.size()) {
            std::unordered_set<uint64_t> keys = sequence_keys(seq, g_.k);
            seqs.emplace_back(std::move(seq), std::move(keys));
        }
        which[p] = (uint8_t)at;
    }
    uint32_t j = 0;      // (an entry's index: 32 bits end to end, like entry_count -- graph2node keeps 128 k-mers a node, src/construct_index.cpp:1592-1596, but nothing here relies on it)
    for (uint32_t pos : node.kmers) {
        uint32_t lm;
        const uint32_t om = carried(r.packed[pos], lm) & need;
        if (lm != 0u && om != 0u) {
            const uint64_t key_hash = g_.keys[key_of[pos]];
            uint32_t drop = 0;
            for (size_t p = 0; p < used.size(); ++p)
                if (((om >> p) & 1u) && seqs[which[p]].second.find(key_hash) == seqs[which[p]].second.end()) drop |= 1u << p;
            if (drop) {
                fix_j.push_back(j);
                fix_mask.push_back((uint16_t)drop);
            }
        }
        ++j;
    }
}

// ---------------------------------------------------------------- posterior of one node (src/genotype.cpp:1387-1522)
void Genotyper::posterior(Node& n, const std::vector<uint16_t>& top, const Run& r) const
{
    const uint64_t bl = g_.bitlen;
    const uint32_t* const key_of = g_.node_key_index.data();
    uint8_t unique_kmers = 0;
    for (uint32_t pos : n.kmers) {
        if (g_.f[key_of[pos]] > 1) continue;
        if (unique_kmers < UINT8_MAX) unique_kmers++;
    }
    const auto& hap_gt = n.gn->hap_gt;

    // selected haplotype -> (#k-mers it carries, sum of their coverage); any other haplotype reads as (0, 0)
    std::vector<uint64_t> hap_num(n_hap_, 0), hap_sum(n_hap_, 0);
    for (uint32_t pos : n.kmers) {
        if (r.packed) {
            const uint64_t w = r.packed[pos];
            const uint8_t c = (uint8_t)w;
            const uint64_t bits = w >> 16;
            for (uint16_t hap : top) {
                if ((bits >> hap) & 1u) {
                    ++hap_num[hap];
                    hap_sum[hap] += c;
                }
            }
            continue;
        }
        const uint8_t c = r.cov[pos];
        const uint32_t key = key_of[pos];
        for (uint16_t hap : top) {
            if (((uint8_t)g_.bitvec[(size_t)key * bl + (hap >> 3)] >> (hap & 7)) & 1u) {
                ++hap_num[hap];
                hap_sum[hap] += c;
            }
        }
    }

    long double denominator = 0.0L;
    for (const auto& s : n.hmm) denominator += s.a * s.b;

    // The reference sums the posteriors per genotype STRING (alleles as decimal strings, sorted as strings, joined by
    // '/') in a std::map and takes the first maximum in key order.  Few distinct strings occur per node: every entry
    // gets the index of its string, the sums run in entry order exactly as the map's do, and the strings themselves
    // are only built once each.
    struct Distinct { std::string text; long double sum = 0.0L; };
    std::vector<Distinct> distinct;
    std::vector<int32_t> gid(n.hmm.size(), -1);
    uint16_t max_allele = 0;
    for (uint16_t a : hap_gt) max_allele = a > max_allele ? a : max_allele;
    std::vector<std::string> allele_text((size_t)max_allele + 1);   // decimal string per allele number, filled on demand
    std::vector<std::vector<uint16_t>> distinct_alleles;  // sorted (as strings) allele tuple of every distinct string
    std::vector<uint16_t> tuple;
    auto text_of = [&](uint16_t a) -> const std::string& {
        if (allele_text[a].empty()) allele_text[a] = std::to_string(a);
        return allele_text[a];
    };
    // the string of an entry depends on its alleles only: pairs of small allele numbers (the usual case) remember theirs
    const size_t na = (size_t)max_allele + 1;
    std::vector<int32_t> pair_id;
    if (na <= 64) pair_id.assign(na * na, -2);
    for (size_t i = 0; i < n.hmm.size(); ++i) {
        if (!n.hmm[i].haps || n.hmm[i].haps->empty()) continue;
        const auto& haps = *n.hmm[i].haps;
        int32_t* memo = nullptr;
        if (haps.size() == 2 && !pair_id.empty()) {
            memo = &pair_id[(size_t)hap_gt[haps[0]] * na + hap_gt[haps[1]]];
            if (*memo != -2) {
                gid[i] = *memo;
                continue;
            }
        }
        tuple.clear();
        for (uint16_t hap : haps) tuple.push_back(hap_gt[hap]);
        std::sort(tuple.begin(), tuple.end(), [&](uint16_t x, uint16_t y) { return text_of(x) < text_of(y); });
        int32_t id = -1;
        for (size_t d = 0; d < distinct_alleles.size(); ++d)
            if (distinct_alleles[d] == tuple) { id = (int32_t)d; break; }
        if (id < 0) {
            id = (int32_t)distinct.size();
            distinct_alleles.push_back(tuple);
            Distinct e;
            for (size_t t = 0; t < tuple.size(); ++t) {
                e.text += text_of(tuple[t]);
                if (t + 1 != tuple.size()) e.text += "/";
            }
            distinct.push_back(std::move(e));
        }
        gid[i] = id;
        if (memo) *memo = id;
    }
    // (a * b) / denominator of every entry: the same expression in both passes of the reference, evaluated once
    std::vector<long double> post(n.hmm.size());
    for (size_t i = 0; i < n.hmm.size(); ++i) {
        const auto& s = n.hmm[i];
        post[i] = (s.a * s.b) / (long double)denominator;
        if (gid[i] < 0) continue;
        distinct[(size_t)gid[i]].sum += post[i];
    }
    std::vector<size_t> order(distinct.size());
    for (size_t d = 0; d < order.size(); ++d) order[d] = d;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return distinct[x].text < distinct[y].text; });
    int32_t best_id = -1;
    long double best = -1.0;
    for (size_t d : order) {
        if (distinct[d].sum > best) {
            best = distinct[d].sum;
            best_id = (int32_t)d;
        }
    }

    // the entry with the largest posterior among those of the winning string (the first one on ties) makes the call
    long double max_post = 0.0L;
    size_t winner = n.hmm.size();
    for (size_t i = 0; i < n.hmm.size(); ++i) {
        if (gid[i] < 0 || gid[i] != best_id) continue;
        n.call.probability = best;
        if (max_post < post[i]) {
            max_post = post[i];
            winner = i;
        }
    }
    if (winner != n.hmm.size()) {
        n.call.haps = *n.hmm[winner].haps;
        n.call.kmer_num.clear();
        n.call.kmer_ave_cov.clear();
        for (uint16_t hap : n.call.haps) {
            const uint64_t num = hap < n_hap_ ? hap_num[hap] : 0;
            const uint64_t sum = hap < n_hap_ ? hap_sum[hap] : 0;
            const float ave = (num != 0) ? static_cast<float>(sum) / (float)num : 0.0;
            n.call.kmer_num.push_back(num);
            n.call.kmer_ave_cov.push_back(ave);
        }
        n.call.unique_kmers = unique_kmers;
    }
    std::vector<HmmScore>().swap(n.hmm);
}

// The per-key arrays (coverage, multiplicity, haplotype bits) are indexed by key number, i.e. at random: a node's keys are
// asked for one node ahead of their use
void Genotyper::prefetch_keys(const Node& n, const Run& r) const
{
    if (n.kmers.empty()) return;
    if (r.packed) {      // a node's entries are neighbours (unless an earlier sample pruned the list): one line per 8
        for (size_t j = 0; j < n.kmers.size(); j += 8) __builtin_prefetch(&r.packed[n.kmers[j]]);
        return;
    }
    const uint64_t bl = g_.bitlen;
    __builtin_prefetch(&r.cov[n.kmers[0]]);
    for (uint32_t pos : n.kmers) {
        const uint32_t key = g_.node_key_index[pos];
        __builtin_prefetch(&g_.f[key]);
        __builtin_prefetch(&g_.bitvec[(size_t)key * bl]);
    }
}


// ---------------------------------------------------------------- emission scores of one node (observable_states, :960-1000)
// Scratch and the sample's memoised libm values, one per thread that scores nodes.
struct Genotyper::ScoreCtx {
    float ave = 0;
    double score_up = 0;
    std::vector<long double> pois_tab = std::vector<long double>(256 * 256);
    std::vector<uint8_t> pois_have = std::vector<uint8_t>(256 * 256, 0);
    long double geo_tab[256];
    bool geo_have[256] = {false};
    std::vector<long double> term_buf, class_prod;
    std::vector<uint8_t> term_have;
};

std::vector<long double> Genotyper::score_states(const NodeStates& ns, ScoreCtx& sc) const
{
    const float ave = sc.ave;
    const double score_up = sc.score_up;
    std::vector<long double>& pois_tab = sc.pois_tab;
    std::vector<uint8_t>& pois_have = sc.pois_have;
    long double* const geo_tab = sc.geo_tab;
    bool* const geo_have = sc.geo_have;
    std::vector<long double>& term_buf = sc.term_buf;
    std::vector<long double>& class_prod = sc.class_prod;
    std::vector<uint8_t>& term_have = sc.term_have;
    std::vector<long double> obs;
    const size_t nk = ns.c.size(), ng = ns.n_genotypes;
    if (ng == 0 || nk == 0) return obs;
    // with classes of equivalent genotypes (hidden_states) every column of h equals its class's first member's:
    // those columns alone say which copy numbers occur
    const bool by_class = !ns.rep.empty();
    uint8_t max_h = 0;
    if (by_class) {
        for (size_t j = 0; j < nk; ++j)
            for (uint16_t g : ns.rep) max_h = std::max(max_h, ns.h[j * ng + g]);
    } else {
        for (uint8_t h : ns.h) max_h = h > max_h ? h : max_h;
    }
    const size_t hs = (size_t)max_h + 1;
    // the term of k-mer j under h copies, for the (j, h) that occur
    term_buf.resize(nk * hs);
    // which copy numbers occur for k-mer j: one pass over its row (a term is only evaluated for those, as the
    // reference evaluates it)
    term_have.assign(nk * hs, 0);
    for (size_t j = 0; j < nk; ++j) {
        const uint8_t* hj = &ns.h[j * ng];
        uint8_t* have = &term_have[j * hs];
        if (by_class) {
            for (uint16_t g : ns.rep) have[hj[g]] = 1;
        } else {
            for (size_t gi = 0; gi < ng; ++gi) have[hj[gi]] = 1;
        }
    }
    for (size_t j = 0; j < nk; ++j) {
        for (size_t hh = 0; hh < hs; ++hh) {
            if (!term_have[j * hs + hh]) continue;
            const uint8_t h = (uint8_t)hh;
            uint8_t c = ns.c[j];
            most_likely_depth(h, c, ns.f[j], ave, score_up);
            if (h == 0) {
                if (!geo_have[c]) {
                    geo_tab[c] = geometric(error_param(ave), c);
                    geo_have[c] = true;
                }
                term_buf[j * hs + h] = geo_tab[c];
            } else {
                const size_t slot = (size_t)h * 256 + c;
                if (!pois_have[slot]) {
                    pois_tab[slot] = poisson_pmf(ave * h, c);
                    pois_have[slot] = 1;
                }
                term_buf[j * hs + h] = pois_tab[slot];
            }
        }
    }
    obs.resize(ng);
    if (!ns.rep.empty()) {
        // one product per class of genotypes with the same column of h (the same factors in the same order give the
        // same bits), four classes in x87 registers at a time
        const size_t nc = ns.rep.size();
        std::vector<long double>& prod = class_prod;
        prod.resize(nc);
        size_t ci = 0;
        for (; ci + 4 <= nc; ci += 4) {
            const size_t g0 = ns.rep[ci], g1 = ns.rep[ci + 1], g2 = ns.rep[ci + 2], g3 = ns.rep[ci + 3];
            long double r0 = 1.0L, r1 = 1.0L, r2 = 1.0L, r3 = 1.0L;
            const uint8_t* hj = ns.h.data();
            const long double* t = term_buf.data();
            for (size_t j = 0; j < nk; ++j, hj += ng, t += hs) {
                r0 *= t[hj[g0]];
                r1 *= t[hj[g1]];
                r2 *= t[hj[g2]];
                r3 *= t[hj[g3]];
            }
            prod[ci] = r0;
            prod[ci + 1] = r1;
            prod[ci + 2] = r2;
            prod[ci + 3] = r3;
        }
        for (; ci < nc; ++ci) {
            long double res = 1.0L;
            const size_t g = ns.rep[ci];
            for (size_t j = 0; j < nk; ++j) res *= term_buf[j * hs + ns.h[j * ng + g]];
            prod[ci] = res;
        }
        for (size_t g = 0; g < ng; ++g) obs[g] = prod[ns.cls[g]];
        return obs;
    }
    size_t gi = 0;
    for (; gi + 4 <= ng; gi += 4) {   // four products in x87 registers, each in k-mer order
        long double r0 = 1.0L, r1 = 1.0L, r2 = 1.0L, r3 = 1.0L;
        const uint8_t* hj = &ns.h[gi];
        const long double* t = term_buf.data();
        for (size_t j = 0; j < nk; ++j, hj += ng, t += hs) {
            r0 *= t[hj[0]];
            r1 *= t[hj[1]];
            r2 *= t[hj[2]];
            r3 *= t[hj[3]];
        }
        obs[gi] = r0;
        obs[gi + 1] = r1;
        obs[gi + 2] = r2;
        obs[gi + 3] = r3;
    }
    for (; gi < ng; ++gi) {
        long double res = 1.0L;
        for (size_t j = 0; j < nk; ++j) res *= term_buf[j * hs + ns.h[j * ng + gi]];
        obs[gi] = res;
    }
    return obs;
}

// ---------------------------------------------------------------- one window: selection, forward, backward, posterior
void Genotyper::window(Chrom& chr, uint32_t first, uint32_t last, const Run& r, WindowWork* work, const std::vector<uint16_t>* forced_top)
{
    const GenotypeConfig& cfg = *r.cfg;
    auto vcf_chr = g_.vcf_info.find(chr.name);
    if (vcf_chr == g_.vcf_info.end()) throw std::runtime_error("'" + chr.name + "' does not exist in the VCF file.");
    if (first >= chr.nodes.size()) return;

    // ---- haplotype selection (src/genotype.cpp:500-610)
    std::unique_ptr<PhaseTimer> t_select(new PhaseTimer(g_phase.select));
    std::vector<uint16_t> top;
    if (n_hap_ <= r.haploid_num)
        for (const auto& kv : g_.hap_names) top.push_back(kv.first);
    std::vector<uint32_t> support(n_hap_, 0);
    const uint64_t bl = g_.bitlen;
    for (uint32_t i = first; i < last; ++i) {
        const Node& n = chr.nodes[i];
        if (n.gn->hap_gt.size() == 1) continue;
        {
            uint32_t nx = i + 1;
            while (nx < last && chr.nodes[nx].gn->hap_gt.size() == 1) ++nx;
            if (nx < last) prefetch_keys(chr.nodes[nx], r);
        }
        if (r.packed) {
            for (uint32_t pos : n.kmers) {
                const uint64_t w = r.packed[pos];
                const uint8_t c = (uint8_t)w;
                if (c <= 1 || (uint8_t)(w >> 8) > 1) continue;
                const uint64_t bits = w >> 16;
                for (const uint16_t hap : hap_ids_)
                    if ((bits >> hap) & 1u) support[hap] += c;
            }
            continue;
        }
        for (uint32_t pos : n.kmers) {
            const uint32_t key = g_.node_key_index[pos];
            const uint8_t c = r.cov[pos];
            if (c <= 1 || g_.f[key] > 1) continue;
            const uint8_t* bits = reinterpret_cast<const uint8_t*>(g_.bitvec.data()) + (size_t)key * bl;
            for (const uint16_t hap : hap_ids_)
                if ((bits[hap >> 3] >> (hap & 7)) & 1u) support[hap] += c;
        }
    }
    HaplotypeSampler sampler(support, (int)r.haploid_num);
    if (top.empty()) top = sampler.top;
    if (forced_top) top = *forced_top;
    std::sort(top.begin(), top.end());
    const std::unordered_map<uint16_t, double>& hap_score = sampler.score;
    t_select.reset();

    double lower = 256.0f, upper = -0.1f;
    poisson_interval(r.hap_cov, lower, upper);

    auto skipped = [&](const Node& n) -> bool {
        if (n.gn->hap_gt.size() <= 1) return true;
        if (cfg.sv_only) {
            auto site = vcf_chr->second.find(n.start);
            if (site == vcf_chr->second.end())
                throw std::runtime_error("'" + chr.name + ":" + std::to_string(n.start) + "' does not exist in the VCF file.");
            if (site->second[3].size() < 50 && site->second[4].size() < 50) return true;
        }
        return false;
    };
    // emission score of every genotype of a node (observable_states, :960-1000).  poisson(ave * h, c) and
    // geometric(p, c) are pure functions of (h, c) within a sample: evaluated once each (the values, and therefore
    // the products, are the reference's bit for bit)
    const float ave = r.hap_cov;
    double score_lo = 256.0f, score_up = -0.1f;
    poisson_interval(ave, score_lo, score_up);
    // genotypes of this window, the haplotypes they use, and how many haplotypes two genotypes share (the size of
    // std::set_intersection of the two sorted vectors)
    const std::vector<std::vector<uint16_t>> genotypes =
        haplotype_combinations(top, cfg.sample_type, cfg.sample_ploidy, (uint16_t)(n_hap_ - 1));
    const size_t n_gt = genotypes.size();
    std::vector<uint16_t> used;
    for (const auto& gtv : genotypes) used.insert(used.end(), gtv.begin(), gtv.end());
    std::sort(used.begin(), used.end());
    used.erase(std::unique(used.begin(), used.end()), used.end());
    auto shared = [](const std::vector<uint16_t>& a, const std::vector<uint16_t>& b) -> int32_t {
        int32_t n = 0;
        for (size_t x = 0, y = 0; x < a.size() && y < b.size();) {
            if (a[x] < b[y]) ++x;
            else if (b[y] < a[x]) ++y;
            else { ++n; ++x; ++y; }
        }
        return n;
    };
    GenotypeList glist;
    glist.off.assign(n_gt + 1, 0);
    for (size_t gi = 0; gi < n_gt; ++gi) {
        glist.flat.insert(glist.flat.end(), genotypes[gi].begin(), genotypes[gi].end());
        glist.off[gi + 1] = (uint32_t)glist.flat.size();
        glist.pairs = glist.pairs && genotypes[gi].size() == 2;
    }
    if (glist.pairs && used.size() <= 16) {
        std::unordered_map<uint16_t, uint8_t> where;
        for (size_t p = 0; p < used.size(); ++p) where[used[p]] = (uint8_t)p;
        glist.pos_a.resize(n_gt);
        glist.pos_b.resize(n_gt);
        for (size_t gi = 0; gi < n_gt; ++gi) {
            glist.pos_a[gi] = where[glist.flat[2 * gi]];
            glist.pos_b[gi] = where[glist.flat[2 * gi + 1]];
        }
    }
    bool all_full = true;   // every genotype has `ploidy` haplotypes
    for (const auto& gtv : genotypes) all_full = all_full && gtv.size() == (size_t)cfg.sample_ploidy;
    std::vector<uint8_t> keep_mat(n_gt * n_gt);
    for (size_t i = 0; i < n_gt; ++i)
        for (size_t j = 0; j < n_gt; ++j) keep_mat[i * n_gt + j] = (uint8_t)shared(genotypes[i], genotypes[j]);

    // emission score of every genotype of a node; empty when the node has no k-mer left (every genotype is skipped).
    // Every genotype's score is the product of its k-mers' terms in k-mer order (observable_states); walking the
    // k-mers in the outer loop keeps that order per genotype and turns one long dependent multiply chain per
    // genotype into n_genotypes independent ones.  A k-mer's term depends on the genotype only through h.
    ScoreCtx sctx;
    sctx.ave = ave;
    sctx.score_up = score_up;
    auto score_states = [&](const NodeStates& ns) -> std::vector<long double> { return this->score_states(ns, sctx); };
    // one step of the forward (alpha) or backward (beta) recursion (:1170-1380); obs[i] = emission of genotype i
    auto recursion = [&](const std::vector<HmmScore>& prev, bool use_alpha, long double recomb, long double no_recomb,
                         const std::vector<long double>& obs) -> std::vector<long double> {
        // pow(no_recomb, keep) and pow(recomb, change) take ploidy + 1 distinct values each per node
        const int32_t max_n = (int32_t)cfg.sample_ploidy;
        std::vector<long double> pow_keep(max_n + 1), pow_change(max_n + 1);
        const bool by_score = recomb == 0.0L && no_recomb == 0.0L;
        if (!by_score)
            for (int32_t i = 0; i <= max_n; ++i) {
                pow_keep[i] = std::pow(no_recomb, i);
                pow_change[i] = std::pow(recomb, i);
            }
        const bool aligned = prev.size() == n_gt;   // the previous node's entries are this window's genotypes, in order
        // (prev * pow(no_recomb, keep)) * pow(recomb, change) does not depend on the genotype being scored: one value
        // per previous entry and number of shared haplotypes -- the same two roundings, taken out of the inner loop
        std::vector<long double> step;
        if (!by_score && aligned) {
            step.resize(prev.size() * (size_t)(max_n + 1));
            for (size_t pi = 0; pi < prev.size(); ++pi) {
                const long double pv = use_alpha ? prev[pi].a : prev[pi].b;
                for (int32_t keep = 0; keep <= max_n; ++keep) {
                    const int32_t change = max_n - keep;   // every genotype has `ploidy` haplotypes
                    step[pi * (size_t)(max_n + 1) + keep] = pv * pow_keep[keep] * pow_change[change];
                }
            }
        }
        std::vector<long double> out;
        out.reserve(obs.size());
        long double total = 0.0L;
        size_t gi0 = 0;
        if (!prev.empty() && !by_score && aligned && all_full) {
            // three genotypes at a time: each sum still adds its terms in the order of the previous entries, but three
            // independent x87 add chains are in flight instead of one (three accumulators and their three emission
            // factors fill the register stack; a fourth chain would spill the factors to 80-bit memory operands)
            const size_t stride = (size_t)(max_n + 1);
            for (; gi0 + 3 <= obs.size(); gi0 += 3) {
                const uint8_t* k0 = &keep_mat[gi0 * n_gt];
                const uint8_t* k1 = k0 + n_gt;
                const uint8_t* k2 = k1 + n_gt;
                const long double o0 = obs[gi0], o1 = obs[gi0 + 1], o2 = obs[gi0 + 2];
                long double r0 = 0.0L, r1 = 0.0L, r2 = 0.0L;
                const long double* sp = step.data();
                for (size_t pi = 0; pi < prev.size(); ++pi, sp += stride) {
                    r0 += sp[k0[pi]] * o0;
                    r1 += sp[k1[pi]] * o1;
                    r2 += sp[k2[pi]] * o2;
                }
                out.push_back(r0);
                total += r0;
                out.push_back(r1);
                total += r1;
                out.push_back(r2);
                total += r2;
            }
        }
        for (size_t gi = gi0; gi < obs.size(); ++gi) {
            const std::vector<uint16_t>& haps = genotypes[gi];
            const int32_t hap_num = (int32_t)haps.size();
            long double res = 0.0L;
            if (prev.empty()) {
                res += obs[gi];
            } else if (!by_score && aligned && hap_num == max_n) {
                const uint8_t* km = &keep_mat[gi * n_gt];
                const long double o = obs[gi];
                const size_t stride = (size_t)(max_n + 1);
                for (size_t pi = 0; pi < prev.size(); ++pi) res += step[pi * stride + km[pi]] * o;
            } else {
                for (size_t pi = 0; pi < prev.size(); ++pi) {
                    const HmmScore& p = prev[pi];
                    const long double pv = use_alpha ? p.a : p.b;
                    if (by_score) {
                        long double t = pv * obs[gi];
                        for (uint16_t hap : haps) {
                            auto it = hap_score.find(hap);
                            if (it == hap_score.end())
                                throw std::runtime_error("'" + std::to_string(hap) + "' does not exist in 'hapIdxScoreMap'.");
                            t *= it->second;
                        }
                        res += t;
                    } else {
                        const int32_t keep = aligned ? (int32_t)keep_mat[gi * n_gt + pi] : shared(haps, *p.haps);
                        const int32_t change = hap_num - keep;
                        const long double pk = keep <= max_n ? pow_keep[keep] : std::pow(no_recomb, keep);
                        const long double pc = (change >= 0 && change <= max_n) ? pow_change[change] : std::pow(recomb, change);
                        res += pv * pk * pc * obs[gi];
                    }
                }
            }
            out.push_back(res);
            total += res;
        }
        if (total > 0.0L) {
            for (auto& x : out) x = x / total;
        } else {
            const long double uniform = 1.0L / (long double)out.size();
            for (auto& x : out) x = uniform;
        }
        return out;
    };

    // ---- forward.  The emissions are kept for the backward pass: it would recompute exactly the same hidden states
    // (it runs on the k-mer lists this pass has just pruned, with the same coverage and the same genotypes).
    // Device recursion: this pass fills hidden states, emission scores and the libm tables of every node; the recursion
    // itself and the posterior follow in window_finish() (the pruning of a node's k-mer list depends on the window's
    // haplotypes, not on alpha, so nothing here waits for the recursion)
    const bool to_device = work != nullptr && work->obs != nullptr && n_gt == work->n_gt && cfg.transition == "rec" && all_full && n_gt <= 2048 &&
                           cfg.sample_ploidy >= 1 && cfg.sample_ploidy <= 4;    // 2048: VGMI_HMM_MAX_GT (csrc/vgmi_kernels.h)
    if (to_device) {
        const uint32_t stride = cfg.sample_ploidy + 1;
        struct Seen { uint32_t start, end; int32_t at; };   // every node the HMM works on; at: its place in work->nodes or -1
        std::vector<Seen> seen;
        NodeStates st;
        for (uint32_t i = first; i < last; ++i) {
            Node& n = chr.nodes[i];
            if (skipped(n)) continue;
            const uint32_t n_start = n.start;
            const uint32_t n_end = (uint32_t)(n_start + n.gn->seqs[0].size() - 1);
            {
                PhaseTimer t(g_phase.states);
                uint32_t nx = i + 1;
                while (nx < last && skipped(chr.nodes[nx])) ++nx;
                st = hidden_states(chr, i, top, genotypes, used, glist, lower, upper, true, r, std::move(st), nx < last ? &chr.nodes[nx] : nullptr);
            }
            std::vector<long double> obs;
            {
                PhaseTimer t(g_phase.emit);
                obs = score_states(st);
            }
            if (obs.empty()) {
                seen.push_back(Seen{n_start, n_end, -1});
                continue;
            }
            seen.push_back(Seen{n_start, n_end, (int32_t)work->nodes.size()});
            if (work->nodes.size() >= work->room || obs.size() != n_gt) throw std::runtime_error("internal: device HMM window larger than announced");
            std::memcpy(work->obs + work->nodes.size() * n_gt, obs.data(), n_gt * sizeof(long double));
            if (!genotype_strings(n, genotypes, work->gid + work->nodes.size() * n_gt, work->order + work->nodes.size() * n_gt)) {
                // more distinct genotype strings at this node than the device's byte-sized ids hold (> 255: dozens of alleles
                // under hundreds of genotypes): the host takes the window, with the haplotypes already drawn
                work->nodes.clear();
                work->on_device = false;
                window(chr, first, last, r, nullptr, &top);
                return;
            }
            work->nodes.push_back(i);
        }
        const size_t m = work->nodes.size();
        auto powers = [&](long double* dst, uint32_t distance) {
            long double recomb, no_recomb;
            std::tie(recomb, no_recomb) = transition_probabilities(distance, (uint16_t)n_hap_);
            for (uint32_t k = 0; k < stride; ++k) {
                dst[k] = std::pow(no_recomb, (int32_t)k);
                dst[stride + k] = std::pow(recomb, (int32_t)k);
            }
        };
        for (size_t q = 0; q < seen.size(); ++q) {
            if (seen[q].at < 0) continue;
            const size_t j = (size_t)seen[q].at, fs = j, bs = m + (m - 1 - j);     // its forward and its backward step
            // forward: the node in front (prev_end = 0 in front of the first); the chain restarts behind a node without scores
            powers(work->pw + fs * 2 * stride, seen[q].start - (q ? seen[q - 1].end : 0u));
            work->restart[fs] = (q == 0 || seen[q - 1].at < 0) ? 1 : 0;
            work->row[fs] = (uint32_t)(work->row0 + j);
            // backward: the node behind (prev_start = 0 behind the last)
            powers(work->pw + bs * 2 * stride, (q + 1 < seen.size() ? seen[q + 1].start : 0u) - seen[q].end);
            work->restart[bs] = (q + 1 == seen.size() || seen[q + 1].at < 0) ? 1 : 0;
            work->row[bs] = (uint32_t)(work->row0 + j);
            work->fwd_step[j] = work->step0 + fs;
            work->bwd_step[j] = work->step0 + bs;
        }
        work->chr = &chr;
        work->on_device = true;
        work->n_gt = n_gt;
        work->ploidy = cfg.sample_ploidy;
        work->top = top;
        work->keep_mat = keep_mat;
        work->genotypes = genotypes;
        return;
    }
    std::vector<std::vector<long double>> emissions(last - first);
    const std::vector<HmmScore> no_prev;
    const std::vector<HmmScore>* prev = &no_prev;   // the entries of the node scored before this one
    uint32_t prev_start = 0, prev_end = 0;
    NodeStates states;
    for (uint32_t i = first; i < last; ++i) {
        Node& n = chr.nodes[i];
        if (skipped(n)) continue;
        const uint32_t n_start = n.start;
        const uint32_t n_end = (uint32_t)(n_start + n.gn->seqs[0].size() - 1);
        {
            PhaseTimer t(g_phase.states);
            uint32_t nx = i + 1;        // the node that is worked on next (nodes with one allele are passed over)
            while (nx < last && skipped(chr.nodes[nx])) ++nx;
            states = hidden_states(chr, i, top, genotypes, used, glist, lower, upper, true, r, std::move(states),
                                   nx < last ? &chr.nodes[nx] : nullptr);
        }
        long double recomb = 0.0L, no_recomb = 0.0L;
        if (cfg.transition == "rec") std::tie(recomb, no_recomb) = transition_probabilities(n_start - prev_end, (uint16_t)n_hap_);
        std::vector<long double>& obs = emissions[i - first];
        {
            PhaseTimer t(g_phase.emit);
            obs = score_states(states);
        }
        std::unique_ptr<PhaseTimer> t_fwd(new PhaseTimer(g_phase.fwd));
        const std::vector<long double> alpha = recursion(*prev, true, recomb, no_recomb, obs);
        t_fwd.reset();
        n.hmm.resize(alpha.size());
        for (size_t j = 0; j < alpha.size(); ++j) {
            n.hmm[j].a = alpha[j];
            n.hmm[j].haps = &genotypes[j];
        }
        prev_start = n_start;
        prev_end = n_end;
        prev = &n.hmm;
    }
    // ---- backward, and the posterior of a node as soon as the node in front of it has used its beta (its alpha / beta
    // entries are still in the cache then; the posterior of one node does not depend on any other's)
    prev = &no_prev;
    prev_start = 0;
    prev_end = 0;
    Node* due = nullptr;
    for (uint32_t i = last; i-- > first;) {
        Node& n = chr.nodes[i];
        if (skipped(n)) continue;
        prefetch_keys(n, r);
        const uint32_t n_start = n.start;
        const uint32_t n_end = (uint32_t)(n_start + n.gn->seqs[0].size() - 1);
        long double recomb = 0.0L, no_recomb = 0.0L;
        if (cfg.transition == "rec") std::tie(recomb, no_recomb) = transition_probabilities(prev_start - n_end, (uint16_t)n_hap_);
        std::unique_ptr<PhaseTimer> t_bwd(new PhaseTimer(g_phase.bwd));
        const std::vector<long double> beta = recursion(*prev, false, recomb, no_recomb, emissions[i - first]);
        t_bwd.reset();
        for (size_t j = 0; j < beta.size(); ++j) n.hmm[j].b = beta[j];
        std::vector<long double>().swap(emissions[i - first]);
        if (due) {
            PhaseTimer t(g_phase.post);
            posterior(*due, top, r);
        }
        due = &n;
        prev_start = n_start;
        prev_end = n_end;
        prev = &n.hmm;
    }
    (void)prev_end;
    if (due) {
        PhaseTimer t(g_phase.post);
        posterior(*due, top, r);
    }
}

// The genotype STRING of every entry of a node (posterior(): alleles as decimal strings, sorted as strings, joined by '/'),
// as small numbers in order of first appearance, and those numbers in string order (0xFF behind the last): what the
// device's posterior groups and ranks by.
bool Genotyper::genotype_strings(const Node& n, const std::vector<std::vector<uint16_t>>& genotypes, uint8_t* gid, uint8_t* order) const
{
    const auto& hap_gt = n.gn->hap_gt;
    uint16_t max_allele = 0;
    for (uint16_t a : hap_gt) max_allele = a > max_allele ? a : max_allele;
    std::vector<std::string> allele_text((size_t)max_allele + 1);
    auto text_of = [&](uint16_t a) -> const std::string& {
        if (allele_text[a].empty()) allele_text[a] = std::to_string(a);
        return allele_text[a];
    };
    std::vector<std::string> texts;
    std::vector<std::vector<uint16_t>> tuples;
    const size_t na = (size_t)max_allele + 1;
    std::vector<int32_t> pair_id;
    if (na <= 64) pair_id.assign(na * na, -2);
    std::vector<uint16_t> tuple;
    for (size_t i = 0; i < genotypes.size(); ++i) {
        const auto& haps = genotypes[i];
        int32_t* memo = nullptr;
        if (haps.size() == 2 && !pair_id.empty()) {
            memo = &pair_id[(size_t)hap_gt[haps[0]] * na + hap_gt[haps[1]]];
            if (*memo != -2) {
                gid[i] = (uint8_t)*memo;
                continue;
            }
        }
        tuple.clear();
        for (uint16_t hap : haps) tuple.push_back(hap_gt[hap]);
        std::sort(tuple.begin(), tuple.end(), [&](uint16_t x, uint16_t y) { return text_of(x) < text_of(y); });
        int32_t id = -1;
        for (size_t d = 0; d < tuples.size(); ++d)
            if (tuples[d] == tuple) { id = (int32_t)d; break; }
        if (id < 0) {
            id = (int32_t)tuples.size();
            if (id >= 255) return false;      // 0xFF ends the order list
            tuples.push_back(tuple);
            std::string t;
            for (size_t q = 0; q < tuple.size(); ++q) {
                t += text_of(tuple[q]);
                if (q + 1 != tuple.size()) t += "/";
            }
            texts.push_back(std::move(t));
        }
        gid[i] = (uint8_t)id;
        if (memo) *memo = id;
    }
    std::vector<size_t> by_text(texts.size());
    for (size_t d = 0; d < by_text.size(); ++d) by_text[d] = d;
    std::sort(by_text.begin(), by_text.end(), [&](size_t x, size_t y) { return texts[x] < texts[y]; });
    for (size_t q = 0; q < genotypes.size(); ++q) order[q] = q < by_text.size() ? (uint8_t)by_text[q] : 0xFF;
    return true;
}

// the device's verdict on the nodes of a window prepared by window(): probability of the winning genotype string and the
// entry that makes the call, per node; the rest of the call (k-mer counts of its haplotypes) as posterior() fills it
// tally / uniq (optional, diploid calls): per node of the window the device's tallies (vgmi_hmm_tallies) -- (k-mers, coverage sum) of
// the two called haplotypes and the count of single-copy k-mers -- instead of the walk over the node's k-mer list
void Genotyper::window_finish(WindowWork& w, const long double* prob, const uint32_t* winner, const Run& r, const uint32_t* tally, const uint8_t* uniq)
{
    const uint64_t bl = g_.bitlen;
    // posterior() tallies the SELECTED haplotypes (w.top); a called haplotype outside the selection -- the reference haplotype,
    // which every genotype list may hold, when -n picked fewer haplotypes than the panel has -- reads as (0, 0) there and here
    std::vector<uint8_t> selected(n_hap_, 0);
    for (uint16_t hap : w.top)
        if (hap < n_hap_) selected[hap] = 1;
    for (size_t j = 0; j < w.nodes.size(); ++j) {
        if (winner[j] >= w.n_gt) continue;            // no entry with a positive posterior: no call
        PhaseTimer t(g_phase.post);
        Node& n = w.chr->nodes[w.nodes[j]];
        // k-mer count and coverage sum of the CALLED haplotypes only (posterior() tallies every selected haplotype and then reads
        // the called ones: the same numbers for a seventh of the work at 15 haplotypes)
        const std::vector<uint16_t>& called = w.genotypes[winner[j]];
        uint64_t num[8] = {0}, sum[8] = {0};
        const size_t nc = std::min<size_t>(called.size(), 8);
        uint8_t unique_kmers = 0;
        const bool from_device = tally && called.size() == 2;
        if (from_device) {
            num[0] = tally[4 * j];
            sum[0] = tally[4 * j + 1];
            num[1] = tally[4 * j + 2];
            sum[1] = tally[4 * j + 3];
            unique_kmers = uniq[j];
        } else if (j + 1 < w.nodes.size()) prefetch_keys(w.chr->nodes[w.nodes[j + 1]], r);
        if (!from_device)
        for (uint32_t pos : n.kmers) {
            if (r.packed) {
                const uint64_t word = r.packed[pos];
                if ((uint8_t)(word >> 8) <= 1 && unique_kmers < UINT8_MAX) unique_kmers++;
                const uint8_t c = (uint8_t)word;
                const uint64_t bits = word >> 16;
                for (size_t q = 0; q < nc; ++q)
                    if (called[q] < n_hap_ && selected[called[q]] && ((bits >> called[q]) & 1u)) {
                        ++num[q];
                        sum[q] += c;
                    }
                continue;
            }
            const uint32_t key = g_.node_key_index[pos];
            if (g_.f[key] <= 1 && unique_kmers < UINT8_MAX) unique_kmers++;
            const uint8_t c = r.cov[pos];
            for (size_t q = 0; q < nc; ++q)
                if (called[q] < n_hap_ && selected[called[q]] && (((uint8_t)g_.bitvec[(size_t)key * bl + (called[q] >> 3)] >> (called[q] & 7)) & 1u)) {
                    ++num[q];
                    sum[q] += c;
                }
        }
        n.call.probability = prob[j];
        n.call.haps = called;
        n.call.kmer_num.clear();
        n.call.kmer_ave_cov.clear();
        for (size_t q = 0; q < called.size(); ++q) {
            const uint64_t nq = q < nc ? num[q] : 0, sq = q < nc ? sum[q] : 0;
            const float ave = (nq != 0) ? static_cast<float>(sq) / (float)nq : 0.0;
            n.call.kmer_num.push_back(nq);
            n.call.kmer_ave_cov.push_back(ave);
        }
        n.call.unique_kmers = unique_kmers;
    }
}

// ---------------------------------------------------------------- driver + VCF text
std::string Genotyper::run(const uint8_t* cov, float hap_kmer_coverage, const std::string& sample_name,
                           const GenotypeConfig& cfg, const uint8_t* cov_node)
{
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t n_entries = g_.node_key_index.size();
    std::vector<uint8_t> gathered;
    if (!cov_node) {      // a caller with per-key counters only (tests, the C API): the per-node gather on the host
        gathered.resize(n_entries);
        for (size_t j = 0; j < n_entries; ++j) gathered[j] = cov[g_.node_key_index[j]];
        cov_node = gathered.data();
    }
    Run r;
    r.cov = cov_node;
    r.hap_cov = hap_kmer_coverage;
    r.cfg = &cfg;
    r.haploid_num = std::min(cfg.haploid_num, n_hap_);
    if (g_.bitlen <= 6) {
        // one word per entry of the node lists: multiplicity and haplotype bits are the graph's (filled once, a gather over the
        // key arrays), the low byte is this sample's coverage -- a sequential pass over the device's cov_node
        const size_t bl = g_.bitlen;
        const bool first = packed_.size() != n_entries;
        if (first) {
            packed_.reserve(n_entries);
            advise_huge_pages(packed_.data(), n_entries * sizeof(uint64_t));
            packed_.resize(n_entries);
        }
        const uint32_t nt = std::max(1u, cfg.threads);
        std::vector<std::thread> fill;
        auto part = [&](size_t a, size_t b) {
            CpuBudget::Hold cpu;
            PhaseTimer tt(g_phase.fill);
            if (first && g_.entry_words.size() == n_entries) {      // the graph's half was gathered once for every Genotyper
                for (size_t j = a; j < b; ++j) packed_[j] = g_.entry_words[j] | cov_node[j];
            } else if (first) {
                for (size_t j = a; j < b; ++j) {
                    const size_t key = g_.node_key_index[j];
                    uint64_t bits = 0;
                    std::memcpy(&bits, &g_.bitvec[key * bl], bl);
                    packed_[j] = (uint64_t)cov_node[j] | (uint64_t)(uint8_t)g_.f[key] << 8 | bits << 16;
                }
            } else {
                for (size_t j = a; j < b; ++j) packed_[j] = (packed_[j] & ~(uint64_t)0xFF) | cov_node[j];
            }
        };
        for (uint32_t t = 1; t < nt; ++t) fill.emplace_back(part, n_entries * t / nt, n_entries * (t + 1) / nt);
        part(0, n_entries / nt);
        for (auto& th : fill) th.join();
        r.packed = packed_.data();
    }

    for (auto& c : chroms_)
        for (auto& n : c.nodes) {
            if (n.hmm.capacity()) std::vector<HmmScore>().swap(n.hmm);
            // (cleared, not replaced: the three small vectors of a call keep their storage from sample to sample)
            n.call.probability = 0;
            n.call.haps.clear();
            n.call.kmer_num.clear();
            n.call.kmer_ave_cov.clear();
            n.call.unique_kmers = 0;
        }

    // windows of `chr_len_thread` bp over the node list of every chromosome (src/genotype.cpp:76-140)
    struct Task { Chrom* chr; uint32_t first, last; };
    std::vector<Task> tasks;
    for (auto& chr : chroms_) {
        const uint64_t n_nodes = chr.nodes.size();
        if (g_.chr_len.find(chr.name) == g_.chr_len.end())
            throw std::runtime_error("'" + chr.name + "' does not exist in the reference genome.");
        const uint32_t chr_len = chr.len;
        const uint32_t step = std::min(cfg.chr_len_thread, chr_len);
        const uint32_t steps = (uint32_t)std::ceil(double(chr_len) / step);
        uint32_t end = 0;
        for (uint32_t i = 0; i < steps; i++) {
            const uint32_t step_end = (i + 1) * step;
            const uint32_t first = end;
            if (first >= n_nodes) break;
            for (uint32_t j = first; j < n_nodes; ++j) {
                if (chr.nodes[j].start > step_end) break;
                end++;
            }
            tasks.push_back({&chr, first, end});
        }
    }
    // With a device context (set_device; VGH_HMM_DEVICE=0 keeps the host): recursion and posterior of the eligible windows on the
    // device (window(), window_finish()).
    // The windows write their emission scores and step tables straight into the run's arrays: room for every node with more
    // than one allele is set aside per window (a node without k-mers leaves its row unused).
    const bool use_device = dev_ != nullptr && [] { const char* e = getenv("VGH_HMM_DEVICE"); return !(e && e[0] == '0'); }() &&
                            cfg.transition == "rec" && cfg.sample_ploidy >= 1 && cfg.sample_ploidy <= 4;
    std::vector<WindowWork> works(use_device ? tasks.size() : 0);
    size_t dev_n_gt = 0, total_room = 0;
    bool dev_emit = false;
    const uint32_t dev_stride = cfg.sample_ploidy + 1;
    struct Raw {
        void* p = nullptr;
        ~Raw() { std::free(p); }
    } raw_obs, raw_pw, raw_row, raw_restart, raw_gid, raw_order, raw_fs, raw_bs, raw_prob, raw_win;
    if (use_device) {
        std::vector<uint16_t> some(std::min<size_t>(r.haploid_num, n_hap_));
        for (size_t i = 0; i < some.size(); ++i) some[i] = (uint16_t)i;
        dev_n_gt = haplotype_combinations(some, cfg.sample_type, cfg.sample_ploidy, (uint16_t)(n_hap_ - 1)).size();
        for (size_t t = 0; t < tasks.size(); ++t) {
            size_t room = 0;
            for (uint32_t i = tasks[t].first; i < tasks[t].last; ++i) room += tasks[t].chr->nodes[i].gn->hap_gt.size() > 1;
            works[t].room = room;
            works[t].row0 = total_room;
            works[t].step0 = 2 * total_room;
            total_room += room;
        }
        // (the emission scores of all windows are held at once, on the host and -- with alpha and beta, three times that -- on
        // the device: beyond 16 GiB, a genome's worth of sites at 120 genotypes, the host runs the recursion as before;
        // VGH_HMM_DEVICE_GIB moves the bound)
        size_t score_gib = 16;
        if (const char* e = getenv("VGH_HMM_DEVICE_GIB")) score_gib = (size_t)std::max(0L, atol(e));
        // ... and what the device has free right now: a part holds its scores, alpha and beta (3 x its scores) until its calls are
        // back, all parts of a sample may be in flight at once, and 4 / dev_parts_ samples share the device (set_device)
        bool fits_device = true;
        {
            size_t free_b = 0, total_b = 0;
            const size_t need = 3 * total_room * dev_n_gt * sizeof(long double) * ((4 + dev_parts_ - 1) / dev_parts_) + (size_t(1) << 30);
            if (vgmi_device_memory(dev_, &free_b, &total_b) == VGMI_OK) fits_device = need <= free_b - free_b / 10;
            if (!fits_device && g_phase_on)
                std::fprintf(stderr, "[varigraph-mi] HMM on the host: %.1f GiB of device memory wanted, %.1f free\n", need / 1073741824.0, free_b / 1073741824.0);
        }
        const bool device_ok = fits_device && dev_n_gt >= 1 && dev_n_gt <= 2048 && total_room && total_room * dev_n_gt * sizeof(long double) <= (score_gib << 30);
        // the emission scores can be computed on the device as well (below): a diploid sample, every haplotype selected, whole lists
        // (round 5: polyploid samples too -- their genotypes are blocks of `ploidy` consecutive haplotypes, :846-873, a handful per window)
        dev_emit = device_ok && !emit_device_off_ && r.packed != nullptr && cfg.sample_ploidy >= 2 && cfg.sample_ploidy <= 4 && n_hap_ <= r.haploid_num && n_hap_ <= 16 &&
                   dev_n_gt <= 128 && lists_whole_.load() && [] { const char* e = getenv("VGH_HMM_EMIT_DEVICE"); return !(e && e[0] == '0'); }();
        if (device_ok && !dev_emit) {
            raw_obs.p = std::malloc(total_room * dev_n_gt * sizeof(long double));
            raw_pw.p = std::malloc(2 * total_room * 2 * dev_stride * sizeof(long double));
            raw_row.p = std::calloc(2 * total_room, sizeof(uint32_t));
            raw_restart.p = std::calloc(2 * total_room, 1);
            raw_gid.p = std::calloc(total_room * dev_n_gt, 1);
            raw_order.p = std::calloc(total_room * dev_n_gt, 1);
            raw_fs.p = std::calloc(total_room, sizeof(uint64_t));
            raw_bs.p = std::calloc(total_room, sizeof(uint64_t));
            raw_prob.p = std::malloc(total_room * sizeof(long double));
            raw_win.p = std::malloc(total_room * sizeof(uint32_t));
            if (!raw_obs.p || !raw_pw.p || !raw_row.p || !raw_restart.p || !raw_gid.p || !raw_order.p || !raw_fs.p || !raw_bs.p || !raw_prob.p || !raw_win.p)
                throw std::runtime_error("out of memory (HMM tables)");
            advise_huge_pages(raw_obs.p, total_room * dev_n_gt * sizeof(long double));       // gigabytes, first touched here and by the copies
            for (size_t t = 0; t < tasks.size(); ++t) {
                works[t].n_gt = dev_n_gt;      // a window whose genotype list has another length takes the host path
                works[t].obs = static_cast<long double*>(raw_obs.p) + works[t].row0 * dev_n_gt;
                works[t].pw = static_cast<long double*>(raw_pw.p) + works[t].step0 * 2 * dev_stride;
                works[t].row = static_cast<uint32_t*>(raw_row.p) + works[t].step0;
                works[t].restart = static_cast<uint8_t*>(raw_restart.p) + works[t].step0;
                works[t].gid = static_cast<uint8_t*>(raw_gid.p) + works[t].row0 * dev_n_gt;
                works[t].order = static_cast<uint8_t*>(raw_order.p) + works[t].row0 * dev_n_gt;
                works[t].fwd_step = static_cast<uint64_t*>(raw_fs.p) + works[t].row0;
                works[t].bwd_step = static_cast<uint64_t*>(raw_bs.p) + works[t].row0;
                // rows and steps a window leaves unused (nodes without k-mers) still point into its own part of the arrays
                std::fill(works[t].row, works[t].row + 2 * works[t].room, (uint32_t)works[t].row0);
                std::fill(works[t].fwd_step, works[t].fwd_step + works[t].room, (uint64_t)works[t].step0);
                std::fill(works[t].bwd_step, works[t].bwd_step + works[t].room, (uint64_t)works[t].step0);
            }
        }
    }
    const bool device_ready = raw_obs.p != nullptr;
    const uint32_t n_threads = std::max(1u, std::min<uint32_t>(cfg.threads, (uint32_t)tasks.size()));
    // ---- VCF (src/genotype.cpp:1579-1696): sites in vcf_info order, only those with a non-reference call.  The
    // reference walks mVcfInfoMap (chromosome, then position) and looks every site up in the graph; the windows are the
    // same nodes in the same order, so every task writes the lines of its own nodes and the pieces are joined in task
    // order (chromosomes of the graph that the VCF lacks have thrown in window() already).  A window's lines are written
    // as soon as its calls are known -- for most windows while the last chains are still on the device.
    std::vector<std::string> pieces(tasks.size());
    std::vector<uint8_t> piece_done(tasks.size(), 0);
    auto make_piece = [&](size_t t) {
        piece_done[t] = 1;
        const Chrom& chr = *tasks[t].chr;
        auto vc = g_.vcf_info.find(chr.name);
        if (vc == g_.vcf_info.end()) return;
        const auto& sites = vc->second;
        std::string out;
        std::vector<uint64_t> gt;
        // the window's nodes come in the order of their start, as the sites of the map do: one walk along the map instead of a
        // search from its root per node (0.4 thread-seconds per chr20-scale sample were these searches)
        auto site = tasks[t].first < tasks[t].last ? sites.lower_bound(chr.nodes[tasks[t].first].start) : sites.end();
        uint32_t prev_start = 0;
        for (uint32_t ni = tasks[t].first; ni < tasks[t].last; ++ni) {
            const Node& node = chr.nodes[ni];
            const SiteCall& call = node.call;
            if (node.start < prev_start) site = sites.lower_bound(node.start);      // (never, with node lists as graph.bin holds them)
            prev_start = node.start;
            while (site != sites.end() && site->first < node.start) ++site;
            if (call.haps.empty()) continue;
            if (site == sites.end() || site->first != node.start) continue;
            const auto& fields = site->second;
            gt.clear();
            bool all_ref = true;
            for (uint16_t hap : call.haps) {
                gt.push_back((uint64_t)node.gn->hap_gt[hap]);
                all_ref = all_ref && gt.back() == 0;
            }
            if (all_ref) continue;
            for (size_t i = 0; i < 9; i++) {
                if (i == 0) out += fields[i];
                else if (i == 6) out += "\tPASS";
                else if (i < 8) { out += '\t'; out += fields[i]; }
                else out += "\tGT:GQ:GPP:NAK:CAK:UK";
            }
            const float gq = phred_scaled(call.probability);
            const bool no_call = gq < cfg.min_gq;
            out += '\t';
            for (size_t i = 0; i < gt.size(); ++i) {
                if (i) out += '/';
                if (no_call) out += '.';
                else append_uint(out, gt[i]);
            }
            out += ':';
            append_fixed1(out, gq);
            out += ':';
            append_fixed1(out, call.probability);
            out += ':';
            for (size_t i = 0; i < call.kmer_num.size(); ++i) {
                if (i) out += ',';
                append_uint(out, call.kmer_num[i]);
            }
            out += ':';
            for (size_t i = 0; i < call.kmer_ave_cov.size(); i++) {
                if (i) out += ',';
                append_fixed1(out, call.kmer_ave_cov[i]);
            }
            out += ':';
            append_uint(out, call.unique_kmers);
            out += '\n';
        }
        pieces[t] = std::move(out);
    };
    // ---- three kinds of work on one pool: a window is prepared (window()), the recursion and posterior of a PART of the
    // windows run on the device (one call per part, on a thread of its own that mostly waits), the calls of a part's windows
    // are written back (window_finish()).  A chain is serial from its first node to its last, so the device takes as long
    // for ten windows as for all of them: the parts go to the device as soon as their windows are prepared, side by side
    // (vgmi_hmm_calls_part: own stream and buffers per call), while the pool prepares the next and finishes the last.
    // At most four parts, over all the samples genotyped at the same time (set_device): a process has four hardware queues by
    // default and a stream beyond them shares one, waiting behind the other stream's kernel for its whole length.
    const size_t max_parts = std::min<size_t>(4, dev_parts_);
    const size_t part_windows = std::max<size_t>(n_threads, (tasks.size() + max_parts - 1) / max_parts);
    const size_t n_parts = device_ready ? (tasks.size() + part_windows - 1) / part_windows : 0;
    std::vector<std::atomic<size_t>> part_done(n_parts);
    for (auto& d : part_done) d.store(0);
    std::vector<std::thread> part_threads(n_parts);
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::deque<size_t> finish_q;          // windows whose calls are back from the device
    size_t parts_open = n_parts;          // under q_mu
    std::atomic<size_t> next{0};
    std::string error;
    std::atomic<bool> failed{false};
    std::atomic<int64_t> dev_first{INT64_MAX}, dev_last{0};
    auto fail_with = [&](const char* what) {
        if (!failed.exchange(true)) error = what;
        std::lock_guard<std::mutex> lock(q_mu);
        q_cv.notify_all();
    };
    auto since_begin = [&]() { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count(); };
    auto run_part = [&](size_t part) {
        try {
            const size_t t0 = part * part_windows, t1 = std::min(tasks.size(), t0 + part_windows);
            std::vector<size_t> dw;
            for (size_t t = t0; t < t1; ++t)
                if (works[t].on_device && !works[t].nodes.empty()) dw.push_back(t);
            if (!dw.empty() && !failed.load()) {
                const size_t n = dev_n_gt;
                std::vector<uint8_t> keep(dw.size() * n * n);
                std::vector<vgmi_hmm_chain> chains;
                for (size_t wi = 0; wi < dw.size(); ++wi) {
                    const WindowWork& w = works[dw[wi]];
                    if (w.n_gt != n || w.ploidy != cfg.sample_ploidy) throw std::runtime_error("internal: a window with another genotype list");
                    std::memcpy(&keep[wi * n * n], w.keep_mat.data(), n * n);
                    const uint64_t m = w.nodes.size();
                    chains.push_back(vgmi_hmm_chain{w.step0, m, (uint32_t)wi, 0});
                    chains.push_back(vgmi_hmm_chain{w.step0 + m, m, (uint32_t)wi, 0});
                }
                const uint64_t row_lo = works[t0].row0, row_hi = works[t1 - 1].row0 + works[t1 - 1].room;
                const long double uniform = 1.0L / (long double)n;
                const int64_t ta = since_begin();
                const int rc = vgmi_hmm_calls_part(dev_, (uint32_t)n, cfg.sample_ploidy, keep.data(), (uint32_t)dw.size(), raw_obs.p, row_lo, row_hi,
                                        static_cast<const uint32_t*>(raw_row.p), static_cast<const uint8_t*>(raw_restart.p), raw_pw.p, 2 * row_lo,
                                        2 * row_hi, &uniform, chains.data(), (uint32_t)chains.size(), static_cast<const uint8_t*>(raw_gid.p),
                                        static_cast<const uint8_t*>(raw_order.p), static_cast<const uint64_t*>(raw_fs.p),
                                        static_cast<const uint64_t*>(raw_bs.p), raw_prob.p, static_cast<uint32_t*>(raw_win.p));
                if (rc == VGMI_E_NOMEM || (rc == VGMI_OK && getenv("VGH_HMM_FAKE_NOMEM") && part % 2 == 1)) {
                    // the device had no room for this part after all (other samples' parts, the table, the read buffers): the
                    // host runs these windows' recursion instead, with the haplotypes the first pass drew -- the sample is
                    // not lost, only slower
                    if (g_phase_on) std::fprintf(stderr, "[varigraph-mi] HMM part %zu: no device memory, %zu windows back on the host\n", part, dw.size());
                    for (size_t t : dw) {
                        CpuBudget::Hold cpu;
                        WindowWork& w = works[t];
                        w.on_device = false;
                        window(*tasks[t].chr, tasks[t].first, tasks[t].last, r, nullptr, &w.top);
                    }
                    dw.clear();
                } else if (rc != VGMI_OK)
                    throw std::runtime_error(std::string("device HMM recursion: ") + vgmi_last_error(dev_));
                const int64_t tb = since_begin();
                if (g_phase_on)
                    std::fprintf(stderr, "[varigraph-mi] HMM part %zu (windows %zu-%zu): on the device from %.3f to %.3f s\n", part, t0, t1 - 1, ta * 1e-9, tb * 1e-9);
                for (int64_t v = dev_first.load(); ta < v && !dev_first.compare_exchange_weak(v, ta);) {}
                for (int64_t v = dev_last.load(); tb > v && !dev_last.compare_exchange_weak(v, tb);) {}
            }
            std::lock_guard<std::mutex> lock(q_mu);
            for (size_t t : dw) finish_q.push_back(t);
            --parts_open;
            q_cv.notify_all();
        } catch (const std::exception& e) {
            fail_with(e.what());
        }
    };
    auto worker = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= tasks.size() || failed.load()) break;
            try {
                {
                    CpuBudget::Hold cpu;
                    window(*tasks[t].chr, tasks[t].first, tasks[t].last, r, device_ready ? &works[t] : nullptr);
                }
                if (device_ready) {
                    const size_t part = t / part_windows;
                    const size_t in_part = std::min(tasks.size(), (part + 1) * part_windows) - part * part_windows;
                    if (part_done[part].fetch_add(1) + 1 == in_part) part_threads[part] = std::thread(run_part, part);
                }
            } catch (const std::exception& e) {
                fail_with(e.what());
                return;
            }
        }
        for (;;) {      // nothing left to prepare: the calls that are back
            size_t t;
            {
                std::unique_lock<std::mutex> lock(q_mu);
                q_cv.wait(lock, [&] { return !finish_q.empty() || parts_open == 0 || failed.load(); });
                if (failed.load() || finish_q.empty()) return;
                t = finish_q.front();
                finish_q.pop_front();
            }
            try {
                CpuBudget::Hold cpu;
                WindowWork& w = works[t];
                window_finish(w, static_cast<const long double*>(raw_prob.p) + w.row0, static_cast<const uint32_t*>(raw_win.p) + w.row0, r);
                make_piece(t);
            } catch (const std::exception& e) {
                fail_with(e.what());
                return;
            }
        }
    };
    // ---- emission scores on the device too (vgmi_hmm_emissions): a diploid sample over a graph all of whose haplotypes are selected
    // (-n >= haplotypes: one genotype list for every window, no k-mer list is ever pruned).  Per part of the windows: the host lists
    // the nodes' entry ranges, the device scores them from the node-ordered coverage it was handed, the host scores the few nodes
    // whose haplotype sequences must be consulted, builds the step tables (libm) and the genotype strings, the device runs recursion
    // and posterior on the scores where they lie.  VGH_HMM_EMIT_DEVICE=0: the host prepares the scores as before.
    bool emitted_on_device = false;
    std::atomic<size_t> emit_windows_done{0};
    if (dev_emit) {
        const double tb0 = since_begin() * 1e-9;
        const float ave = r.hap_cov;
        double lower = 256.0f, upper = -0.1f;
        poisson_interval(ave, lower, upper);
        // one genotype list for the whole sample
        std::vector<uint16_t> top;
        for (const auto& kv : g_.hap_names) top.push_back(kv.first);
        std::sort(top.begin(), top.end());
        const std::vector<std::vector<uint16_t>> genotypes = haplotype_combinations(top, cfg.sample_type, cfg.sample_ploidy, (uint16_t)(n_hap_ - 1));
        const size_t n_gt = genotypes.size();
        std::vector<uint16_t> used;
        for (const auto& gtv : genotypes) used.insert(used.end(), gtv.begin(), gtv.end());
        std::sort(used.begin(), used.end());
        used.erase(std::unique(used.begin(), used.end()), used.end());
        bool pairs = true;      // (every genotype holds `ploidy` haplotypes: pairs for a diploid sample)
        for (const auto& gtv : genotypes) pairs = pairs && gtv.size() == cfg.sample_ploidy;
        if (pairs && n_gt >= 1 && n_gt <= 128 && used.size() <= 16) {
            GenotypeList glist;
            glist.off.assign(n_gt + 1, 0);
            std::vector<uint8_t> where(n_hap_ + 1, 0), used8(used.size());
            for (size_t p2 = 0; p2 < used.size(); ++p2) {
                where[used[p2]] = (uint8_t)p2;
                used8[p2] = (uint8_t)used[p2];
            }
            glist.pos_a.resize(n_gt);
            glist.pos_b.resize(n_gt);
            glist.pairs = cfg.sample_ploidy == 2;
            std::vector<uint8_t> pos_all(n_gt * cfg.sample_ploidy);      // per genotype its haplotypes' places in `used`
            for (size_t gi = 0; gi < n_gt; ++gi) {
                glist.flat.insert(glist.flat.end(), genotypes[gi].begin(), genotypes[gi].end());
                glist.off[gi + 1] = (uint32_t)glist.flat.size();
                glist.pos_a[gi] = where[genotypes[gi][0]];
                glist.pos_b[gi] = where[genotypes[gi][1]];
                for (uint32_t q = 0; q < cfg.sample_ploidy; ++q) pos_all[gi * cfg.sample_ploidy + q] = where[genotypes[gi][q]];
            }
            if (cfg.sample_ploidy != 2) {      // (hidden_states' pair shuffle is for pairs; the host-scored path of VGH_HMM_FIX_DEVICE=0 takes the general loop)
                glist.pos_a.clear();
                glist.pos_b.clear();
            }
            auto shared2 = [](const std::vector<uint16_t>& a, const std::vector<uint16_t>& b) -> uint8_t {
                uint8_t n2 = 0;
                for (size_t x = 0, y = 0; x < a.size() && y < b.size();) {
                    if (a[x] < b[y]) ++x;
                    else if (b[y] < a[x]) ++y;
                    else { ++n2; ++x; ++y; }
                }
                return n2;
            };
            std::vector<uint8_t> keep_mat(n_gt * n_gt);
            for (size_t i = 0; i < n_gt; ++i)
                for (size_t j = 0; j < n_gt; ++j) keep_mat[i * n_gt + j] = shared2(genotypes[i], genotypes[j]);
            uint64_t top_mask = 0;
            for (uint16_t hap : top) top_mask |= 1ULL << hap;
            // the sample's libm values: geometric(error_param(ave), c) for h = 0, poisson(ave * h, c) for h = 1 .. ploidy
            std::vector<long double> tab((size_t)(cfg.sample_ploidy + 1) * 256);
            for (int c2 = 0; c2 < 256; ++c2) {
                tab[c2] = geometric(error_param(ave), (uint8_t)c2);
                for (uint8_t h = 1; h <= cfg.sample_ploidy; ++h) tab[(size_t)h * 256 + c2] = poisson_pmf(ave * h, (uint8_t)c2);
            }
            if (!entries_uploaded_) {
                if (vgmi_hmm_entries_upload(dev_, packed_.data(), packed_.size()) != VGMI_OK) throw std::runtime_error(std::string("device HMM emissions: ") + vgmi_last_error(dev_));
                entries_uploaded_ = true;
            }
            if (vgmi_hmm_sample_upload(dev_, cov_node, n_entries) != VGMI_OK) throw std::runtime_error(std::string("device HMM emissions: ") + vgmi_last_error(dev_));
            const uint32_t stride = cfg.sample_ploidy + 1;
            const size_t max_parts_e = std::min<size_t>(4, dev_parts_);
            const size_t per_part = std::max<size_t>(1, (tasks.size() + max_parts_e - 1) / max_parts_e);
            const size_t n_parts_e = (tasks.size() + per_part - 1) / per_part;
            std::atomic<bool> broken{false};
            std::mutex err_mu;
            std::string err_text;
            const std::string cache_key = std::to_string(tasks.size()) + "/" + std::to_string(per_part) + "/" + std::to_string(cfg.sv_only) + "/" +
                                          std::to_string(cfg.sample_ploidy) + "/" + cfg.sample_type + "/" + std::to_string(n_gt) + "/" +
                                          std::to_string(r.haploid_num) + "/" + std::to_string(cfg.chr_len_thread);
            // What a part's device calls need beyond a sample's coverage is a function of the graph and the options: the rows (entry
            // ranges, reference-allele masks), and the PLAN -- genotype strings, which rows have a score, the step tables (libm), chains,
            // all of it resident on the device (vgmi_hmm_plan).  Made by the first sample that gets there, kept with the graph, shared
            // by every Genotyper of the run (eight consumers of one device built eight of everything in round 4).
            int dev_id = 0;
            (void)vgmi_device_of(dev_, &dev_id);
            if (emit_cache_.size() != n_parts_e) {
                emit_cache_.clear();
                emit_cache_.resize(n_parts_e);
            }
            for (size_t part = 0; part < n_parts_e; ++part) {
                const std::string slot = "emit/" + std::to_string(dev_id) + "/" + std::to_string(n_parts_e) + "/" + std::to_string(part) + "/" + cache_key;
                std::lock_guard<std::mutex> lock(g_.shared_mu);
                auto it = g_.shared_slots.find(slot);
                if (it == g_.shared_slots.end()) it = g_.shared_slots.emplace(slot, std::static_pointer_cast<void>(std::make_shared<EmitPartCache>())).first;
                emit_cache_[part] = std::static_pointer_cast<EmitPartCache>(it->second);
            }
            auto part_fn = [&](size_t part) {
                try {
                    const size_t t0 = part * per_part, t1 = std::min(tasks.size(), t0 + per_part);
                    EmitPartCache& pc = *emit_cache_[part];
                    // rows: every node the HMM works on, window after window (the same for every sample and every Genotyper: listed once)
                    {
                        std::lock_guard<std::mutex> lock(pc.mu);
                        if (pc.key != cache_key) {
                            CpuBudget::Hold cpu;
                            PhaseTimer t_list(g_phase.list);
                            pc.e_begin.clear(); pc.e_count.clear(); pc.row_node.clear(); pc.gt0.clear();
                            pc.plan.reset();
                            pc.win_row0.assign(t1 - t0 + 1, 0);
                            for (size_t t = t0; t < t1; ++t) {
                                Chrom& chr = *tasks[t].chr;
                                auto vcf_chr = g_.vcf_info.find(chr.name);
                                if (vcf_chr == g_.vcf_info.end()) throw std::runtime_error("'" + chr.name + "' does not exist in the VCF file.");
                                for (uint32_t i = tasks[t].first; i < tasks[t].last; ++i) {
                                    const Node& n = chr.nodes[i];
                                    if (n.gn->hap_gt.size() <= 1) continue;
                                    if (cfg.sv_only) {
                                        auto site = vcf_chr->second.find(n.start);
                                        if (site == vcf_chr->second.end())
                                            throw std::runtime_error("'" + chr.name + ":" + std::to_string(n.start) + "' does not exist in the VCF file.");
                                        if (site->second[3].size() < 50 && site->second[4].size() < 50) continue;
                                    }
                                    if (!n.kmers.empty() && (size_t)(n.kmers.back() - n.kmers.front()) + 1 != n.kmers.size()) {
                                        broken = true;      // a pruned list after all: the host path
                                        return;
                                    }
                                    pc.e_begin.push_back(n.kmers.empty() ? 0 : n.kmers.front());
                                    pc.e_count.push_back((uint32_t)n.kmers.size());
                                    uint16_t m = 0;
                                    for (size_t p2 = 0; p2 < used.size(); ++p2) m |= (uint16_t)((n.gn->hap_gt[used[p2]] == 0) << p2);
                                    pc.gt0.push_back(m);
                                    pc.row_node.push_back(i);
                                }
                                pc.win_row0[t - t0 + 1] = pc.e_begin.size();
                            }
                            pc.key = cache_key;
                        }
                    }
                    const std::vector<uint64_t>& e_begin = pc.e_begin;
                    const std::vector<uint32_t>&e_count = pc.e_count, &row_node = pc.row_node;
                    const std::vector<uint16_t>& gt0 = pc.gt0;
                    const std::vector<size_t>& win_row0 = pc.win_row0;
                    const size_t n_rows = e_begin.size();
                    if (n_rows == 0) return;      // no node of this part is the HMM's business (--sv over a part without long alleles): no calls, no lines
                    std::vector<uint32_t> n_kept(n_rows ? n_rows : 1);
                    std::vector<uint8_t> flags(n_rows ? n_rows : 1);
                    struct PartHandle {
                        vgmi_hmm_part* p = nullptr;
                        ~PartHandle() { vgmi_hmm_part_free(p); }
                    } ph;
                    const int64_t ta = since_begin();
                    int64_t t_emit = 0, t_a = 0, t_rows = 0, t_b = 0, t_calls = 0;
                    size_t n_fixed_rows = 0;
                    if (vgmi_hmm_emissions_ploidy(dev_, (uint32_t)n_gt, cfg.sample_ploidy, (uint32_t)used.size(), used8.data(), pos_all.data(), top_mask,
                                                  (uint32_t)g_.bitlen, ave, lower, upper, tab.data(), n_rows, e_begin.data(), e_count.data(), gt0.data(), n_kept.data(),
                                                  flags.data(), &ph.p) != VGMI_OK)
                        throw std::runtime_error(std::string("device HMM emissions: ") + vgmi_last_error(dev_));
                    t_emit = since_begin();
                    for (size_t rr = 0; rr < n_rows; ++rr)
                        if (flags[rr] & 2u) {
                            broken = true;          // a k-mer no haplotype carries: the host path prunes it
                            return;
                        }
                    // The part's windows on `helpers` threads.  A: the nodes whose haplotype sequences must be consulted -- which entries
                    // lose which haplotypes, for the device to score those rows again.  (Once per graph, under the part's lock: the
                    // genotype strings, which nodes have a score at all, B: the step tables (libm) -- the plan.)  Then recursion and
                    // posterior on the device.  C: the calls and the VCF lines.
                    const size_t nw = t1 - t0;
                    const size_t helpers = std::max<size_t>(1, std::min<size_t>(nw, n_threads / n_parts_e));
                    auto over_windows = [&](std::atomic<long long>& spent, const std::function<void(size_t)>& fn) {
                        std::atomic<size_t> nextw{0};
                        std::string herr;
                        std::mutex hmu;
                        auto body = [&]() {
                            for (;;) {
                                const size_t wi = nextw.fetch_add(1);
                                if (wi >= nw) return;
                                try {
                                    CpuBudget::Hold cpu;
                                    PhaseTimer tt(spent);
                                    fn(wi);
                                } catch (const std::exception& e) {
                                    std::lock_guard<std::mutex> lock(hmu);
                                    if (herr.empty()) herr = e.what();
                                }
                            }
                        };
                        std::vector<std::thread> hs;
                        for (size_t h2 = 1; h2 < helpers; ++h2) hs.emplace_back(body);
                        body();
                        for (auto& th : hs) th.join();
                        if (!herr.empty()) throw std::runtime_error(herr);
                    };
                    std::vector<std::vector<uint64_t>> host_rows(nw);
                    std::vector<std::vector<long double>> host_obs(nw);
                    static const bool fix_on_device = !(getenv("VGH_HMM_FIX_DEVICE") && getenv("VGH_HMM_FIX_DEVICE")[0] == '0');
                    std::vector<std::vector<uint64_t>> fix_rows(nw);
                    std::vector<std::vector<uint32_t>> fix_cnt(nw);
                    std::vector<std::vector<uint32_t>> fix_j(nw);
                    std::vector<std::vector<uint16_t>> fix_mask(nw);
                    over_windows(g_phase.pass_a, [&](size_t wi) {
                        Chrom& chr = *tasks[t0 + wi].chr;
                        ScoreCtx sctx;
                        sctx.ave = ave;
                        sctx.score_up = upper;
                        NodeStates st;
                        for (size_t rr = win_row0[wi]; rr < win_row0[wi + 1]; ++rr) {
                            if (!(flags[rr] & 1u)) continue;
                            if (fix_on_device) {
                                // a multi-copy, under-covered, carried k-mer: the reference consults the haplotype's sequence (:760-800).
                                // The strings are the host's; the row's products stay on the device, scored again below with the
                                // haplotypes the sequences rule out taken off the entries concerned
                                PhaseTimer tt(g_phase.states);
                                const size_t before = fix_j[wi].size();
                                sequence_fixes(chr, row_node[rr], used, gt0[rr], lower, upper, r, fix_j[wi], fix_mask[wi]);
                                if (fix_j[wi].size() != before) {
                                    fix_rows[wi].push_back(rr);
                                    fix_cnt[wi].push_back((uint32_t)(fix_j[wi].size() - before));
                                }
                            } else {
                                // ... VGH_HMM_FIX_DEVICE=0: scored by the host (hidden states, products in x87 arithmetic), handed in as rows
                                {
                                    PhaseTimer tt(g_phase.states);
                                    st = hidden_states(chr, row_node[rr], top, genotypes, used, glist, lower, upper, true, r, std::move(st), nullptr);
                                }
                                PhaseTimer tt(g_phase.emit);
                                const std::vector<long double> obs = score_states(st, sctx);
                                n_kept[rr] = (uint32_t)(obs.empty() ? 0 : st.c.size());
                                if (!obs.empty()) {
                                    host_rows[wi].push_back(rr);
                                    host_obs[wi].insert(host_obs[wi].end(), obs.begin(), obs.end());
                                }
                            }
                        }
                    });
                    t_a = since_begin();
                    {
                        std::vector<uint64_t> all_rows;
                        std::vector<long double> all_obs;
                        for (size_t wi = 0; wi < nw; ++wi) {
                            all_rows.insert(all_rows.end(), host_rows[wi].begin(), host_rows[wi].end());
                            all_obs.insert(all_obs.end(), host_obs[wi].begin(), host_obs[wi].end());
                            std::vector<long double>().swap(host_obs[wi]);
                        }
                        if (!all_rows.empty() && vgmi_hmm_part_set_rows(ph.p, all_rows.size(), all_rows.data(), all_obs.data()) != VGMI_OK)
                            throw std::runtime_error(std::string("device HMM emissions: ") + vgmi_last_error(dev_));
                        std::vector<uint64_t> f_rows;
                        std::vector<uint32_t> f_off(1, 0);
                        std::vector<uint32_t> f_j;
                        std::vector<uint16_t> f_m;
                        for (size_t wi = 0; wi < nw; ++wi) {
                            f_rows.insert(f_rows.end(), fix_rows[wi].begin(), fix_rows[wi].end());
                            for (uint32_t cnt2 : fix_cnt[wi]) f_off.push_back(f_off.back() + cnt2);
                            f_j.insert(f_j.end(), fix_j[wi].begin(), fix_j[wi].end());
                            f_m.insert(f_m.end(), fix_mask[wi].begin(), fix_mask[wi].end());
                        }
                        n_fixed_rows = f_rows.size();
                        if (!f_rows.empty() && vgmi_hmm_part_fix_rows(ph.p, f_rows.size(), f_rows.data(), f_off.data(), f_j.data(), f_m.data()) != VGMI_OK)
                            throw std::runtime_error(std::string("device HMM emissions: ") + vgmi_last_error(dev_));
                    }
                    t_rows = since_begin();
                    // ---- the plan: for the rows that have a score -- the same for every sample of this path (a row's kept k-mers are
                    // those some haplotype carries, and every haplotype is selected); held against the pattern all the same
                    std::vector<uint8_t> scored(n_rows);
                    for (size_t rr = 0; rr < n_rows; ++rr) scored[rr] = n_kept[rr] != 0;
                    std::shared_ptr<EmitPartPlan> plan;
                    {
                        std::lock_guard<std::mutex> lock(pc.mu);
                        if (!pc.plan || pc.plan->scored != scored) {
                            auto np = std::make_shared<EmitPartPlan>();
                            np->scored = scored;
                            np->win_nodes.resize(nw);
                            np->win_rows.resize(nw);
                            struct Seen { uint32_t start, end; int64_t row; };
                            std::vector<std::vector<Seen>> seen(nw);
                            std::vector<uint8_t> gid(n_rows * n_gt, 0), order(n_rows * n_gt, 0);
                            over_windows(g_phase.pass_a, [&](size_t wi) {
                                Chrom& chr = *tasks[t0 + wi].chr;
                                // the genotype strings of a node with two alleles depend on which haplotypes carry the reference allele only:
                                // one evaluation per distinct mask (the strings themselves as genotype_strings builds them)
                                std::unordered_map<uint32_t, uint32_t> gs_memo;      // mask -> a row that holds the pattern
                                for (size_t rr = win_row0[wi]; rr < win_row0[wi + 1]; ++rr) {
                                    Node& n = chr.nodes[row_node[rr]];
                                    const uint32_t n_start = n.start, n_end = (uint32_t)(n_start + n.gn->seqs[0].size() - 1);
                                    if (!scored[rr]) {
                                        seen[wi].push_back(Seen{n_start, n_end, -1});
                                        continue;
                                    }
                                    seen[wi].push_back(Seen{n_start, n_end, (int64_t)rr});
                                    bool biallelic = true;
                                    for (uint16_t hap : used) biallelic = biallelic && n.gn->hap_gt[hap] <= 1;
                                    auto it = biallelic ? gs_memo.find(gt0[rr]) : gs_memo.end();
                                    if (it != gs_memo.end()) {
                                        std::memcpy(gid.data() + rr * n_gt, gid.data() + (size_t)it->second * n_gt, n_gt);
                                        std::memcpy(order.data() + rr * n_gt, order.data() + (size_t)it->second * n_gt, n_gt);
                                    } else {
                                        (void)genotype_strings(n, genotypes, gid.data() + rr * n_gt, order.data() + rr * n_gt);     // <= 128 strings: always fits
                                        if (biallelic) gs_memo.emplace(gt0[rr], (uint32_t)rr);
                                    }
                                    np->win_nodes[wi].push_back(row_node[rr]);
                                    np->win_rows[wi].push_back((uint32_t)rr);
                                }
                            });
                            {   // the lines' shared heads, window by window (one walk along the chromosome's site map per window)
                                std::vector<std::string> heads(nw);
                                std::vector<std::vector<uint32_t>> head_len(nw);
                                over_windows(g_phase.pass_a, [&](size_t wi) {
                                    const Chrom& chr = *tasks[t0 + wi].chr;
                                    auto vc = g_.vcf_info.find(chr.name);
                                    std::string& out = heads[wi];
                                    head_len[wi].assign(win_row0[wi + 1] - win_row0[wi], 0);
                                    if (vc == g_.vcf_info.end()) return;
                                    const auto& sites = vc->second;
                                    auto site = win_row0[wi] < win_row0[wi + 1] ? sites.lower_bound(chr.nodes[row_node[win_row0[wi]]].start) : sites.end();
                                    uint32_t prev_start = 0;
                                    for (size_t rr = win_row0[wi]; rr < win_row0[wi + 1]; ++rr) {
                                        const Node& node = chr.nodes[row_node[rr]];
                                        if (node.start < prev_start) site = sites.lower_bound(node.start);
                                        prev_start = node.start;
                                        while (site != sites.end() && site->first < node.start) ++site;
                                        if (site == sites.end() || site->first != node.start) continue;
                                        const auto& fields = site->second;
                                        const size_t before = out.size();
                                        for (size_t i = 0; i < 9; i++) {
                                            if (i == 0) out += fields[i];
                                            else if (i == 6) out += "\tPASS";
                                            else if (i < 8) { out += '\t'; out += fields[i]; }
                                            else out += "\tGT:GQ:GPP:NAK:CAK:UK";
                                        }
                                        out += '\t';
                                        head_len[wi][rr - win_row0[wi]] = (uint32_t)(out.size() - before);
                                    }
                                });
                                np->line_head_off.assign(n_rows + 1, 0);
                                size_t total = 0;
                                for (size_t wi = 0; wi < nw; ++wi) total += heads[wi].size();
                                np->line_head.reserve(total);
                                for (size_t wi = 0; wi < nw; ++wi) {
                                    for (size_t rr = win_row0[wi]; rr < win_row0[wi + 1]; ++rr)
                                        np->line_head_off[rr + 1] = np->line_head_off[rr] + head_len[wi][rr - win_row0[wi]];
                                    np->line_head += heads[wi];
                                    std::string().swap(heads[wi]);
                                }
                            }
                            std::vector<size_t> win_step0(nw + 1, 0);
                            for (size_t wi = 0; wi < nw; ++wi) win_step0[wi + 1] = win_step0[wi] + 2 * np->win_rows[wi].size();
                            const size_t n_steps = win_step0[nw];
                            np->n_steps = n_steps;
                            if (n_steps) {
                                std::vector<long double> pw(n_steps * 2 * stride);
                                std::vector<uint32_t> row(n_steps, 0);
                                std::vector<uint8_t> restart(n_steps, 0);
                                std::vector<uint64_t> fwd(n_rows, 0), bwd(n_rows, 0);
                                std::vector<vgmi_hmm_chain> chains;
                                for (size_t wi = 0; wi < nw; ++wi) {
                                    const size_t m = np->win_rows[wi].size();
                                    if (!m) continue;
                                    chains.push_back(vgmi_hmm_chain{win_step0[wi], m, 0, 0});
                                    chains.push_back(vgmi_hmm_chain{win_step0[wi] + m, m, 0, 0});
                                }
                                over_windows(g_phase.pass_b, [&](size_t wi) {
                                    const size_t m = np->win_rows[wi].size();
                                    if (!m) return;
                                    const size_t step0 = win_step0[wi];
                                    // the tables of powers are a function of the distance alone: neighbouring nodes are tens to hundreds of bases
                                    // apart, so a window's few thousand steps share a few hundred distinct tables (same libm calls, each made once)
                                    constexpr uint32_t kMemo = 4096;
                                    std::vector<long double> memo((size_t)kMemo * 2 * stride);
                                    std::vector<uint8_t> memo_have(kMemo, 0);
                                    auto powers = [&](long double* dst, uint32_t distance) {
                                        if (distance < kMemo && memo_have[distance]) {
                                            std::memcpy(dst, &memo[(size_t)distance * 2 * stride], 2 * stride * sizeof(long double));
                                            return;
                                        }
                                        long double recomb, no_recomb;
                                        std::tie(recomb, no_recomb) = transition_probabilities(distance, (uint16_t)n_hap_);
                                        for (uint32_t k = 0; k < stride; ++k) {
                                            dst[k] = std::pow(no_recomb, (int32_t)k);
                                            dst[stride + k] = std::pow(recomb, (int32_t)k);
                                        }
                                        if (distance < kMemo) {
                                            std::memcpy(&memo[(size_t)distance * 2 * stride], dst, 2 * stride * sizeof(long double));
                                            memo_have[distance] = 1;
                                        }
                                    };
                                    const std::vector<Seen>& sn = seen[wi];
                                    size_t j = 0;
                                    for (size_t q = 0; q < sn.size(); ++q) {
                                        if (sn[q].row < 0) continue;
                                        const size_t fs = step0 + j, bs = step0 + m + (m - 1 - j);
                                        powers(pw.data() + fs * 2 * stride, sn[q].start - (q ? sn[q - 1].end : 0u));
                                        restart[fs] = (q == 0 || sn[q - 1].row < 0) ? 1 : 0;
                                        row[fs] = (uint32_t)sn[q].row;
                                        powers(pw.data() + bs * 2 * stride, (q + 1 < sn.size() ? sn[q + 1].start : 0u) - sn[q].end);
                                        restart[bs] = (q + 1 == sn.size() || sn[q + 1].row < 0) ? 1 : 0;
                                        row[bs] = (uint32_t)sn[q].row;
                                        fwd[sn[q].row] = fs;
                                        bwd[sn[q].row] = bs;
                                        ++j;
                                    }
                                });
                                const long double uniform = 1.0L / (long double)n_gt;
                                if (vgmi_hmm_plan_create(dev_, (uint32_t)n_gt, cfg.sample_ploidy, keep_mat.data(), 1, n_rows, row.data(), restart.data(), pw.data(), n_steps,
                                                         &uniform, chains.data(), (uint32_t)chains.size(), gid.data(), order.data(), fwd.data(), bwd.data(), &np->plan) != VGMI_OK)
                                    throw std::runtime_error(std::string("device HMM plan: ") + vgmi_last_error(dev_));
                            }
                            pc.plan = np;
                        }
                        plan = pc.plan;
                    }
                    const std::vector<std::vector<uint32_t>>&win_nodes = plan->win_nodes, &win_rows = plan->win_rows;
                    const size_t n_steps = plan->n_steps;
                    t_b = since_begin();
                    std::vector<long double> prob(n_rows ? n_rows : 1);
                    std::vector<uint32_t> winner(n_rows ? n_rows : 1, 0xFFFFFFFFu);
                    if (n_steps && vgmi_hmm_part_calls_plan(ph.p, plan->plan, prob.data(), winner.data()) != VGMI_OK)
                        throw std::runtime_error(std::string("device HMM recursion: ") + vgmi_last_error(dev_));
                    const int64_t tbb = since_begin();
                    t_calls = tbb;
                    if (g_phase_on)
                        std::fprintf(stderr, "[varigraph-mi] HMM part %zu (windows %zu-%zu): emissions, recursion and posterior on the device from %.3f to %.3f s (%zu of %zu nodes scored by the host, %zu scored again on the device): "
                                     "emission kernel %.3f, sequence checks (host) %.3f, rows fixed on the device %.3f, plan (strings + step tables: once per graph) %.3f, recursion + posterior %.3f\n",
                                     part, t0, t1 - 1, ta * 1e-9, tbb * 1e-9, [&] { size_t c2 = 0; for (auto& v : host_rows) c2 += v.size(); return c2; }(), n_rows, n_fixed_rows,
                                     (t_emit - ta) * 1e-9, (t_a - t_emit) * 1e-9, (t_rows - t_a) * 1e-9, (t_b - t_rows) * 1e-9, (t_calls - t_b) * 1e-9);
                    for (int64_t v = dev_first.load(); ta < v && !dev_first.compare_exchange_weak(v, ta);) {}
                    for (int64_t v = dev_last.load(); tbb > v && !dev_last.compare_exchange_weak(v, tbb);) {}
                    // the calls' k-mer tallies on the device too (the node lists and the sample's coverage are there for the emissions):
                    // per sample, the walk over every called node's k-mer list was 0.8 of 1.7 host thread-seconds (VGH_DEVICE_TALLIES=0: the walk)
                    std::vector<uint32_t> tally;
                    std::vector<uint8_t> tally_uniq;
                    static const bool device_tallies = !(getenv("VGH_DEVICE_TALLIES") && getenv("VGH_DEVICE_TALLIES")[0] == '0');
                    if (device_tallies && n_steps && cfg.sample_ploidy == 2 && n_hap_ <= 64 && n_gt <= 128) {
                        std::vector<uint8_t> hap_ab(2 * n_gt, 0xFF);
                        bool pairs = true;
                        for (size_t g2 = 0; g2 < n_gt; ++g2) {
                            if (genotypes[g2].size() != 2 || genotypes[g2][0] > 254 || genotypes[g2][1] > 254) { pairs = false; break; }
                            hap_ab[2 * g2] = (uint8_t)genotypes[g2][0];
                            hap_ab[2 * g2 + 1] = (uint8_t)genotypes[g2][1];
                        }
                        uint64_t sel = 0;
                        for (uint16_t hap : top)
                            if (hap < n_hap_ && hap < 64) sel |= 1ull << hap;
                        if (pairs) {
                            tally.resize(4 * n_rows);
                            tally_uniq.resize(n_rows);
                            if (vgmi_hmm_tallies(dev_, n_rows, e_begin.data(), e_count.data(), winner.data(), (uint32_t)n_gt, hap_ab.data(), n_hap_, sel, tally.data(),
                                                 tally_uniq.data()) != VGMI_OK)
                                throw std::runtime_error(std::string("device tallies: ") + vgmi_last_error(dev_));
                        }
                    }
                    over_windows(g_phase.pass_c, [&](size_t wi) {
                        if (!tally.empty()) {
                            // a diploid sample with the calls' tallies from the device: the line is written straight from what came back
                            // -- the called genotype's two haplotypes, their k-mer counts and coverage sums, the posterior -- without the
                            // detour through the nodes' call records (window_finish writes them, make_piece reads them back: two walks over
                            // half a million scattered nodes per sample)
                            piece_done[t0 + wi] = 1;
                            const Chrom& chr = *tasks[t0 + wi].chr;
                            const std::vector<uint32_t>&nodes_w = win_nodes[wi], &rows_w = win_rows[wi];
                            std::string out;
                            size_t room = 0;
                            for (size_t q = 0; q < rows_w.size(); ++q) room += (size_t)(plan->line_head_off[rows_w[q] + 1] - plan->line_head_off[rows_w[q]]) + 48;
                            out.reserve(room);
                            for (size_t q = 0; q < rows_w.size(); ++q) {
                                const size_t rw = rows_w[q];
                                if (winner[rw] >= n_gt) continue;            // no entry with a positive posterior: no call
                                const uint64_t h0 = plan->line_head_off[rw], h1 = plan->line_head_off[rw + 1];
                                if (h0 == h1) continue;                      // no such site in the VCF
                                const Node& node = chr.nodes[nodes_w[q]];
                                const std::vector<uint16_t>& called = genotypes[winner[rw]];
                                const uint64_t ga = node.gn->hap_gt[called[0]], gb = node.gn->hap_gt[called[1]];
                                if (ga == 0 && gb == 0) continue;
                                out.append(plan->line_head, h0, h1 - h0);
                                const long double pr = prob[rw];
                                const float gq = phred_scaled(pr);
                                if (gq < cfg.min_gq) out += "./.";
                                else {
                                    append_uint(out, ga);
                                    out += '/';
                                    append_uint(out, gb);
                                }
                                out += ':';
                                append_fixed1(out, gq);
                                out += ':';
                                append_fixed1(out, pr);
                                out += ':';
                                const uint32_t* tl = &tally[4 * rw];
                                append_uint(out, tl[0]);
                                out += ',';
                                append_uint(out, tl[2]);
                                out += ':';
                                append_fixed1(out, tl[0] ? static_cast<float>((uint64_t)tl[1]) / (float)(uint64_t)tl[0] : 0.0f);
                                out += ',';
                                append_fixed1(out, tl[2] ? static_cast<float>((uint64_t)tl[3]) / (float)(uint64_t)tl[2] : 0.0f);
                                out += ':';
                                append_uint(out, tally_uniq[rw]);
                                out += '\n';
                            }
                            pieces[t0 + wi] = std::move(out);
                            emit_windows_done += !nodes_w.empty();
                            return;
                        }
                        WindowWork w;
                        w.chr = tasks[t0 + wi].chr;
                        w.n_gt = n_gt;
                        w.genotypes = genotypes;
                        w.top = top;
                        w.nodes = win_nodes[wi];
                        std::vector<long double> pr(w.nodes.size());
                        std::vector<uint32_t> wn(w.nodes.size());
                        std::vector<uint32_t> tl(tally.empty() ? 0 : 4 * w.nodes.size());
                        std::vector<uint8_t> tu(tally.empty() ? 0 : w.nodes.size());
                        for (size_t q = 0; q < w.nodes.size(); ++q) {
                            const size_t rw = win_rows[wi][q];
                            pr[q] = prob[rw];
                            wn[q] = winner[rw];
                            if (!tally.empty()) {
                                std::memcpy(&tl[4 * q], &tally[4 * rw], 16);
                                tu[q] = tally_uniq[rw];
                            }
                        }
                        window_finish(w, pr.data(), wn.data(), r, tally.empty() ? nullptr : tl.data(), tally.empty() ? nullptr : tu.data());
                        // the window's lines (make_piece's, for the scored nodes -- no other node has a call -- with the head of every
                        // line taken from the plan instead of the site map)
                        {
                            piece_done[t0 + wi] = 1;
                            const Chrom& chr = *tasks[t0 + wi].chr;
                            std::string out;
                            size_t room = 0;
                            for (size_t q = 0; q < w.nodes.size(); ++q) room += (size_t)(plan->line_head_off[win_rows[wi][q] + 1] - plan->line_head_off[win_rows[wi][q]]) + 48;
                            out.reserve(room);
                            for (size_t q = 0; q < w.nodes.size(); ++q) {
                                const Node& node = chr.nodes[w.nodes[q]];
                                const SiteCall& call = node.call;
                                if (call.haps.empty()) continue;
                                const size_t rw = win_rows[wi][q];
                                const uint64_t h0 = plan->line_head_off[rw], h1 = plan->line_head_off[rw + 1];
                                if (h0 == h1) continue;      // no such site in the VCF
                                bool all_ref = true;
                                for (uint16_t hap : call.haps) all_ref = all_ref && node.gn->hap_gt[hap] == 0;
                                if (all_ref) continue;
                                out.append(plan->line_head, h0, h1 - h0);
                                const float gq = phred_scaled(call.probability);
                                const bool no_call = gq < cfg.min_gq;
                                for (size_t i = 0; i < call.haps.size(); ++i) {
                                    if (i) out += '/';
                                    if (no_call) out += '.';
                                    else append_uint(out, (uint64_t)node.gn->hap_gt[call.haps[i]]);
                                }
                                out += ':';
                                append_fixed1(out, gq);
                                out += ':';
                                append_fixed1(out, call.probability);
                                out += ':';
                                for (size_t i = 0; i < call.kmer_num.size(); ++i) {
                                    if (i) out += ',';
                                    append_uint(out, call.kmer_num[i]);
                                }
                                out += ':';
                                for (size_t i = 0; i < call.kmer_ave_cov.size(); i++) {
                                    if (i) out += ',';
                                    append_fixed1(out, call.kmer_ave_cov[i]);
                                }
                                out += ':';
                                append_uint(out, call.unique_kmers);
                                out += '\n';
                            }
                            pieces[t0 + wi] = std::move(out);
                        }
                        emit_windows_done += !w.nodes.empty();
                    });
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lock(err_mu);
                    if (err_text.empty()) err_text = e.what();
                }
            };
            std::vector<std::thread> pthreads;
            for (size_t p2 = 1; p2 < n_parts_e; ++p2) pthreads.emplace_back(part_fn, p2);
            if (n_parts_e) part_fn(0);
            for (auto& th : pthreads) th.join();
            if (!err_text.empty()) throw std::runtime_error(err_text);
            if (broken.load()) {
                // a list was pruned after all (or would be): this graph takes the host's preparation from now on
                emit_device_off_ = true;
                return run(cov, hap_kmer_coverage, sample_name, cfg, cov_node);
            }
            emitted_on_device = true;
            if (g_phase_on) std::fprintf(stderr, "[varigraph-mi] HMM emissions on the device: %zu parts, %.2f s\n", n_parts_e, since_begin() * 1e-9 - tb0);
        }
    }
    if (!emitted_on_device) {
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    for (auto& th : part_threads)
        if (th.joinable()) th.join();
    if (failed.load()) throw std::runtime_error(error);
    }
    last_device_seconds = dev_last.load() > 0 ? (double)(dev_last.load() - dev_first.load()) * 1e-9 : 0;
    last_windows = tasks.size();
    last_device_windows = emitted_on_device ? emit_windows_done.load() : 0;
    for (const auto& w : works) last_device_windows += w.on_device && !w.nodes.empty();
    const auto t_hmm = std::chrono::steady_clock::now();
    last_hmm_seconds = std::chrono::duration<double>(t_hmm - t_begin).count();
    if (g_phase_on) {
        std::fprintf(stderr, "[varigraph-mi] HMM thread-seconds: selection %.2f, hidden states %.2f, emissions %.2f, forward %.2f, backward %.2f, posterior %.2f (wall %.2f on %u threads)\n",
                     g_phase.select.exchange(0) * 1e-9, g_phase.states.exchange(0) * 1e-9, g_phase.emit.exchange(0) * 1e-9, g_phase.fwd.exchange(0) * 1e-9,
                     g_phase.bwd.exchange(0) * 1e-9, g_phase.post.exchange(0) * 1e-9, last_hmm_seconds, n_threads);
        if (last_device_seconds > 0) std::fprintf(stderr, "[varigraph-mi] HMM recursion on the device: %.2f s\n", last_device_seconds);
        if (emitted_on_device)
            std::fprintf(stderr, "[varigraph-mi] host thread-seconds around the device (all samples in flight since the last line of this kind): node lists %.2f, "
                         "host-scored nodes + genotype strings %.2f, step tables %.2f, calls + VCF lines %.2f, coverage words %.2f, text joined %.2f, in line for a thread %.2f\n",
                         g_phase.list.exchange(0) * 1e-9, g_phase.pass_a.exchange(0) * 1e-9, g_phase.pass_b.exchange(0) * 1e-9, g_phase.pass_c.exchange(0) * 1e-9,
                         g_phase.fill.exchange(0) * 1e-9, g_phase.text.exchange(0) * 1e-9, g_cpu_wait_ns.exchange(0) * 1e-9);
    }

    // ---- the pieces that are not written yet (windows without a device call), then the whole text
    {
        std::atomic<size_t> next_text{0};
        auto text_worker = [&]() {
            for (;;) {
                const size_t t = next_text.fetch_add(1);
                if (t >= tasks.size()) return;
                if (!piece_done[t]) {
                    CpuBudget::Hold cpu;
                    make_piece(t);
                }
            }
        };
        std::vector<std::thread> tpool;
        for (uint32_t t = 1; t < n_threads; ++t) tpool.emplace_back(text_worker);
        text_worker();
        for (auto& th : tpool) th.join();
    }
    // the graph's chromosomes are a std::map like mVcfInfoMap: the tasks are already in the reference's output order
    CpuBudget::Hold cpu;
    PhaseTimer t_text(g_phase.text);
    std::ostringstream oss;
    oss << g_.vcf_head + "\t" + sample_name + "\n";
    for (const auto& piece : pieces) oss << piece;
    // SAVE::save strips the newlines around each 10 MB chunk and adds one back (src/save.cpp:16-24); on the whole
    // text that is: no leading newline, exactly one trailing
    std::string text = strip_newlines(oss.str());
    if (!text.empty()) text += "\n";
    last_text_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_hmm).count();
    return text;
}

// SAVE writes the text through gzwrite (src/save.cpp:11-30); the bytes of a .gz depend on the zlib build anyway, what
// has to be identical is the content.  Here the text goes out as block gzip (BGZF: gzip members of <= 64 KiB with the
// 'BC' extra field, the form bgzip / tabix expect for VCFs) and the blocks are deflated by `threads` workers.
void Genotyper::write_gz(const std::string& path, const std::string& text, unsigned threads)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("'" + path + "': No such file or directory or possibly reached the maximum open file limit.");
    constexpr size_t kBlock = 0xff00;
    const size_t n_blocks = (text.size() + kBlock - 1) / kBlock;
    std::vector<std::string> out(n_blocks);
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto worker = [&]() {
        std::vector<unsigned char> buf(compressBound(kBlock) + 64);
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= n_blocks) return;
            CpuBudget::Hold cpu;
            const size_t off = b * kBlock, len = std::min(kBlock, text.size() - off);
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { failed = true; return; }
            zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(text.data() + off));
            zs.avail_in = (uInt)len;
            zs.next_out = buf.data();
            zs.avail_out = (uInt)buf.size();
            const int rc = deflate(&zs, Z_FINISH);
            const size_t clen = buf.size() - zs.avail_out;
            deflateEnd(&zs);
            if (rc != Z_STREAM_END || clen + 26 > 65536) { failed = true; return; }
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef*>(text.data() + off), (uInt)len);
            const uint16_t bsize = (uint16_t)(clen + 25);
            std::string& o = out[b];
            o.reserve(clen + 26);
            static const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            o.append(reinterpret_cast<const char*>(head), 16);
            o.push_back((char)(bsize & 0xFF));
            o.push_back((char)(bsize >> 8));
            o.append(reinterpret_cast<const char*>(buf.data()), clen);
            for (uint32_t v : {crc, (uint32_t)len})
                for (int sh = 0; sh < 32; sh += 8) o.push_back((char)((v >> sh) & 0xFF));
        }
    };
    const unsigned n_threads = (unsigned)std::max<size_t>(1, std::min<size_t>(threads ? threads : 1, n_blocks));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    bool ok = !failed.load();
    for (size_t b = 0; ok && b < n_blocks; ++b) ok = fwrite(out[b].data(), 1, out[b].size(), f) == out[b].size();
    static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ok = ok && fwrite(eof_block, 1, sizeof eof_block, f) == sizeof eof_block;
    ok = (fclose(f) == 0) && ok;
    if (!ok) throw std::runtime_error("'" + path + "': write error");
}

}  // namespace vgh
