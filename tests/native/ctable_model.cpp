// ctable_model.cpp -- host-side model of the CONTEXT TABLE (varigraph_amd/csrc/vgmi_ctable.h): the entry algebra the device
// code uses -- ct_make_from_unitig, ct_orient, ct_match, ct_id, ct_window_kmer, the bucket / mark / overflow trail -- driven
// by a plain C++ restatement of the build (unitigs of the key set, one entry per occurrence of a 16-mer) and of a read's grid
// walk, and held against a brute-force dictionary count of every 27-mer of every read.  Test infrastructure: nothing here is
// product code; the kernels (vgmi_ctable.hip) are checked against the oracle on the GPU (tests/test_gpu_large.py,
// test_gpu_parity.py).
//
//   ctable_model <seed> <genome> <variants> <reads> <load percent> [repeat copies] [k = 27 | 19 | 21 | 23 | 25]
// k < 27 (round 5): flanks of F = k - 16 bases, F + 1 windows per entry, and the read's side looks a 16-mer up every G = 6 (k = 19: 4)
// bases for the G windows that end in those G bases, with the bases behind X that a lane of the kernel does not have set to zero.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "vgmi_ctable.h"

static const uint64_t M54 = (1ULL << 56) - 1;      // the k-mer bits of an okmer word (k <= 28; 54 until round 6)
static const uint32_t OKF = 56;                     // ... first of its unitig at bit 56, k-mers behind it from bit 57
static uint32_t KK = 27, FF = 11, GG = 12, FT = 11, EX = 0;      // k, flank bases an entry stores, grid spacing, k - 16, k - 16 - FF (k = 28: 1)
static uint64_t MK = M54;                       // 2 k bits
static const uint32_t NONE = 0xFFFFFFFFu;

static int code(char c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

struct Model {
    std::vector<uint64_t> keys;                       // canonical k-mers
    std::unordered_map<uint64_t, uint32_t> index;     // canonical k-mer -> key index
    std::vector<uint32_t> id_of_key;
    std::vector<uint64_t> okmer;
    std::vector<CtBucket> cb;
    uint64_t n_buckets = 0;
    std::unordered_map<uint64_t, uint32_t> over;      // canonical k-mer -> id
    uint64_t n_entries = 0, n_unitigs = 0, n_moved = 0, n_over = 0;

    uint32_t find(uint64_t k) const
    {
        const uint64_t rc = vg_revcomp(k, KK);
        auto it = index.find(k < rc ? k : rc);
        return it == index.end() ? NONE : it->second;
    }

    // vgmi_ptable.hip pt_links_kernel / pt_mutual_kernel / pt_walk_kernel / pt_rest_kernel, serially
    void number()
    {
        const size_t n = keys.size();
        std::vector<uint32_t> link(2 * n, NONE), link2(2 * n, NONE), pos(n, NONE);
        for (size_t i = 0; i < n; ++i)
            for (uint32_t side = 0; side < 2; ++side) {
                const uint64_t K = keys[i];
                uint32_t found = NONE, cnt = 0, enter = 0;
                for (uint64_t b = 0; b < 4; ++b) {
                    const uint64_t N = side ? ((K << 2) | b) & MK : (K >> 2) | (b << (2 * KK - 2));
                    const uint64_t rc = vg_revcomp(N, KK);
                    const bool flipped = N > rc;
                    const uint32_t f = find(N);
                    if (f == NONE) continue;
                    ++cnt;
                    found = f;
                    enter = side ? (flipped ? 1u : 0u) : (flipped ? 0u : 1u);
                }
                link[2 * i + side] = (cnt == 1 && found != (uint32_t)i) ? (found | enter << 31) : NONE;
            }
        for (size_t g = 0; g < 2 * n; ++g) {
            const uint32_t l = link[g];
            if (l == NONE) continue;
            const uint64_t nb = l & 0x7FFFFFFFu, es = l >> 31;
            const uint32_t back = link[2 * nb + es];
            if (back != NONE && (back & 0x7FFFFFFFu) == (uint32_t)(g >> 1) && (back >> 31) == ((uint32_t)g & 1u)) link2[g] = l;
        }
        uint64_t cursor = 0;
        for (size_t g = 0; g < 2 * n; ++g) {
            if (link2[g] != NONE) continue;
            const uint32_t start = (uint32_t)(g >> 1), s0 = (uint32_t)g & 1u;
            uint32_t cur = start, out = s0 ^ 1u;
            uint64_t len = 1;
            for (;;) {
                const uint32_t l = link2[2ull * cur + out];
                if (l == NONE || len > n) break;
                cur = l & 0x7FFFFFFFu;
                out = (l >> 31) ^ 1u;
                ++len;
            }
            const bool own = start < cur || (start == cur && (s0 == 0u || link2[2ull * start] != NONE));
            if (!own || len > n) continue;
            const uint64_t base = cursor;
            cursor += len;
            cur = start;
            out = s0 ^ 1u;
            for (uint64_t q = 0; q < len; ++q) {
                pos[cur] = (uint32_t)(base + q) | out << 31;
                const uint32_t l = link2[2ull * cur + out];
                if (l == NONE) break;
                cur = l & 0x7FFFFFFFu;
                out = (l >> 31) ^ 1u;
            }
        }
        for (size_t i = 0; i < n; ++i)
            if (pos[i] == NONE) pos[i] = (uint32_t)(cursor++) | 1u << 31;
        // ct_okmer_kernel
        okmer.assign(n, 0);
        id_of_key.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t pk = pos[i], p = pk & 0x7FFFFFFFu, out = pk >> 31;
            const uint64_t Kw = out ? keys[i] : vg_revcomp(keys[i], KK);
            bool first = true;
            const uint32_t lb = link2[2 * i + (out ^ 1u)];
            if (lb != NONE) {
                const uint32_t pkn = pos[lb & 0x7FFFFFFFu];
                if ((pkn >> 31) == (lb >> 31) && (pkn & 0x7FFFFFFFu) + 1u == p) first = false;
            }
            uint32_t cur = (uint32_t)i, o = out, q = p, cnt = 0;
            while (cnt < 15u) {
                const uint32_t l = link2[2ull * cur + o];
                if (l == NONE) break;
                const uint32_t nx = l & 0x7FFFFFFFu, nxo = (l >> 31) ^ 1u, pkx = pos[nx];
                if ((pkx >> 31) != nxo || (pkx & 0x7FFFFFFFu) != q + 1u) break;
                cur = nx; o = nxo; ++q; ++cnt;
            }
            okmer[p] = Kw | (uint64_t)first << OKF | (uint64_t)cnt << (OKF + 1);
            id_of_key[i] = p;
            n_unitigs += first;
        }
    }

    // ct_insert_kernel, serially
    void build(double load)
    {
        const size_t n = keys.size();
        n_entries = n + (KK == 28 ? 10 : FT) * n_unitigs;      // (k = 28: a unitig's first and last occurrence have no window an entry can hold)
        n_buckets = (uint64_t)((double)n_entries / (4.0 * load)) + 1;
        CtBucket empty;
        memset(&empty, 0, sizeof empty);
        for (auto& x : empty.x) x = 0xFFFFFFFFu;
        cb.assign(n_buckets + CT_HOPS, empty);
        for (size_t p = 0; p < n; ++p)
            for (uint32_t o = 0; o <= FT; ++o) {
                const uint64_t ok = okmer[p];
                const bool first = (ok >> OKF) & 1;
                if (o != FT && !first) continue;
                const uint32_t rem = (uint32_t)(ok >> (OKF + 1)) & 15u;
                const uint32_t n_win = (o < rem ? o : rem) + 1u;
                CtEntry e[2];
                uint32_t keep = 0x1FFFu;      // ct_insert_kernel: even k, no window bit for a k-mer that is its own reverse complement
                if (!(KK & 1u))
                    for (uint32_t j = 0; j < n_win; ++j) {
                        const uint64_t kw = okmer[p + j] & M54;
                        if (kw == vg_revcomp(kw, KK)) keep &= ~(1u << (o - j));
                    }
                const int ne = ct_make_from_unitig(ok & M54, okmer[p + n_win - 1] & M54, o, n_win, (uint32_t)p, e, KK, keep);
                for (int q = 0; q < ne; ++q) {
                    const uint64_t b = ((uint64_t)ct_hash(e[q].d0) * n_buckets) >> 32;
                    bool placed = false;
                    for (uint32_t hop = 0; hop <= CT_HOPS && !placed; ++hop) {
                        CtBucket& B = cb[b + hop];
                        for (int s = 0; s < 4 && !placed; ++s)
                            if (B.x[s] == 0xFFFFFFFFu) {
                                B.x[s] = e[q].d0;
                                B.rest[s][0] = e[q].d1;
                                B.rest[s][1] |= e[q].d2;
                                B.rest[s][2] = e[q].d3;
                                placed = true;
                                n_moved += hop != 0;
                            }
                        if (!placed) B.rest[0][1] |= ct_mark(e[q].d0);
                    }
                    if (!placed)
                        for (uint32_t j = 0; j < n_win; ++j) {
                            const uint64_t kw = okmer[p + j] & M54, rc = vg_revcomp(kw, KK);
                            if (kw == rc) continue;
                            over[kw < rc ? kw : rc] = (uint32_t)(p + j);
                            ++n_over;
                        }
                }
            }
    }

    uint32_t over_find(uint64_t k) const
    {
        const uint64_t rc = vg_revcomp(k, KK);
        auto it = over.find(k < rc ? k : rc);
        return it == over.end() ? NONE : it->second;
    }

    // one grid position of count27c_kernel: counts[id] += 1 for every window that is a graph k-mer
    void probe(uint32_t x, uint32_t l, uint32_t r, uint32_t vw, std::vector<uint32_t>& counts) const
    {
        uint32_t cx, cl, cr, vs;
        ct_orient(x, l, r, vw, cx, cl, cr, vs, FF, EX);
        const uint64_t b0 = ((uint64_t)ct_hash(cx) * n_buckets) >> 32;
        uint32_t found = 0;
        for (uint32_t hop = 0;; ++hop) {
            const CtBucket& B = cb[b0 + hop];
            for (int q = 0; q < 4; ++q) {
                if (B.x[q] != cx) continue;
                const CtEntry e = {B.x[q], B.rest[q][0], B.rest[q][1], B.rest[q][2]};
                uint32_t h = ct_match(e, cx, cl, cr, FF, EX) & vs;
                if (h & found) { fprintf(stderr, "window matched twice\n"); exit(2); }
                found |= h;
                while (h) {
                    const uint32_t s = ct_ctz(h);
                    h &= h - 1;
                    ++counts[ct_id(e, s)];
                }
            }
            const bool more = B.x[3] != 0xFFFFFFFFu && (B.rest[0][1] & ct_mark(cx)) && (vs & ~found);
            if (more && hop == CT_HOPS) {
                uint32_t rest = vs & ~found;
                while (rest) {
                    const uint32_t s = ct_ctz(rest);
                    rest &= rest - 1;
                    const uint32_t id = over_find(ct_window_kmer(cx, cl, cr, s, KK));
                    if (id != NONE) ++counts[id];
                }
            }
            if (!more || hop == CT_HOPS) break;
        }
    }

    // ct_find (vgmi_xtable.h)
    uint32_t find_ct(uint64_t kmer) const
    {
        uint32_t cx, cl, cr, vs;
        ct_orient_kmer(kmer, cx, cl, cr, vs, KK);
        const uint64_t b0 = ((uint64_t)ct_hash(cx) * n_buckets) >> 32;
        for (uint32_t hop = 0; hop <= CT_HOPS; ++hop) {
            const CtBucket& B = cb[b0 + hop];
            for (int q = 0; q < 4; ++q) {
                if (B.x[q] != cx) continue;
                const CtEntry e = {B.x[q], B.rest[q][0], B.rest[q][1], B.rest[q][2]};
                const uint32_t h = ct_match(e, cx, cl, cr, FF, EX) & vs;
                if (h) return ct_id(e, ct_ctz(h));
            }
            if (B.x[3] == 0xFFFFFFFFu || !(B.rest[0][1] & ct_mark(cx))) return NONE;
        }
        return over_find(kmer);
    }
};

static void add_kmers(const std::string& s, std::vector<uint64_t>& out)
{
    uint64_t f = 0;
    int len = 0;
    for (char c : s) {
        const int b = code(c);
        if (b > 3) { len = 0; continue; }
        f = ((f << 2) | (uint64_t)b) & MK;
        if (++len >= (int)KK) {
            const uint64_t rc = vg_revcomp(f, KK);
            out.push_back(f < rc ? f : rc);
        }
    }
}

int main(int argc, char** argv)
{
    const uint64_t seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    const size_t G = argc > 2 ? strtoull(argv[2], 0, 10) : 20000;
    const size_t V = argc > 3 ? strtoull(argv[3], 0, 10) : 400;
    const size_t R = argc > 4 ? strtoull(argv[4], 0, 10) : 2000;
    const double load = (argc > 5 ? atoi(argv[5]) : 40) / 100.0;
    const size_t copies = argc > 6 ? strtoull(argv[6], 0, 10) : 0;
    if (argc > 7) KK = (uint32_t)atoi(argv[7]);
    if (KK < 19 || KK > 28) { fprintf(stderr, "k = 19 .. 28\n"); return 2; }
    FT = KK - 16;
    FF = ct_flank(KK);
    EX = ct_excess(KK);
    GG = KK == 27 ? 12 : KK <= 20 ? 4 : 6;      // (even k: the table's algebra under the windows-of-bases rule; the reference's run counter is the device pass's business)
    MK = (1ULL << (2 * KK)) - 1;
    std::mt19937_64 rng(seed);
    const char ACGT[] = "ACGT";
    std::string ref(G, 'A');
    for (auto& c : ref) c = ACGT[rng() & 3];
    if (copies) {      // a reference made of diverged copies of one element: 16-mers with many contexts (the overflow trail)
        const std::string unit = ref.substr(0, 300);
        for (size_t i = 0; i + 300 <= G; i += 300)
            for (size_t j = 0; j < 300; ++j) ref[i + j] = (rng() % 100 < 2) ? ACGT[rng() & 3] : unit[j];
        (void)copies;
    }
    // planted oddities: homopolymers, a 16-mer that is its own reverse complement inside a hairpin-free context, tandem repeats
    if (G > 4000) {
        ref.replace(500, 40, std::string(40, 'A'));
        ref.replace(700, 16, "ACGTACGTACGTACGT");           // palindromic 16-mer (its own reverse complement)
        ref.replace(900, 16, "AAAACCCCGGGGTTTT");           // another
        ref.replace(1100, 36, "ACACACACACACACACACACACACACACACACACAC");
        ref.replace(1300, 28, "GATTACAGATTACAGATTACAGATTACA");
    }
    // variants: SNPs, some of them close together, a few short indels; the key set = every k-mer within 26 bases of a variant on
    // either allele (what a variant node of the reference's graph holds in essence), alleles combined with the reference around
    std::vector<size_t> vpos;
    for (size_t i = 0; i < V; ++i) vpos.push_back(100 + rng() % (G - 200));
    for (size_t i = 0; i + 1 < V; i += 7) vpos[i + 1] = vpos[i] + 1 + rng() % 20;     // neighbours closer than k
    std::sort(vpos.begin(), vpos.end());
    vpos.erase(std::unique(vpos.begin(), vpos.end()), vpos.end());
    std::string hap = ref;
    std::vector<uint64_t> ks;
    for (size_t p : vpos) {
        char alt = ACGT[(code(ref[p]) + 1 + rng() % 3) & 3];
        hap[p] = alt;
        const size_t a = p >= KK - 1 ? p - (KK - 1) : 0, b = std::min(G, p + KK);
        std::string w = ref.substr(a, b - a);
        add_kmers(w, ks);
        w[p - a] = alt;
        add_kmers(w, ks);
        if (rng() % 10 == 0) {           // a deletion of 1..5 bases behind the site, as a third allele
            std::string d = ref.substr(a, p + 1 - a) + ref.substr(std::min(G, p + 1 + 1 + rng() % 5), KK - 1);
            add_kmers(d, ks);
        }
    }
    std::sort(ks.begin(), ks.end());
    ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
    Model m;
    m.keys = ks;
    for (size_t i = 0; i < ks.size(); ++i) m.index[ks[i]] = (uint32_t)i;
    m.number();
    m.build(load);

    // reads: both haplotypes, both strands, errors, N, lower case, ragged lengths; '\n'-joined stream as the kernels see it
    std::string stream;
    for (size_t i = 0; i < R; ++i) {
        const std::string& h = (rng() & 1) ? hap : ref;
        const size_t len = (rng() % 8 == 0) ? KK + rng() % 200 : 150;
        const size_t at = rng() % (G - std::min(len, G - 1));
        std::string rd = h.substr(at, len);
        if (rng() & 1) {
            std::reverse(rd.begin(), rd.end());
            for (auto& c : rd) c = ACGT[3 - code(c)];
        }
        for (auto& c : rd) {
            const uint64_t u = rng() % 400;
            if (u == 0) c = 'N';
            else if (u < 3) c = ACGT[rng() & 3];
            else if (u < 6) c = (char)(c | 0x20);
        }
        stream += rd;
        stream += '\n';
    }
    const size_t n = stream.size();
    std::vector<uint8_t> cd(n);
    for (size_t i = 0; i < n; ++i) cd[i] = (uint8_t)code(stream[i]);
    // brute force: every stream position whose last 27 bytes are bases
    std::vector<uint32_t> want(ks.size(), 0), got_by_id(ks.size(), 0);
    {
        uint64_t f = 0;
        int len = 0;
        for (size_t i = 0; i < n; ++i) {
            if (cd[i] > 3) { len = 0; continue; }
            f = ((f << 2) | cd[i]) & MK;
            if (++len >= (int)KK) {
                uint32_t k = m.find(f);
                if (f == vg_revcomp(f, KK)) k = NONE;      // (even k: never emitted by the reference, never counted here)
                if (k != NONE) ++want[k];
                const uint32_t id = m.find_ct(f);       // the generic kernels' tail lookup, on every k-mer
                if (id != (k == NONE ? NONE : m.id_of_key[k])) { fprintf(stderr, "ct_find differs at %zu\n", i); return 1; }
            }
        }
    }
    // the grid walk, as countc_body schedules it.  K = 27: X = the 16 bases ending at stream position 11 (mod 12), twelve windows.  K = 19 .. 22:
    // an X every G bases, G windows each.  K = 23 .. 26: three lookups per PAIR of lanes (24 bytes) -- the even lane's X (ends at 23 mod 24)
    // with NW = K - 15 windows, the odd lane's first X 12 - NW bases in front of its stretch (NW windows: the even lane's last ends and its
    // own first ones), its second for the rest.  Window w ends at e + w; `avail` = the bases behind X the lane has (the rest are zeros).
    struct Pos { uint32_t at, period, n_win, avail; };
    std::vector<Pos> sched;
    if (KK == 27) sched.push_back({11, 12, 12, 11});
    else if (KK == 28) {      // windows 1 .. 11 behind an X: the schedule of K = 26 with every X one base earlier
        sched.push_back({22, 24, 12, 11});
        sched.push_back({9, 24, 12, 11});
        sched.push_back({20, 24, 3, 3});
    } else if (KK <= 22) for (uint32_t j = 0; j < 12 / GG; ++j) sched.push_back({(GG * j + 11) % 12, 12, GG, std::min(FF, 12 - GG * j)});
    else {
        const uint32_t NW = FF + 1;
        sched.push_back({23, 24, NW, FF});
        sched.push_back({11 - (12 - NW), 24, NW, FF});
        sched.push_back({2 * NW - 1, 24, 24 - 2 * NW, std::min(FF, 24 - 2 * NW)});
    }
    for (size_t e = 15; e < n; ++e) {
        for (const Pos& ps : sched) {
            if (e % ps.period != ps.at) continue;
            uint32_t x = 0, l = 0, r = 0, vw = 0;
            bool okx = true;
            for (size_t j = e - 15; j <= e; ++j) { okx &= cd[j] < 4; x = (x << 2) | (cd[j] & 3u); }
            if (!okx) continue;
            for (int j = (int)FF; j >= 1; --j) {           // bases e - 15 - j: in front of X
                const long q = (long)e - 15 - j;
                l = (l << 2) | (q >= 0 ? cd[q] & 3u : 0u);
            }
            for (uint32_t j = 1; j <= FF; ++j) r = (r << 2) | (j <= ps.avail && e + j < n ? cd[e + j] & 3u : 0u);
            for (uint32_t w = 0; w < ps.n_win; ++w) {
                bool ok = e + w < n && e + w >= KK - 1;
                for (size_t j = 0; ok && j < KK; ++j) ok = cd[e + w - j] < 4;
                vw |= (uint32_t)ok << w;
            }
            vw &= ~((1u << EX) - 1u);      // (k = 28: no window ends at X's last base)
            if (vw) m.probe(x, l, r, vw, got_by_id);
        }
    }
    // the first stream positions (e < 15 never happens for e = 11 only when n is tiny) are covered: windows ending before 26 do not exist
    size_t bad = 0, hits = 0;
    for (size_t i = 0; i < ks.size(); ++i) {
        hits += want[i];
        if (got_by_id[m.id_of_key[i]] != want[i]) ++bad;
    }
    printf("{\"keys\": %zu, \"unitigs\": %llu, \"entries\": %llu, \"buckets\": %llu, \"moved\": %llu, \"over_kmers\": %llu, \"hits\": %zu, \"bad\": %zu}\n",
           ks.size(), (unsigned long long)m.n_unitigs, (unsigned long long)m.n_entries, (unsigned long long)m.n_buckets,
           (unsigned long long)m.n_moved, (unsigned long long)m.n_over, hits, bad);
    return bad ? 1 : 0;
}
