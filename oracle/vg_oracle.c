/*
 * vg_oracle.c -- CPU restatement of varigraph's per-sample genotyping hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see vg_oracle.h).  Parity status: PINNED against the real
 * reference build in oracle/_ref/ and the golden vectors in tests/golden/.
 * Citations are file:line under /root/reference/.
 */
#include "vg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ A1 */
/* include/seq_nt4_table.hpp:5-22: bytes 0..3 map to themselves, A/a C/c G/g T/t U/u
 * to 0 1 2 3 3, everything else to 4. */
const uint8_t vgo_nt4_table[256] = {
    [0 ... 255] = 4,
    [0] = 0,   [1] = 1,   [2] = 2,   [3] = 3,
    ['A'] = 0, ['C'] = 1, ['G'] = 2, ['T'] = 3, ['U'] = 3,
    ['a'] = 0, ['c'] = 1, ['g'] = 2, ['t'] = 3, ['u'] = 3,
};

/* ------------------------------------------------------------------ A2 */
/* include/hash64.hpp:5-14 */
uint64_t vgo_hash64(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

/* ------------------------------------------------------------------ A3 */
/* State machine of src/kmer.cpp:126-146 (identical in the bf/construct/genotype
 * copies): registers start at 0 per call; a valid base shifts both registers, then a
 * palindrome (`fwd == rc`) is skipped WITHOUT ++l (:134), else ++l and emit when l>=k
 * (:135-138; kmer_span is always k there, and k<=28<256); an invalid base only zeroes
 * l (:145) -- the registers keep their bits. */
typedef void (*vgo_sink)(void *ctx, uint64_t key);

static int64_t vgo_sketch_core(const char *s, size_t len, uint32_t k, vgo_sink sink, void *ctx)
{
    if (!(len > 0 && (k > 0 && k <= 28))) return -1; /* src/kmer.cpp:124 assert */
    /* `unsigned int len = str.length()` (src/kmer.cpp:122) */
    unsigned int ulen = (unsigned int)len;
    uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1, kmer[2] = {0, 0};
    int64_t n = 0;
    int l = 0;
    int kmer_span = 0;
    for (unsigned int i = 0; i < ulen; ++i) {
        int c = vgo_nt4_table[(uint8_t)s[i]];
        if (c < 4) {
            kmer_span = l + 1 < (int)k ? l + 1 : (int)k;
            kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
            if (kmer[0] == kmer[1]) continue;
            int z = kmer[0] < kmer[1] ? 0 : 1;
            ++l;
            if (l >= (int)k && kmer_span < 256) {
                sink(ctx, vgo_hash64(kmer[z], mask) << 8 | (uint64_t)kmer_span);
                ++n;
            }
        } else {
            l = 0;
            kmer_span = 0;
        }
    }
    return n;
}

struct out_ctx { uint64_t *out; size_t n; };
static void sink_out(void *c, uint64_t key)
{
    struct out_ctx *o = (struct out_ctx *)c;
    o->out[o->n++] = key;
}

size_t vgo_sketch(const char *s, size_t len, uint32_t k, uint64_t *out)
{
    struct out_ctx o = {out, 0};
    int64_t r = vgo_sketch_core(s, len, k, sink_out, &o);
    return r < 0 ? (size_t)-1 : (size_t)r;
}

/* ------------------------------------------------------------------ A6 / A5 */
/* The reference table is std::unordered_map<uint64_t,kmerCovFreBitVec>
 * (include/construct_index.hpp:140); only membership and the u8 counter matter here,
 * and the final state does not depend on iteration order. */
struct vgo_table {
    size_t n, cap;   /* cap: power of two */
    uint64_t *slot;  /* key or EMPTY */
    uint32_t *idx;   /* slot -> input index */
    uint8_t *c;      /* per input index */
};
#define VGO_EMPTY UINT64_MAX

static inline uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    return x;
}

vgo_table *vgo_table_new(const uint64_t *keys, size_t n)
{
    vgo_table *t = (vgo_table *)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->n = n;
    t->cap = 16;
    while (t->cap < 2 * n + 2) t->cap <<= 1;
    t->slot = (uint64_t *)malloc(t->cap * sizeof(uint64_t));
    t->idx = (uint32_t *)malloc(t->cap * sizeof(uint32_t));
    t->c = (uint8_t *)calloc(n ? n : 1, 1);
    if (!t->slot || !t->idx || !t->c) { vgo_table_free(t); return NULL; }
    for (size_t i = 0; i < t->cap; ++i) t->slot[i] = VGO_EMPTY;
    for (size_t i = 0; i < n; ++i) {
        size_t p = mix(keys[i]) & (t->cap - 1);
        while (t->slot[p] != VGO_EMPTY) {
            if (t->slot[p] == keys[i]) { vgo_table_free(t); return NULL; }
            p = (p + 1) & (t->cap - 1);
        }
        t->slot[p] = keys[i];
        t->idx[p] = (uint32_t)i;
    }
    return t;
}

void vgo_table_free(vgo_table *t)
{
    if (!t) return;
    free(t->slot); free(t->idx); free(t->c); free(t);
}

size_t vgo_table_size(const vgo_table *t) { return t->n; }

int64_t vgo_table_find(const vgo_table *t, uint64_t key)
{
    size_t p = mix(key) & (t->cap - 1);
    while (t->slot[p] != VGO_EMPTY) {
        if (t->slot[p] == key) return (int64_t)t->idx[p];
        p = (p + 1) & (t->cap - 1);
    }
    return -1;
}

static void sink_count(void *c, uint64_t key)
{
    vgo_table *t = (vgo_table *)c;
    int64_t i = vgo_table_find(t, key);   /* src/kmer.cpp:140 */
    if (i >= 0 && t->c[i] < UINT8_MAX)    /* src/fastq_kmer.cpp:133-137 */
        t->c[i]++;
}

struct hit_ctx { vgo_table *t; int64_t hits; };
static void sink_count_hits(void *c, uint64_t key)
{
    struct hit_ctx *h = (struct hit_ctx *)c;
    int64_t i = vgo_table_find(h->t, key);
    if (i >= 0) {
        h->hits++;
        if (h->t->c[i] < UINT8_MAX) h->t->c[i]++;
    }
}

int64_t vgo_count_read(vgo_table *t, const char *s, size_t len, uint32_t k)
{
    struct hit_ctx h = {t, 0};
    if (vgo_sketch_core(s, len, k, sink_count_hits, &h) < 0) return -1;
    return h.hits;
}

int64_t vgo_count_block(vgo_table *t, const char *block, size_t n_bytes, uint32_t k,
                        uint64_t *read_base)
{
    int64_t hits = 0;
    size_t p = 0;
    (void)sink_count;
    while (p < n_bytes) {
        const char *nl = (const char *)memchr(block + p, '\n', n_bytes - p);
        size_t len = nl ? (size_t)(nl - (block + p)) : n_bytes - p;
        int64_t h = vgo_count_read(t, block + p, len, k);
        if (h < 0) return -1;
        hits += h;
        if (read_base) *read_base += len; /* src/fastq_kmer.cpp:105 */
        p += len + 1;
    }
    return hits;
}

void vgo_table_counts(const vgo_table *t, uint8_t *c_out) { memcpy(c_out, t->c, t->n); }
void vgo_table_reset(vgo_table *t) { memset(t->c, 0, t->n); }

/* ------------------------------------------------------------------ B1 */
/* src/counting_bloom_filter.cpp:70-72: ceil((n*log(p)) / log(1.0/pow(2.0, log(2.0)))) */
uint64_t vgo_bloom_size(uint64_t n, double p)
{
    return (uint64_t)ceil(((double)n * log(p)) / log(1.0 / pow(2.0, log(2.0))));
}
/* src/counting_bloom_filter.cpp:75-77: round(m*log(2.0)/n) */
uint32_t vgo_bloom_num_hashes(uint64_t n, uint64_t m)
{
    return (uint32_t)round((double)m * log(2.0) / (double)n);
}

/* ------------------------------------------------------------------ B2 */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
/* MurmurHash3_x64_128 (src/MurmurHash3.cpp:255-332) for len == 8: nblocks = 0, tail
 * `case 8` (:305-313) folds the 8 little-endian key bytes into k1, finalisation
 * (:319-331); _murmur_hash returns hashValue[0]+hashValue[1]
 * (src/counting_bloom_filter.cpp:90-98) and takes `unsigned int seed`. */
uint64_t vgo_murmur_sum(uint64_t key, uint64_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint32_t seed32 = (uint32_t)seed;
    uint64_t h1 = seed32, h2 = seed32;
    uint64_t k1 = key;
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 ^= 8; h2 ^= 8;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    return h1 + h2;
}

/* ------------------------------------------------------------------ B3 */
void vgo_bloom_add(uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh, uint64_t key)
{
    for (uint32_t i = 0; i < nh; ++i) {
        uint64_t pos = vgo_murmur_sum(key, seeds[i]) % m;
        if (filter[pos] < 255) filter[pos]++;
    }
}

struct bf_ctx { uint8_t *filter; uint64_t m; const uint64_t *seeds; uint32_t nh; };
static void sink_bf(void *c, uint64_t key)
{
    struct bf_ctx *b = (struct bf_ctx *)c;
    vgo_bloom_add(b->filter, b->m, b->seeds, b->nh, key);
}

int64_t vgo_bloom_add_seq(uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh,
                          const char *s, size_t len, uint32_t k)
{
    struct bf_ctx b = {filter, m, seeds, nh};
    return vgo_sketch_core(s, len, k, sink_bf, &b);
}

/* ------------------------------------------------------------------ B4 */
uint8_t vgo_bloom_count(const uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh,
                        uint64_t key)
{
    uint8_t min_freq = UINT8_MAX;
    for (uint32_t i = 0; i < nh; ++i) {
        uint64_t pos = vgo_murmur_sum(key, seeds[i]) % m;
        if (filter[pos] < min_freq) min_freq = filter[pos];
    }
    return min_freq;
}

int vgo_bloom_find(const uint8_t *filter, uint64_t m, const uint64_t *seeds, uint32_t nh,
                   uint64_t key)
{
    for (uint32_t i = 0; i < nh; ++i)
        if (filter[vgo_murmur_sum(key, seeds[i]) % m] == 0) return 0;
    return 1;
}

/* ------------------------------------------------------------------ A8 */
void vgo_hom_hist(const uint8_t *c, const uint8_t *f, const int8_t *bitvec, size_t bitlen,
                  size_t n, uint32_t hap_num, uint32_t vcf_ploidy, uint64_t hist[256])
{
    memset(hist, 0, 256 * sizeof(uint64_t));
    for (size_t r = 0; r < n; ++r) {
        if (c[r] == 0 || f[r] > 1) continue;               /* src/varigraph.cpp:263-265 */
        const int8_t *bv = bitvec + r * bitlen;
        int total_hom = 0;
        uint32_t index = 0, sample_count = 0;
        for (uint32_t i = 1; i < hap_num; ++i) {           /* :271 */
            index++;
            if ((bv[i >> 3] >> (i & 7)) & 1) sample_count++; /* :275, make_QRmap construct_index.cpp:484-489 */
            if (index == vcf_ploidy) {                     /* :279 */
                index = 0;
                if (sample_count == vcf_ploidy) { total_hom++; break; }
                sample_count = 0;
            }
        }
        if (total_hom > 0) hist[c[r]]++;                   /* :290-293 */
    }
}

int vgo_hom_peak(const uint64_t hist[256], float read_depth, uint8_t *max_cov, uint8_t *hom_cov)
{
    /* kmerCovFreMap is a std::map holding only the coverages that occur, so the
     * neighbour tests below run over the compacted list (src/varigraph.cpp:310-325). */
    uint8_t cov[256];
    uint64_t fre[256];
    int nb = 0;
    int index = -1, max_index = -1;
    uint8_t maxc = 0, homc = 0;
    uint64_t maxf = 0;
    for (int v = 0; v < 256; ++v) {
        if (hist[v] == 0) continue;
        cov[nb] = (uint8_t)v; fre[nb] = hist[v]; nb++;
        index++;
        if (v > 1 && hist[v] >= maxf && v < UINT8_MAX) {   /* :321 */
            max_index = index; maxc = (uint8_t)v; maxf = hist[v]; homc = (uint8_t)v;
        }
    }
    if (max_index == -1) return -1;                         /* :330-334 */
    for (size_t i = (size_t)max_index + 1; i < (size_t)nb - 1; i++) { /* :337 */
        if ((float)cov[i] > read_depth) break;              /* :338 */
        if (fre[i] >= fre[i - 1] && fre[i] >= fre[i + 1]) homc = cov[i];
    }
    *max_cov = maxc; *hom_cov = homc;
    return 0;
}

float vgo_read_depth(uint64_t read_base, uint64_t genome_size)
{
    return read_base / (float)genome_size;                  /* src/varigraph.cpp:198 */
}

uint8_t vgo_use_depth_cov(float read_depth)
{
    uint8_t hom = read_depth * 0.8;                         /* src/varigraph.cpp:230-232 */
    return hom;
}

float vgo_hap_kmer_cov(uint8_t hom_cov, uint32_t sample_ploidy, float read_depth)
{
    return (hom_cov > 0 && sample_ploidy > 0)               /* src/varigraph.cpp:360-362 */
               ? (float)hom_cov / (float)sample_ploidy
               : read_depth / (float)sample_ploidy;
}
