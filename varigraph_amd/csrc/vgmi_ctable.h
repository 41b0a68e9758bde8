// vgmi_ctable.h -- the CONTEXT TABLE of large graphs (k = 27): entry algebra shared by the device code (vgmi_ctable.hip,
// the generic kernels' tail lookups) and the host-side model the CPU suite runs (tests/native/ctable_model.cpp).
//
// Reference behaviour served: src/kmer.cpp:140-142 (exact membership of the canonical k-mer, one unordered_map::find per
// k-mer), src/fastq_kmer.cpp:128-139 (saturating count).
//
// Why.  The grid-16-mer table of round 2 (vgmi_xtable.hip) answers a candidate run -- the twelve windows of a read around one
// grid 16-mer X -- with one 128-byte line, but needs a filter word in front of it (one memory-side request per 12 bases: 12.6 of
// the 21.8 requests per read the kernel is bound by) and stores every k-mer twelve times.  The twelve windows are
// not independent: they are consecutive k-mers of one haplotype path, and so are the graph k-mers they may equal (the idea of
// the small graphs' path table, vgmi_ptable.hip).  So the table stores, per OCCURRENCE of a 16-mer X in a unitig of the key set,
// ONE 16-byte entry holding the occurrence's whole 38-base context:
//     d0  X, canonical (32 bits; 0xFFFFFFFF cannot be canonical: the slot is empty)
//     d1  L: the 11 bases in front of X (22 bits, the base next to X least significant) | mask bits 0..9 << 22
//     d2  R: the 11 bases behind X (22 bits, the base next to X most significant) | mask bits 10..11 << 22 | dir << 24
//         | marks in bits 26..31 (slot 0 of a bucket only): an entry that found this bucket full and went on to the next one set
//           bit 26 + ct_mark(X) -- a position goes on only if the bit of ITS X is set (six bits: five in six of the positions that
//           meet a full bucket with one displaced entry stop there)
//     d3  id0
// A bucket of four entries is laid out for the negative probe: the four X first (16 bytes: one load says whether anything in the
// bucket concerns this position), then the four (d1, d2, d3) triples -- fetched only for the slots whose X matched, from a line
// that is in the vector cache by then (CtBucket).
// Window s (0..11) of the context takes s bases of L, X, and 11 - s bases of R.  mask bit s says the unitig holds that window
// as a k-mer, and its counter is id0 - s (dir = 0) or id0 + s (dir = 1): counters are numbered along the unitigs, so the
// windows of one entry are neighbours.  Bases of L / R the unitig does not have are zero and no masked window uses them.
// A read position compares its own context base by base: the mismatch nearest to X on either side bounds the windows that
// are equal -- two find-first-bit instructions answer all twelve windows, exactly, in ONE lane -- so there is no filter: the
// bucket (four entries = one 64-byte memory-side request) is both the membership test and the answer, and a k-mer costs
// (L_unitig + 11) / L_unitig entries instead of twelve.
//
// Exactly one (entry, s) per (graph k-mer, offset of X in it): every k-mer of a unitig at position u holds X-occurrences
// u .. u + 11; the occurrence's entry lists the k-mers u - 11 .. u that contain it.  X is stored in its canonical
// orientation; an occurrence that reads the other way is stored reverse-complemented (L and R swap, s -> 11 - s, the ids run
// the other way); a 16-mer that is its own reverse complement is stored both ways, and no window can match both (that would
// put a k-mer and its reverse complement -- one key -- at two places of one unitig).
//
// Other odd k (19 .. 25; round 5).  The algebra does not depend on 27: with F = k - 16 flank bases on either side an entry holds the
// F + 1 windows of k bases around its X (window s: s bases of L, X, F - s of R), and every function below takes F (or k) next to the
// values -- 11 / 27 by default.  The fields keep their places (22 bits per flank, twelve mask bits), the upper ones stay zero.  What
// changes is on the read's side: a 16-mer every TWELVE bases no longer meets every window of k < 27 bases, so the count kernel looks
// one up every G bases, G = 6 (k = 21 .. 25) or 4 (k = 19), and asks each only for the G windows that end in its own G bases.
//
// k = 28 (round 6; the reference's upper bound, main.cpp:187).  A k-mer of 28 bases has 12 bases around a 16-mer, an entry has room for
// flanks of 11 -- so an entry of k = 28 keeps 11 + 11 flank bases and answers the ELEVEN windows that fit its 38-base context: window s
// (0 .. 10) takes s + 1 bases of L, X, and 11 - s bases of R.  The two windows of an occurrence it cannot hold (X at the very start or the
// very end of the k-mer) are windows 0 / 10 of the occurrences next to it, so every k-mer still has an entry for 11 of its 13 sixteen-mers
// and a read finds it through any of them.  In the functions below: f = bases of flank stored (ct_flank(k): 11 for k >= 27), ex = k - 16 - f
// (ct_excess(k): 1 for k = 28, else 0) -- window s takes s + ex bases of L and f - s of R, s = 0 .. f - ex.
#ifndef VGMI_CTABLE_H
#define VGMI_CTABLE_H

#include <stdint.h>

#include "vgmi_device.h"

#define CT_HOPS 7u                 // an entry sits in its home bucket or one of the CT_HOPS buckets behind it
#define CT_MARKS 0xFC000000u
#define CT_DIR (1u << 24)
#define CT_M22 0x3FFFFFu

VG_HD uint32_t ct_flank(uint32_t k) { return k - 16u > 11u ? 11u : k - 16u; }
VG_HD uint32_t ct_excess(uint32_t k) { return k - 16u - ct_flank(k); }

struct CtEntry {
    uint32_t d0, d1, d2, d3;
};
struct CtBucket {                  // 64 bytes
    uint32_t x[4];                 // d0 of the four entries, 0xFFFFFFFF = empty; slots fill in order, so x[3] set = bucket full
    uint32_t rest[4][3];           // their (d1, d2, d3); rest[0][1] carries the marks
};

// bijection of 32 bits (bucket = (ct_hash(X) * n_buckets) >> 32)
VG_HD uint32_t ct_hash(uint32_t cx)
{
    uint32_t h = cx * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}

// which of the six marks of a full bucket an entry with this X sets when it goes on (the low half of the hash; the bucket comes from the top)
VG_HD uint32_t ct_mark(uint32_t cx) { return 1u << (26u + (((ct_hash(cx) & 0xFFFFu) * 6u) >> 16)); }

VG_HD uint32_t ct_rc11(uint32_t x, uint32_t f = 11u) { return vg_revcomp16(x) >> (32u - 2u * f); }      // f bases in the low 2 f bits
VG_HD uint32_t ct_rev12(uint32_t m, uint32_t f = 11u)                                                    // bit s -> bit f - s
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(m) >> (31u - f);
#else
    uint32_t r = 0;
    for (uint32_t s = 0; s <= f; ++s) r |= ((m >> s) & 1u) << (f - s);
    return r;
#endif
}
VG_HD uint32_t ct_ctz(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_ctz(x);
#else
    uint32_t n = 0;
    while (!((x >> n) & 1u)) ++n;
    return n;
#endif
}
VG_HD uint32_t ct_clz(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_clz(x);
#else
    uint32_t n = 0;
    while (!((x << n) & 0x80000000u)) ++n;
    return n;
#endif
}

// A read position's context in the table's orientation.  x, l, r as read (l: the 11 bases in front of x, the one next to x
// least significant; r: the 11 behind it, the one next to x most significant); vw bit w = the window that ends w bases
// behind x's last base is made of bases only (the scan's numbering).  Out: canonical x, its flanks, vs bit s = window s valid.
// (ex > 0: the windows that end fewer than ex bases behind x do not exist; bit w of vw for them is ignored)
VG_HD void ct_orient(uint32_t x, uint32_t l, uint32_t r, uint32_t vw, uint32_t& cx, uint32_t& cl, uint32_t& cr, uint32_t& vs, uint32_t f = 11u, uint32_t ex = 0u)
{
    const uint32_t rc = vg_revcomp16(x);
    const bool as_is = x <= rc;
    cx = as_is ? x : rc;
    cl = as_is ? l : ct_rc11(r, f);
    cr = as_is ? r : ct_rc11(l, f);
    vs = as_is ? ct_rev12(vw, f) & ((2u << (f - ex)) - 1u) : vw >> ex;     // window w takes w bases behind x: s = f - w as read, s = w - ex reversed
}

// windows of the context (cx, cl, cr) that equal k-mers of entry e: bit s
VG_HD uint32_t ct_match(const CtEntry& e, uint32_t cx, uint32_t cl, uint32_t cr, uint32_t f = 11u, uint32_t ex = 0u)
{
    if (e.d0 != cx) return 0u;
    const uint32_t mf = (1u << (2u * f)) - 1u;
    const uint32_t tl = ((e.d1 ^ cl) & mf) | (1u << (2u * f));
    const uint32_t nl = ct_ctz(tl) >> 1;                           // bases of L equal next to X: 0..f
    const uint32_t tr = (((e.d2 ^ cr) & mf) << (32u - 2u * f)) | (1u << (31u - 2u * f));
    const uint32_t nr = ct_clz(tr) >> 1;                           // bases of R equal next to X: 0..f
    const uint32_t range = (((2u << nl) >> ex) - 1u) & ~((1u << (f - nr)) - 1u);       // f - nr <= s <= nl - ex
    const uint32_t emask = (e.d1 >> 22) | ((e.d2 >> 12) & 0xC00u);
    return range & emask;
}
VG_HD uint32_t ct_id(const CtEntry& e, uint32_t s) { return (e.d2 & CT_DIR) ? e.d3 + s : e.d3 - s; }      // (any f)

// The entry of one occurrence, from the unitig as the numbering walks it: xu the 16-mer, lu / ru its flanks (missing bases
// zero), mask bit s = the unitig holds window s, whose counter is id0 - s.  Returns 1 entry, or 2 when xu is its own
// reverse complement (both readings).
VG_HD int ct_make(uint32_t xu, uint32_t lu, uint32_t ru, uint32_t mask, uint32_t id0, CtEntry out[2], uint32_t f = 11u, uint32_t ex = 0u)
{
    const uint32_t rc = vg_revcomp16(xu), mf = (1u << (2u * f)) - 1u;
    int n = 0;
    if (xu <= rc) {
        out[n].d0 = xu;
        out[n].d1 = (lu & mf) | (mask & 0x3FFu) << 22;
        out[n].d2 = (ru & mf) | ((mask >> 10) & 3u) << 22;
        out[n].d3 = id0;
        ++n;
    }
    if (xu >= rc) {
        const uint32_t m = ct_rev12(mask, f - ex);      // window s reads as window f - ex - s the other way
        out[n].d0 = rc;
        out[n].d1 = ct_rc11(ru & mf, f) | (m & 0x3FFu) << 22;
        out[n].d2 = ct_rc11(lu & mf, f) | ((m >> 10) & 3u) << 22 | CT_DIR;
        out[n].d3 = id0 - (f - ex);
        ++n;
    }
    return n;
}

// The occurrence led by the k-mer at unitig position u (kf, as walked) with X starting o bases into it (0..f): the k-mers
// u .. u + n_win - 1 hold it (n_win = min(o, k-mers behind kf in the unitig) + 1; kl = the last of them, id p of kf).
// keep: window bits that may be set (even k: not those of k-mers that are their own reverse complement, which the reference never emits)
VG_HD int ct_make_from_unitig(uint64_t kf, uint64_t kl, uint32_t o, uint32_t n_win, uint32_t p, CtEntry out[2], uint32_t k = 27u, uint32_t keep = 0x1FFFu)
{
    const uint32_t ft = k - 16u, f = ct_flank(k), ex = ft - f;      // keep: bits by X's offset in the k-mer, 0 .. ft
    const uint32_t xu = (uint32_t)(kf >> (2u * (ft - o)));
    const uint32_t lu = o ? (uint32_t)(kf >> (2u * (k - o))) : 0u;                  // the o bases in front of X (ct_make keeps the f next to X)
    const uint32_t n_r = n_win + ft - 1u - o;                                       // bases behind X the last k-mer reaches: 0..ft
    const uint32_t ru = n_r ? (((uint32_t)kl & ((1u << (2u * n_r)) - 1u)) << (2u * (ft - n_r))) >> (2u * ex) : 0u;      // ... the f next to X
    const uint32_t by_off = (((1u << n_win) - 1u) << (o + 1u - n_win)) & keep;      // X's offset in the k-mers that hold the occurrence: o - n_win + 1 .. o
    const uint32_t mask = (by_off >> ex) & ((2u << (f - ex)) - 1u);                 // window s = offset - ex; offsets below ex and above f have no window here
    if (!mask) return 0;
    return ct_make(xu, lu, ru, mask, p + o - ex, out, f, ex);
}

// A single k-mer as a one-window context: X = its last 16 bases, window s = f as read
VG_HD void ct_orient_kmer(uint64_t kmer, uint32_t& cx, uint32_t& cl, uint32_t& cr, uint32_t& vs, uint32_t k = 27u)
{
    const uint32_t f = ct_flank(k), ex = ct_excess(k);      // (ex = 1: X = the 16 bases in front of the k-mer's last base)
    ct_orient((uint32_t)(kmer >> (2u * ex)), (uint32_t)(kmer >> (32u + 2u * ex)) & ((1u << (2u * f)) - 1u),
              ex ? ((uint32_t)kmer & ((1u << (2u * ex)) - 1u)) << (2u * (f - ex)) : 0u, 1u << ex, cx, cl, cr, vs, f, ex);
}

// the k-mer (in the table's orientation) of window s of a context: for the exact overflow table
VG_HD uint64_t ct_window_kmer(uint32_t cx, uint32_t cl, uint32_t cr, uint32_t s, uint32_t k = 27u)
{
    const uint32_t f = ct_flank(k), ex = ct_excess(k);
    const uint64_t left = (uint64_t)(cl & ((1u << (2u * (s + ex))) - 1u));                 // the s + ex bases of L next to X
    const uint64_t right = (uint64_t)(cr & ((1u << (2u * f)) - 1u)) >> (2u * s);          // the first f - s bases of R
    return left << (2u * (16u + f - s)) | (uint64_t)cx << (2u * (f - s)) | right;
}

#endif
