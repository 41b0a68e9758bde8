"""The HMM's forward / backward recursion on the device (vgmi_hmm.hip, csrc/vg_x80.h) against the same recursion in x87
`long double` arithmetic on the host (numpy.longdouble: the C type, element by element, in the reference's order of
operations, src/genotype.cpp:1170-1380).  Emission scores down to 1e-4900 (gradual underflow), chains that restart, zero
totals, one to four haplotypes per genotype."""
import itertools

import numpy as np
import pytest

from varigraph_amd import vgmi

pytestmark = pytest.mark.gpu
LD = np.longdouble


def _host_chain(keep, obs_rows, restart, pows, uniform, ploidy):
    n = keep.shape[0]
    out = np.zeros((len(obs_rows), n), dtype=LD)
    prev = None
    for s, o in enumerate(obs_rows):
        if restart[s] or prev is None:
            r = LD(0) + o
        else:
            pk, pc = pows[s, 0], pows[s, 1]
            step = np.zeros((n, ploidy + 1), dtype=LD)
            for k in range(ploidy + 1):
                step[:, k] = (prev * pk[k]) * pc[ploidy - k]
            r = np.zeros(n, dtype=LD)
            for p in range(n):                       # the previous entries in their order; all genotypes at once
                r = r + step[p, keep[:, p]] * o
        total = LD(0)
        for g in range(n):
            total = total + r[g]
        out[s] = r / total if total > 0 else uniform
        prev = out[s]
    return out


@pytest.mark.parametrize("ploidy,n_hap,waves", [(2, 15, 4), (2, 15, 2), (2, 5, 4), (1, 9, 2), (3, 6, 4), (4, 5, 2)])
def test_recursion_equals_x87(ploidy, n_hap, waves, monkeypatch):
    """waves: the kernel with four wavefronts per chain (what a launch of few chains gets) or two (a launch of many)."""
    monkeypatch.setenv("VGMI_HMM_WAVES", str(waves))
    assert np.finfo(LD).nmant == 63, "numpy.longdouble is not the x87 format here"
    rng = np.random.default_rng(ploidy * 100 + n_hap)
    genotypes = list(itertools.combinations_with_replacement(range(n_hap), ploidy))[:128]
    n = len(genotypes)

    def shared(a, b):
        a, b, k = list(a), list(b), 0
        for x in a:
            if x in b:
                b.remove(x)
                k += 1
        return k
    n_windows = 3
    keep = np.zeros((n_windows, n, n), dtype=np.uint8)
    for w in range(n_windows):
        perm = rng.permutation(n)
        for i in range(n):
            for j in range(n):
                keep[w, i, j] = shared(genotypes[perm[i]], genotypes[perm[j]])
    n_rows = 40
    # emission scores: products of many small probabilities -- a few plausible genotypes, the rest far down, some beyond the
    # normal range, some exactly zero
    expo = rng.choice([0, -20, -300, -2000, -4800, -4940, -4960], size=(n_rows, n), p=[.15, .2, .25, .2, .1, .05, .05])
    obs = (rng.random((n_rows, n)).astype(LD) + LD(0.01)) * np.power(LD(10), expo.astype(LD))
    obs[rng.random((n_rows, n)) < 0.02] = 0
    obs[7] = 0                                           # a node whose scores are all zero: the uniform fallback
    steps_row, steps_restart, pows, chains = [], [], [], []
    for w in range(n_windows):
        for direction in (1, -1):
            rows = list(range(n_rows))[::direction][w:]
            first = len(steps_row)
            for i, r in enumerate(rows):
                steps_row.append(r)
                steps_restart.append(1 if i == 0 or (i % 13 == 5) else 0)
                d = LD(rng.integers(1, 50_000))
                recomb = (LD(1) - np.exp(-d / LD(30))) * (LD(1) / LD(30))
                no_recomb = np.exp(-d / LD(30)) + recomb
                pows.append([[no_recomb ** LD(k) for k in range(ploidy + 1)], [recomb ** LD(k) for k in range(ploidy + 1)]])
            chains.append((first, len(rows), w))
    pows = np.array(pows, dtype=LD)
    uniform = LD(1) / LD(n)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        got = ctx.hmm_recursion(keep, obs, steps_row, steps_restart, pows, uniform, chains, ploidy)
    finally:
        ctx.close()
    for first, count, w in chains:
        want = _host_chain(keep[w], [obs[r] for r in steps_row[first:first + count]], steps_restart[first:first + count],
                           pows[first:first + count], uniform, ploidy)
        g = got[first:first + count]
        assert np.array_equal(g, want), (ploidy, w, int(np.argmax((g != want).any(axis=1))))
    assert (got > 0).any() and (got == 0).any()


def test_posterior_on_the_device_equals_x87():
    """vgmi_hmm_calls: per node the sum of a * b, the posteriors, their sums per genotype string in entry order, the first
    maximum in string order and the first entry of that string with the largest posterior -- as posterior() of the host
    (src/genotype.cpp:1387-1522) computes them in long double."""
    rng = np.random.default_rng(77)
    ploidy, n_hap = 2, 12
    genotypes = list(itertools.combinations_with_replacement(range(n_hap), ploidy))
    n = len(genotypes)
    keep = np.zeros((1, n, n), dtype=np.uint8)
    for i, a in enumerate(genotypes):
        for j, b in enumerate(genotypes):
            keep[0, i, j] = len(set(a) & set(b)) if a[0] != a[1] or b[0] != b[1] else (2 if a == b else (1 if a[0] in b else 0))
    n_rows = 30
    expo = rng.choice([0, -10, -200, -3000, -4900], size=(n_rows, n), p=[.2, .3, .3, .15, .05])
    obs = (rng.random((n_rows, n)).astype(LD) + LD(0.01)) * np.power(LD(10), expo.astype(LD))
    obs[11] = 0
    row = list(range(n_rows)) + list(range(n_rows))[::-1]
    restart = [1] + [0] * (n_rows - 1) + [1] + [0] * (n_rows - 1)
    pows = np.zeros((2 * n_rows, 2, ploidy + 1), dtype=LD)
    for s in range(2 * n_rows):
        d = LD(rng.integers(1, 20_000))
        recomb = (LD(1) - np.exp(-d / LD(24))) * (LD(1) / LD(24))
        no_recomb = np.exp(-d / LD(24)) + recomb
        pows[s, 0] = [no_recomb ** LD(k) for k in range(ploidy + 1)]
        pows[s, 1] = [recomb ** LD(k) for k in range(ploidy + 1)]
    chains = [(0, n_rows, 0), (n_rows, n_rows, 0)]
    # genotype strings: a few alleles per node, ids by first appearance, order by text
    gid = np.zeros((n_rows, n), dtype=np.uint8)
    order = np.full((n_rows, n), 0xFF, dtype=np.uint8)
    for r in range(n_rows):
        alleles = rng.integers(0, 12 if r % 5 == 0 else 3, size=n_hap)
        texts = {}
        for g, (a, b) in enumerate(genotypes):
            t = "/".join(sorted([str(alleles[a]), str(alleles[b])]))
            gid[r, g] = texts.setdefault(t, len(texts))
        ranked = sorted(texts, key=lambda t: t)
        order[r, :len(ranked)] = [texts[t] for t in ranked]
    fwd = np.arange(n_rows, dtype=np.uint64)
    bwd = np.array([n_rows + (n_rows - 1 - j) for j in range(n_rows)], dtype=np.uint64)
    uniform = LD(1) / LD(n)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        prob, winner, ab = ctx.hmm_calls(keep, obs, row, restart, pows, uniform, chains, ploidy, gid, order, fwd, bwd)
    finally:
        ctx.close()
    n_called = 0
    for r in range(n_rows):
        a, b = ab[fwd[r]], ab[bwd[r]]
        den = LD(0)
        for g in range(n):
            den = den + a[g] * b[g]
        if den == 0:
            assert winner[r] == 0xFFFFFFFF
            continue
        post = (a * b) / den
        sums = {}
        for g in range(n):
            sums[gid[r, g]] = sums.get(gid[r, g], LD(0)) + post[g]
        best, best_id = LD(-1), None
        for k in order[r]:
            if k == 0xFF:
                break
            if sums[k] > best:
                best, best_id = sums[k], k
        mx, win = LD(0), 0xFFFFFFFF
        for g in range(n):
            if gid[r, g] == best_id and mx < post[g]:
                mx, win = post[g], g
        assert winner[r] == win and prob[r] == best, r
        n_called += win != 0xFFFFFFFF
    assert n_called > 20


def test_parts_side_by_side_equal_the_whole_call():
    """vgmi_hmm_calls_part on three parts of a run's arrays, from three threads at once on one context, fills prob / winner
    exactly as vgmi_hmm_calls on the whole arrays does (global row / step numbers, rows a window leaves unused included)."""
    import threading
    rng = np.random.default_rng(5)
    ploidy, n_hap = 2, 15
    genotypes = list(itertools.combinations_with_replacement(range(n_hap), ploidy))
    n = len(genotypes)
    n_windows, room = 6, 25
    keep = rng.integers(0, ploidy + 1, size=(n_windows, n, n), dtype=np.uint8)
    n_rows = n_windows * room
    expo = rng.choice([0, -10, -200, -3000, -4900], size=(n_rows, n), p=[.2, .3, .3, .15, .05])
    obs = (rng.random((n_rows, n)).astype(LD) + LD(0.01)) * np.power(LD(10), expo.astype(LD))
    row = np.zeros(2 * n_rows, dtype=np.uint32)
    restart = np.zeros(2 * n_rows, dtype=np.uint8)
    fwd = np.zeros(n_rows, dtype=np.uint64)
    bwd = np.zeros(n_rows, dtype=np.uint64)
    chains = []
    for w in range(n_windows):
        row0, step0, used = w * room, 2 * w * room, room - (w % 3)          # some windows leave rows unused
        row[step0:step0 + 2 * room] = row0
        fwd[row0:row0 + room] = step0
        bwd[row0:row0 + room] = step0
        row[step0:step0 + used] = np.arange(row0, row0 + used)
        row[step0 + used:step0 + 2 * used] = np.arange(row0, row0 + used)[::-1]
        restart[step0] = restart[step0 + used] = 1
        fwd[row0:row0 + used] = step0 + np.arange(used)
        bwd[row0:row0 + used] = step0 + used + (used - 1 - np.arange(used))
        chains.append((step0, used, w))
        chains.append((step0 + used, used, w))
    pows = np.zeros((2 * n_rows, 2, ploidy + 1), dtype=LD)
    for s in range(2 * n_rows):
        d = LD(rng.integers(1, 20_000))
        recomb = (LD(1) - np.exp(-d / LD(24))) * (LD(1) / LD(24))
        no_recomb = np.exp(-d / LD(24)) + recomb
        pows[s, 0] = [no_recomb ** LD(k) for k in range(ploidy + 1)]
        pows[s, 1] = [recomb ** LD(k) for k in range(ploidy + 1)]
    gid = rng.integers(0, 4, size=(n_rows, n), dtype=np.uint8)
    order = np.full((n_rows, n), 0xFF, dtype=np.uint8)
    order[:, :4] = [2, 0, 3, 1]
    uniform = LD(1) / LD(n)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        prob, winner, _ = ctx.hmm_calls(keep, obs, row, restart, pows, uniform, chains, ploidy, gid, order, fwd, bwd)
        prob2 = np.full(n_rows, LD(-1), dtype=LD)
        winner2 = np.full(n_rows, 12345, dtype=np.uint32)
        errors = []

        def part(w0, w1):
            try:
                ch = [(f, c, k - w0) for f, c, k in chains if w0 <= k < w1]
                ctx.hmm_calls_part(np.ascontiguousarray(keep[w0:w1]), obs, row, restart, pows, uniform, ch, ploidy, gid, order, fwd, bwd,
                                   (w0 * room, w1 * room), (2 * w0 * room, 2 * w1 * room), prob2, winner2)
            except Exception as e:      # noqa: BLE001 -- reported below
                errors.append(e)
        threads = [threading.Thread(target=part, args=a) for a in ((0, 2), (2, 3), (3, 6))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        ctx.close()
    assert not errors, errors
    used = np.zeros(n_rows, dtype=bool)
    for w in range(n_windows):
        used[w * room:w * room + room - (w % 3)] = True
    assert np.array_equal(winner[used], winner2[used]) and np.array_equal(prob[used], prob2[used])
    assert (winner[used] != 0xFFFFFFFF).sum() > 100


def _keep_matrix(genotypes, n_hap, perm=None):
    """Haplotypes two genotypes share (multiset intersection, std::set_intersection of the sorted vectors) for all pairs."""
    m = np.zeros((len(genotypes), n_hap), dtype=np.int16)
    for i, g in enumerate(genotypes):
        for h in g:
            m[i, h] += 1
    if perm is not None:
        m = m[perm]
    return np.minimum(m[:, None, :], m[None, :, :]).sum(axis=2).astype(np.uint8)


@pytest.mark.parametrize("ploidy,n_hap,n_want", [(2, 16, 136), (2, 20, 210), (4, 10, 715), (4, 13, 1820), (1, 300, 300), (3, 14, 560)])
def test_recursion_and_posterior_beyond_128_genotypes_equal_x87(ploidy, n_hap, n_want):
    """More genotypes than lanes of the one-lane-per-genotype kernel: `-n 16` and up on a diploid sample (n (n + 1) / 2 pairs),
    and any list the C ABI is handed -- multisets of four out of ten and thirteen haplotypes here.  hmm_recursion_big_kernel
    (several genotypes per lane, keep matrix read by rows from global memory) and hmm_posterior_big_kernel against the same
    computations in numpy.longdouble, bit for bit."""
    assert np.finfo(LD).nmant == 63
    rng = np.random.default_rng(ploidy * 1000 + n_hap)
    genotypes = list(itertools.combinations_with_replacement(range(n_hap), ploidy))
    n = len(genotypes)
    assert n == n_want
    n_windows = 2
    keep = np.stack([_keep_matrix(genotypes, n_hap, rng.permutation(n)) for _ in range(n_windows)])
    assert np.array_equal(keep[0], keep[0].T)
    n_rows = 10 if n > 1000 else 16
    expo = rng.choice([0, -20, -300, -2000, -4800, -4940, -4960], size=(n_rows, n), p=[.15, .2, .25, .2, .1, .05, .05])
    obs = (rng.random((n_rows, n)).astype(LD) + LD(0.01)) * np.power(LD(10), expo.astype(LD))
    obs[rng.random((n_rows, n)) < 0.02] = 0
    obs[3] = 0
    # window w: rows [w * half, (w + 1) * half) forward, then backward
    half = n_rows // n_windows
    steps_row, steps_restart, pows, chains = [], [], [], []
    fwd = np.zeros(n_rows, dtype=np.uint64)
    bwd = np.zeros(n_rows, dtype=np.uint64)
    for w in range(n_windows):
        for direction in (1, -1):
            rows = list(range(w * half, (w + 1) * half))[::direction]
            first = len(steps_row)
            for i, r in enumerate(rows):
                (fwd if direction == 1 else bwd)[r] = len(steps_row)
                steps_row.append(r)
                steps_restart.append(1 if i == 0 or (i % 4 == 3 and w == 1) else 0)
                d = LD(rng.integers(1, 50_000))
                recomb = (LD(1) - np.exp(-d / LD(30))) * (LD(1) / LD(30))
                no_recomb = np.exp(-d / LD(30)) + recomb
                pows.append([[no_recomb ** LD(k) for k in range(ploidy + 1)], [recomb ** LD(k) for k in range(ploidy + 1)]])
            chains.append((first, len(rows), w))
    pows = np.array(pows, dtype=LD)
    uniform = LD(1) / LD(n)
    gid = rng.integers(0, 7, size=(n_rows, n), dtype=np.uint8)
    order = np.full((n_rows, n), 0xFF, dtype=np.uint8)
    order[:, :7] = [4, 0, 6, 2, 1, 5, 3]
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        prob, winner, ab = ctx.hmm_calls(keep, obs, steps_row, steps_restart, pows, uniform, chains, ploidy, gid, order, fwd, bwd)
        # a keep matrix that is not symmetric is refused beyond 128 genotypes (the kernel reads it by rows)
        bad = keep.copy()
        bad[0, 1, 0] ^= 1
        with pytest.raises(Exception):
            ctx.hmm_recursion(bad, obs, steps_row, steps_restart, pows, uniform, chains, ploidy)
    finally:
        ctx.close()
    for first, count, w in chains:
        want = _host_chain(keep[w], [obs[r] for r in steps_row[first:first + count]], steps_restart[first:first + count],
                           pows[first:first + count], uniform, ploidy)
        g = ab[first:first + count]
        assert np.array_equal(g, want), (ploidy, n, w, int(np.argmax((g != want).any(axis=1))))
    assert (ab > 0).any() and (ab == 0).any()
    n_called = 0
    for r in range(n_rows):
        a, b = ab[fwd[r]], ab[bwd[r]]
        den = LD(0)
        p_ = a * b
        for g in range(n):
            den = den + p_[g]
        if den == 0:
            assert winner[r] == 0xFFFFFFFF
            continue
        post = p_ / den
        sums = {}
        for g in range(n):
            sums[gid[r, g]] = sums.get(gid[r, g], LD(0)) + post[g]
        best, best_id = LD(-1), None
        for k in order[r]:
            if k == 0xFF:
                break
            if sums[k] > best:
                best, best_id = sums[k], k
        mx, win = LD(0), 0xFFFFFFFF
        for g in range(n):
            if gid[r, g] == best_id and mx < post[g]:
                mx, win = post[g], g
        assert winner[r] == win and prob[r] == best, r
        n_called += win != 0xFFFFFFFF
    assert n_called >= n_rows // 2


def test_emission_scores_on_the_device_equal_the_host_arithmetic():
    """vgmi_hmm_emissions (hmm_emissions_kernel): hidden states and observable states of a node (src/genotype.cpp:640-830, 960-1000,
    most_likely_depth :1118-1145) against the same computation spelled out here -- float32 / float64 steps as the host's SSE
    code takes them, the product in numpy.longdouble (x87) in k-mer order -- on random node lists: coverage 0..255, multiplicity
    1..4, reference-allele flags, k-mers inside and outside the Poisson interval, terms down to 1e-300 so that products run through
    the denormal range; plus the per-row counts and the two flags."""
    assert np.finfo(LD).nmant == 63
    rng = np.random.default_rng(2027)
    n_hap, bit_len = 15, 2
    used = np.arange(n_hap, dtype=np.uint8)
    pairs = list(itertools.combinations_with_replacement(range(n_hap), 2))
    pos_a = np.array([a for a, _ in pairs], dtype=np.uint8)
    pos_b = np.array([b for _, b in pairs], dtype=np.uint8)
    n_gt = len(pairs)
    top_mask = (1 << n_hap) - 1
    ave = np.float32(23.5)
    lower, upper = float(ave) - 1.96 * float(np.sqrt(np.float64(ave))), float(ave) + 1.96 * float(np.sqrt(np.float64(ave)))
    expo = rng.integers(-300, 1, size=768)
    tables = (rng.random(768).astype(LD) + LD(0.05)) * np.power(LD(10), expo.astype(LD))
    n_rows = 300
    counts = rng.integers(0, 70, size=n_rows)
    counts[5] = 0
    entry_begin = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    n_entries = int(counts.sum())
    f = rng.choice([1, 1, 1, 2, 3, 4], size=n_entries).astype(np.uint64)
    bits = rng.integers(1, 1 << n_hap, size=n_entries).astype(np.uint64)         # some haplotype carries every k-mer
    lb = rng.integers(0, 2, size=n_entries).astype(np.uint64)
    bits |= lb << np.uint64(8 * bit_len - 1)
    cov = rng.choice([0, 1, 5, 14, 15, 20, 23, 24, 30, 33, 34, 60, 255], size=n_entries).astype(np.uint8)
    entries = (f << np.uint64(8)) | (bits << np.uint64(16))
    gt0 = rng.integers(0, 1 << n_hap, size=n_rows).astype(np.uint16)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        obs, n_kept, flags = ctx.hmm_emissions(entries, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0)
        # an entry no selected haplotype carries must be reported (the caller's guarantee is checked, not assumed)
        e2 = entries.copy()
        e2[int(entry_begin[7])] &= np.uint64(0xFFFF) | (np.uint64(1) << np.uint64(16 + 8 * bit_len - 1))
        _, _, flags2 = ctx.hmm_emissions(e2, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0)
    finally:
        ctx.close()
    assert flags2[7] & 2 and not (flags & 2).any()

    def mld(h, c, ff):
        if ff == 1:
            return c
        cf = np.float32(c)
        if h > 0 and cf > ave * np.float32(h):
            return int(ave * np.float32(h)) & 0xFF
        if h == 0 and cf > ave:
            return 0 if float(ff) > float(cf) / upper else int(cf / np.float32(ff)) & 0xFF
        if h == 0:
            return int(cf / np.float32(ff)) & 0xFF
        return c
    n_flagged = 0
    for r in range(n_rows):
        prod = np.ones(n_gt, dtype=LD)
        flag = 0
        for j in range(int(entry_begin[r]), int(entry_begin[r]) + int(counts[r])):
            c, ff, b = int(cov[j]), int(f[j]), int(bits[j])
            l = (b >> (8 * bit_len - 1)) & 1
            in_interval = l == 1 and lower <= c <= upper
            one = [1 if (in_interval and (int(gt0[r]) >> p) & 1) else (b >> p) & 1 for p in range(n_hap)]
            if c < lower and ff >= 2 and any(one):
                flag |= 1
            fj = 2 if (l == 1 and ff == 1) else ff
            term = {}
            for h in (0, 1, 2):
                term[h] = tables[h * 256 + mld(h, c, fj)]
            hs = np.array([one[a] + one[b2] for a, b2 in pairs])
            prod = prod * np.where(hs == 0, term[0], np.where(hs == 1, term[1], term[2]))
        assert n_kept[r] == counts[r] and flags[r] == flag, r
        assert np.array_equal(obs[r], prod), (r, int(np.argmax(obs[r] != prod)))
        n_flagged += flag
    assert 20 < n_flagged < n_rows and (obs == 0).any() and (obs > 0).any()


def test_flagged_rows_scored_again_on_the_device_and_plans():
    """Round 5.  (a) vgmi_hmm_part_fix_rows: rows scored again with haplotypes taken off entries (what the host's sequence check rules
    out, src/genotype.cpp:760-800) equal the product spelled out here with those haplotypes' bits cleared; rows that are not named keep
    their scores.  (b) vgmi_hmm_plan: a part's recursion and posterior on device-resident inputs give the bits of vgmi_hmm_part_calls
    with host arrays, call after call."""
    rng = np.random.default_rng(77)
    n_hap, bit_len = 9, 2
    used = np.arange(n_hap, dtype=np.uint8)
    pairs = list(itertools.combinations_with_replacement(range(n_hap), 2))
    pos_a = np.array([a for a, _ in pairs], dtype=np.uint8)
    pos_b = np.array([b for _, b in pairs], dtype=np.uint8)
    n_gt = len(pairs)
    top_mask = (1 << n_hap) - 1
    ave = np.float32(23.5)
    lower, upper = float(ave) - 1.96 * float(np.sqrt(np.float64(ave))), float(ave) + 1.96 * float(np.sqrt(np.float64(ave)))
    tables = (rng.random(768).astype(LD) + LD(0.05)) * np.power(LD(10), rng.integers(-40, 1, size=768).astype(LD))
    n_rows = 120
    counts = rng.integers(1, 60, size=n_rows)
    entry_begin = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    n_entries = int(counts.sum())
    f = rng.choice([1, 1, 2, 3], size=n_entries).astype(np.uint64)
    bits = rng.integers(1, 1 << n_hap, size=n_entries).astype(np.uint64)
    lb = rng.integers(0, 2, size=n_entries).astype(np.uint64)
    bits |= lb << np.uint64(8 * bit_len - 1)
    cov = rng.choice([0, 1, 5, 14, 15, 20, 23, 24, 30, 60], size=n_entries).astype(np.uint8)
    entries = (f << np.uint64(8)) | (bits << np.uint64(16))
    gt0 = rng.integers(0, 1 << n_hap, size=n_rows).astype(np.uint16)
    # fixes on every third row: a few entries each, ascending, random masks
    f_rows, f_off, f_j, f_m = [], [0], [], []
    drop = {}
    for r in range(0, n_rows, 3):
        js = np.sort(rng.choice(int(counts[r]), size=min(int(counts[r]), int(rng.integers(1, 5))), replace=False))
        f_rows.append(r)
        for j in js:
            m = int(rng.integers(1, 1 << n_hap))
            f_j.append(int(j))
            f_m.append(m)
            drop[(r, int(j))] = m
        f_off.append(len(f_j))
    # a chain over the rows in both directions (one window), random step tables and strings
    stride = 3
    n_steps = 2 * n_rows
    row = np.concatenate([np.arange(n_rows), np.arange(n_rows)[::-1]]).astype(np.uint32)
    restart = np.zeros(n_steps, dtype=np.uint8)
    restart[0] = restart[n_rows] = 1
    pw = (rng.random((n_steps, 2 * stride)).astype(LD) * LD(0.5) + LD(0.25))
    keep = np.array([[len({a, b} & {c, d}) + (a == b == c == d) for (c, d) in pairs] for (a, b) in pairs], dtype=np.uint8)
    keep = np.minimum(keep, 2).astype(np.uint8)
    keep = np.maximum(keep, keep.T)
    gid = rng.integers(0, 6, size=(n_rows, n_gt)).astype(np.uint8)
    order = np.tile(np.arange(n_gt, dtype=np.uint8), (n_rows, 1))
    for r in range(n_rows):      # order: the distinct ids of the row, then padding (as genotype_strings lays them out)
        ids = np.unique(gid[r])
        order[r, :ids.size] = ids
        order[r, ids.size:] = 0xFF
    fwd = np.arange(n_rows, dtype=np.uint64)
    bwd = (n_rows + (n_rows - 1 - np.arange(n_rows))).astype(np.uint64)
    calls = dict(ploidy=2, keep=keep[None], row=row, restart=restart, pow=pw, uniform=LD(1) / LD(n_gt), chains=[(0, n_rows, 0), (n_rows, n_rows, 0)],
                 gid=gid, order=order, fwd=fwd, bwd=bwd)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        obs0, n_kept, flags = ctx.hmm_emissions(entries, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0)
        obs1, n_kept1, flags1, both = ctx.hmm_emissions(entries, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0,
                                                        fixes=(f_rows, f_off, f_j, f_m), calls=calls)
    finally:
        ctx.close()
    assert np.array_equal(n_kept, n_kept1) and np.array_equal(flags, flags1)

    def mld(h, c, ff):
        if ff == 1:
            return c
        cf = np.float32(c)
        if h > 0 and cf > ave * np.float32(h):
            return int(ave * np.float32(h)) & 0xFF
        if h == 0 and cf > ave:
            return 0 if float(ff) > float(cf) / upper else int(cf / np.float32(ff)) & 0xFF
        if h == 0:
            return int(cf / np.float32(ff)) & 0xFF
        return c
    changed = 0
    for r in range(n_rows):
        if r % 3:
            assert np.array_equal(obs1[r], obs0[r]), r
            continue
        prod = np.ones(n_gt, dtype=LD)
        for jj in range(int(counts[r])):
            j = int(entry_begin[r]) + jj
            c, ff, b = int(cov[j]), int(f[j]), int(bits[j])
            l = (b >> (8 * bit_len - 1)) & 1
            in_interval = l == 1 and lower <= c <= upper
            one = [1 if (in_interval and (int(gt0[r]) >> p) & 1) else (b >> p) & 1 for p in range(n_hap)]
            m = drop.get((r, jj), 0)
            one = [o if not (m >> p) & 1 else 0 for p, o in enumerate(one)]
            fj = 2 if (l == 1 and ff == 1) else ff
            term = {h: tables[h * 256 + mld(h, c, fj)] for h in (0, 1, 2)}
            hs = np.array([one[a] + one[b2] for a, b2 in pairs])
            prod = prod * np.where(hs == 0, term[0], np.where(hs == 1, term[1], term[2]))
        assert np.array_equal(obs1[r], prod), r
        changed += not np.array_equal(obs1[r], obs0[r])
    assert changed > n_rows // 6
    (p1, w1), (p2, w2), (p3, w3) = both
    assert np.array_equal(p1.view(np.uint8), p2.view(np.uint8)) and np.array_equal(w1, w2)
    assert np.array_equal(p1.view(np.uint8), p3.view(np.uint8)) and np.array_equal(w1, w3)
    assert (w1 != 0xFFFFFFFF).sum() > n_rows // 2


def test_fix_of_an_entry_beyond_65535_of_its_row():
    """an entry's index inside its row is 32 bits from sequence_fixes through vgmi_hmm_part_fix_rows to the kernel's test
    (it was uint16_t: a row of more than 65 535 entries -- the API takes any entry_count; graph2node keeps 128 a node,
    src/construct_index.cpp:1592-1596 -- had its late fixes refused or applied to entry j mod 65 536).  One row of 70 000 entries, fixes at
    entries 3, 65 536 + 3 and 69 990: the product spelled out with those haplotypes' bits cleared (src/genotype.cpp:760-800)."""
    rng = np.random.default_rng(65536)
    n_hap, bit_len = 5, 1
    used = np.arange(n_hap, dtype=np.uint8)
    pairs = list(itertools.combinations_with_replacement(range(n_hap), 2))
    pos_a = np.array([a for a, _ in pairs], dtype=np.uint8)
    pos_b = np.array([b for _, b in pairs], dtype=np.uint8)
    n_gt = len(pairs)
    top_mask = (1 << n_hap) - 1
    ave = np.float32(23.5)
    lower, upper = float(ave) - 1.96 * float(np.sqrt(np.float64(ave))), float(ave) + 1.96 * float(np.sqrt(np.float64(ave)))
    tables = LD(1) - rng.random(768).astype(LD) * LD(1e-4)          # 70 000 factors stay in range
    counts = np.array([70000, 40], dtype=np.int64)
    entry_begin = np.array([0, 70000], dtype=np.uint64)
    n_entries = int(counts.sum())
    f = rng.choice([2, 3], size=n_entries).astype(np.uint64)
    bits = rng.integers(1, 1 << n_hap, size=n_entries).astype(np.uint64)      # bit 7 (the interval flag of bit_len 1) stays clear
    cov = rng.choice([0, 1, 5, 14, 20, 23, 30, 60], size=n_entries).astype(np.uint8)
    entries = (f << np.uint64(8)) | (bits << np.uint64(16))
    gt0 = np.zeros(2, dtype=np.uint16)
    drop = {3: 0b00101, 65536 + 3: 0b11000, 69990: 0b00011}
    for j, m in drop.items():
        bits[j] |= np.uint64(m)                                                 # the haplotypes taken off really carry the k-mer
    entries = (f << np.uint64(8)) | (bits << np.uint64(16))
    f_j = np.array(sorted(drop), dtype=np.uint32)
    fixes = ([0], [0, f_j.size], f_j, [drop[int(j)] for j in f_j])
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        obs0, n_kept, flags = ctx.hmm_emissions(entries, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0)
        obs1, n_kept1, flags1 = ctx.hmm_emissions(entries, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0,
                                                  fixes=fixes)[:3]
        with pytest.raises(RuntimeError):      # an index behind the row's last entry is refused, whatever its low 16 bits say
            ctx.hmm_emissions(entries, cov, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0,
                              fixes=([0], [0, 1], [70000 + 3], [1]))
    finally:
        ctx.close()
    assert np.array_equal(n_kept, n_kept1) and np.array_equal(obs1[1], obs0[1])

    def mld(h, c, ff):
        cf = np.float32(c)
        if h > 0 and cf > ave * np.float32(h):
            return int(ave * np.float32(h)) & 0xFF
        if h == 0 and cf > ave:
            return 0 if float(ff) > float(cf) / upper else int(cf / np.float32(ff)) & 0xFF
        if h == 0:
            return int(cf / np.float32(ff)) & 0xFF
        return c
    pa, pb = pos_a.astype(np.int64), pos_b.astype(np.int64)
    for fixed, obs in ((False, obs0), (True, obs1)):
        prod = np.ones(n_gt, dtype=LD)
        for j in range(70000):
            c, ff, b = int(cov[j]), int(f[j]), int(bits[j])
            if fixed:
                b &= ~drop.get(j, 0)
            hs = ((b >> pa) & 1) + ((b >> pb) & 1)
            t = np.array([tables[h * 256 + mld(h, c, ff)] for h in (0, 1, 2)], dtype=LD)
            prod = prod * t[hs]
        assert np.array_equal(obs[0], prod), fixed
    assert not np.array_equal(obs0[0], obs1[0])


@pytest.mark.parametrize("ploidy", [3, 4])
def test_emission_scores_of_polyploid_genotypes_on_the_device(ploidy):
    """vgmi_hmm_emissions_ploidy (round 5): genotypes of three and four haplotypes -- a polyploid sample's are blocks of consecutive
    haplotypes of the panel (src/genotype.cpp:846-873), a haplotype may stand in a genotype more than once -- against the product spelled
    out in numpy.longdouble: copy numbers 0 .. ploidy, Poisson terms per copy number."""
    rng = np.random.default_rng(900 + ploidy)
    n_hap, bit_len = 13, 2
    used = np.arange(n_hap, dtype=np.uint8)
    blocks = [[(b + q) % n_hap for q in range(ploidy)] for b in range(0, n_hap - 1, ploidy)] + [[0] * ploidy, [1, 1] + [2] * (ploidy - 2)]
    pos = np.array(blocks, dtype=np.uint8)
    n_gt = pos.shape[0]
    top_mask = (1 << n_hap) - 1
    ave = np.float32(11.25)
    lower, upper = float(ave) - 1.96 * float(np.sqrt(np.float64(ave))), float(ave) + 1.96 * float(np.sqrt(np.float64(ave)))
    tables = (rng.random((ploidy + 1) * 256).astype(LD) + LD(0.05)) * np.power(LD(10), rng.integers(-200, 1, size=(ploidy + 1) * 256).astype(LD))
    n_rows = 150
    counts = rng.integers(0, 70, size=n_rows)
    entry_begin = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    n_entries = int(counts.sum())
    f = rng.choice([1, 1, 1, 2, 3], size=n_entries).astype(np.uint64)
    bits = rng.integers(1, 1 << n_hap, size=n_entries).astype(np.uint64)
    lb = rng.integers(0, 2, size=n_entries).astype(np.uint64)
    bits |= lb << np.uint64(8 * bit_len - 1)
    cov = rng.choice([0, 1, 4, 5, 8, 11, 12, 17, 18, 30, 255], size=n_entries).astype(np.uint8)
    entries = (f << np.uint64(8)) | (bits << np.uint64(16))
    gt0 = rng.integers(0, 1 << n_hap, size=n_rows).astype(np.uint16)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        obs, n_kept, flags = ctx.hmm_emissions(entries, cov, used, None, None, top_mask, bit_len, ave, lower, upper, tables, entry_begin, counts, gt0, pos=pos)
    finally:
        ctx.close()

    def mld(h, c, ff):
        if ff == 1:
            return c
        cf = np.float32(c)
        if h > 0 and cf > ave * np.float32(h):
            return int(ave * np.float32(h)) & 0xFF
        if h == 0 and cf > ave:
            return 0 if float(ff) > float(cf) / upper else int(cf / np.float32(ff)) & 0xFF
        if h == 0:
            return int(cf / np.float32(ff)) & 0xFF
        return c
    seen_h = set()
    for r in range(n_rows):
        prod = np.ones(n_gt, dtype=LD)
        for j in range(int(entry_begin[r]), int(entry_begin[r]) + int(counts[r])):
            c, ff, b = int(cov[j]), int(f[j]), int(bits[j])
            l = (b >> (8 * bit_len - 1)) & 1
            in_interval = l == 1 and lower <= c <= upper
            one = [1 if (in_interval and (int(gt0[r]) >> p) & 1) else (b >> p) & 1 for p in range(n_hap)]
            fj = 2 if (l == 1 and ff == 1) else ff
            hs = np.array([sum(one[p] for p in blk) for blk in blocks])
            seen_h.update(hs.tolist())
            prod = prod * np.array([tables[h * 256 + mld(h, c, fj)] for h in hs], dtype=LD)
        assert n_kept[r] == counts[r], r
        assert np.array_equal(obs[r], prod), (r, int(np.argmax(obs[r] != prod)))
    assert seen_h == set(range(ploidy + 1))
