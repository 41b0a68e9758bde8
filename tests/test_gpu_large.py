"""BASELINE.json configs 3-5 under test: tables that live in HBM.

C3 (chr20 class: 60 Mb reference, 500 k SNPs, 2.56e7 graph k-mers, 12 M read pairs) at full size through the C ABI,
checked against the oracle on a 2 M-read prefix and through size-independent properties on the whole sample; the
WGS-class table geometry (>= 2^27 keys: slot numbers beyond 2^31) on a small read set; and the ordering of the per-sample
reset against host-staged blocks (the reset runs on the main stream, host blocks on the stages' own)."""
import numpy as np
import pytest

import oracle_lib as o
from varigraph_amd import synth, vgmi

pytestmark = pytest.mark.gpu
L = 150


@pytest.fixture(scope="module")
def chr20():
    import torch
    keys, haps = synth.snp_graph(60_000_000, 500_000)
    ctx = vgmi.Context(0, buffer_mib=64)
    ctx.table_upload(keys, 27)
    cat = np.concatenate(haps)
    off = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.uint64)
    n_reads = 24_000_000
    d_cat = torch.from_numpy(cat).cuda()
    d_block = torch.empty(n_reads * (L + 1), dtype=torch.uint8, device="cuda")
    for first in range(0, n_reads, 8_000_000):
        ctx.synth_reads_device(99, first, 8_000_000, L, d_cat, off, d_block[first * (L + 1):])
    del d_cat
    yield {"ctx": ctx, "keys": keys, "block": d_block, "n_reads": n_reads}
    ctx.close()


def test_c3_full_size_prefix_equals_oracle_and_properties(chr20):
    ctx, keys, d_block, n_reads = chr20["ctx"], chr20["keys"], chr20["block"], chr20["n_reads"]
    assert keys.size > 2.5e7
    ctx.counts_reset()
    ctx.reads_submit_device(d_block, d_block.numel(), n_reads)
    full, _, _ = ctx.counts_finish()
    assert ctx.read_base() == n_reads * L
    # (1) one launch == ragged sub-launches cut at 16-read multiples (16-byte aligned device pointers)
    cuts = [0, 16 * 100_003, 16 * 700_001, 16 * 700_002, 16 * 1_299_999, n_reads]
    ctx.counts_reset()
    for a, b in zip(cuts[:-1], cuts[1:]):
        ctx.reads_submit_device(d_block[a * (L + 1):], (b - a) * (L + 1), b - a)
    split, _, _ = ctx.counts_finish()
    assert np.array_equal(full, split)
    # (2) a 2 M-read prefix, counter by counter against the oracle
    pre = 2_000_000
    ctx.counts_reset()
    ctx.reads_submit_device(d_block, pre * (L + 1), pre)
    part, _, _ = ctx.counts_finish()
    t = o.Table(keys)
    t.count_block(d_block[: pre * (L + 1)].cpu().numpy(), 27)
    want = t.counts()
    assert np.array_equal(part, want)
    assert want.sum() > 10 * pre          # the dense graph: ~20 hits per read
    # (3) monotone in the input, (4) 30x coverage touches nearly every k-mer of the sequenced haplotypes
    assert (full >= part).all()
    assert (full != 0).mean() > 0.85      # (keys that pair an allele with a neighbouring site's other allele are not on the sample)
    assert int(full.max()) < 255 or (full == 255).sum() < 100


def test_reset_is_ordered_before_host_staged_blocks(chr20):
    """counts_reset (memset + flag sweep of a 3 GiB table, on the main stream) followed at once by a small
    vgmi_reads_submit (pinned staging on the stages' own streams): no count may be lost to the reset."""
    ctx, keys, d_block = chr20["ctx"], chr20["keys"], chr20["block"]
    small = d_block[: 20_000 * (L + 1)].cpu().numpy()
    t = o.Table(keys)
    t.count_block(small, 27)
    want = t.counts()
    for _ in range(3):
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, 4_000_000 * (L + 1), 4_000_000)   # leave plenty to clear
        ctx.counts_finish_device(None, None, None)
        ctx.counts_reset()
        ctx.reads_submit(small, 20_000)
        got, _, _ = ctx.counts_finish()
        assert np.array_equal(got, want)


def test_wgs_class_table_geometry_small_read_set():
    """> 2^27 keys: the 2^31-slot table geometry BASELINE config 5 runs with, on a small read set.  The key set is
    1.4e8 distinct 54-bit values (i * odd mod 2^54: hash64 keys of arbitrary 27-mers; the non-canonical half can
    never be emitted by a read and is filler) of which 4 096 are made canonical and planted into reads on both
    strands, so hits land all over the table."""
    n, step = 140_000_000, 34_179
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64((1 << 54) - 1)
    idx = np.arange(0, n, step)[:4096]
    sample = x[idx].copy()
    rc = np.zeros_like(sample)
    y = sample.copy()
    for _ in range(27):
        rc = (rc << np.uint64(2)) | (np.uint64(3) - (y & np.uint64(3)))
        y >>= np.uint64(2)
    sample = np.minimum(sample, rc)
    x[idx] = sample
    keys = (synth.hash64_np(x, 27) << np.uint64(8)) | np.uint64(27)
    del x
    codes = np.empty((4096, 27), dtype=np.uint8)
    for j in range(27):
        codes[:, j] = ((sample >> np.uint64(2 * (26 - j))) & np.uint64(3)).astype(np.uint8)
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    for i in range(4096):
        seq = acgt[3 - codes[i][::-1]] if i & 1 else acgt[codes[i]]   # odd reads carry the reverse complement
        filler = acgt[rng.integers(0, 4, size=150 - 27)]
        reads.append(np.concatenate([filler[:30], seq, filler[30:], np.array([10], dtype=np.uint8)]))
    block = np.ascontiguousarray(np.concatenate(reads), dtype=np.uint8)
    ctx = vgmi.Context(0, buffer_mib=16)
    try:
        ctx.table_upload(keys, 27)      # a duplicate among the planted keys would fail here (VGMI_E_DUPLICATE_KEY)
        info = ctx.table_info()
        assert info["n_slots"] >= 1 << 30
        ctx.counts_reset()
        ctx.reads_submit(block, 4096)
        got, _, _ = ctx.counts_finish()
        ctx.counts_reset()
        ctx.reads_submit(block, 4096)
        ctx.reads_submit(block, 4096)
        twice, _, _ = ctx.counts_finish()
    finally:
        ctx.close()
    # expected counts without a 1.4e8-entry CPU hash table: the oracle's emitted keys of every read, looked up by
    # binary search in the sorted key list
    order = np.argsort(keys)
    sk = keys[order]
    emitted = []
    start = 0
    for e in np.flatnonzero(block == 10).tolist():
        emitted.append(o.sketch(block[start:e].tobytes(), 27))
        start = e + 1
    emitted = np.concatenate(emitted)
    pos = np.searchsorted(sk, emitted)
    pos[pos == sk.size] = 0
    hit = sk[pos] == emitted
    assert hit.sum() >= 4096
    want = np.zeros(keys.size, dtype=np.int64)
    np.add.at(want, order[pos[hit]], 1)
    assert np.array_equal(got, np.minimum(want, 255).astype(np.uint8))
    assert np.array_equal(twice, np.minimum(2 * want, 255).astype(np.uint8))


def _oracle_counts_by_search(sorted_keys, block, n_reads):
    """Expected counters without a CPU hash table over 2.7e8 keys: the oracle's emitted keys of every read (vgo_sketch, the
    reference's state machine), looked up by binary search in the sorted key list.  Returns (index array, count array)."""
    rows = block.reshape(n_reads, L + 1)
    emitted = np.concatenate([o.sketch(rows[i, :L].tobytes(), 27) for i in range(n_reads)])
    pos = np.searchsorted(sorted_keys, emitted)
    pos[pos == sorted_keys.size] = 0
    hit = sorted_keys[pos] == emitted
    idx, cnt = np.unique(pos[hit], return_counts=True)
    return idx, cnt


def test_c5_wgs_class_single_gpu_slice():
    """BASELINE config 5's single-GPU slice at its stated size: the whole-genome class graph (3 Gb reference, 5 M SNPs,
    2.67e8 graph k-mers) resident in HBM -- compact image, context table at 30 % load (~20 GB) -- NEXT TO
    the construct-side Bloom filter of a 3 Gb genome (28.8 GB), 2e7 reads generated on the device.  Checked: a prefix
    counter by counter against the oracle's emitted keys, whole == ragged split, monotone, nothing spilled past the table."""
    import torch
    keys, haps = synth.snp_graph(3_000_000_000, 5_000_000)
    assert keys.size > 2.6e8 and np.all(keys[1:] > keys[:-1])     # np.unique: sorted, so binary search needs no argsort
    ctx = vgmi.Context(0, buffer_mib=64)
    try:
        m, nh = vgmi.bloom_params(3_000_000_000, 0.01)
        assert m > 28_000_000_000
        ctx.bloom_create(m, nh, np.arange(1, nh + 1, dtype=np.uint64))      # resident for the whole test
        ctx.table_upload(keys, 27)
        info, xinfo = ctx.table_info(), ctx.ctable_info()
        assert info["n_keys"] == keys.size and info["n_slots"] >= 1 << 31
        # the default table of this class (context table at <= 30 % load), not the fallback for a short device
        assert xinfo["n_buckets"] * 4 * 0.31 >= xinfo["n_entries"] > keys.size and xinfo["overflow_kmers"] < keys.size // 100
        free_b, total_b = ctx.device_memory()
        assert total_b - free_b > 60e9            # image + context table + Bloom really are resident together
        n_reads = 20_000_000
        off = np.array([0, haps[0].size, haps[0].size + haps[1].size], dtype=np.uint64)
        d_cat = torch.empty(int(off[-1]), dtype=torch.uint8, device="cuda")
        d_cat[: haps[0].size] = torch.from_numpy(haps[0]).cuda()
        d_cat[haps[0].size:] = torch.from_numpy(haps[1]).cuda()
        del haps
        d_block = torch.empty(n_reads * (L + 1), dtype=torch.uint8, device="cuda")
        for first in range(0, n_reads, 10_000_000):
            ctx.synth_reads_device(4711, first, 10_000_000, L, d_cat, off, d_block[first * (L + 1):])
        del d_cat
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, d_block.numel(), n_reads)
        full, _, _ = ctx.counts_finish()
        ms_full, _ = ctx.count_kernel_ms()
        assert ctx.read_base() == n_reads * L
        cuts = [0, 16 * 70_001, 16 * 600_000, 16 * 600_001, n_reads]
        ctx.counts_reset()
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctx.reads_submit_device(d_block[a * (L + 1):], (b - a) * (L + 1), b - a)
        split, _, _ = ctx.counts_finish()
        assert np.array_equal(full, split)
        pre = 150_000
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, pre * (L + 1), pre)
        part, _, _ = ctx.counts_finish()
        idx, cnt = _oracle_counts_by_search(keys, d_block[: pre * (L + 1)].cpu().numpy(), pre)
        want = np.zeros(keys.size, dtype=np.uint8)
        want[idx] = np.minimum(cnt, 255).astype(np.uint8)
        assert np.array_equal(part, want)
        assert int(cnt.sum()) > 4 * pre           # ~4.7 hits per read on this graph
        assert (full >= part).all()
        assert 4.0 < full.astype(np.int64).sum() / n_reads < 6.0
        print(f"C5 slice: {n_reads} reads in {ms_full:.1f} ms of count kernel, {xinfo['n_buckets'] * 64 / 1e9:.1f} GB context table, {xinfo}")
    finally:
        ctx.close()
