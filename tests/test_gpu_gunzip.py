"""Ordinary gzip members inflated on the device (vgmi_gunzip.hip through vgmi_gunzip_buffer) against zlib: FASTQ text of several
shapes and sizes at several compression levels -- stretches between guessed block starts decoded side by side with placeholders for
the window in front, chained, resolved.  What the device hands back must be a PREFIX of zlib's output ending at a block boundary,
and the whole of it when it reports the member's end."""
import gzip
import zlib

import numpy as np
import pytest

from varigraph_amd import vgmi

pytestmark = pytest.mark.gpu


def _fastq_text(n_reads, seed, qual="fixed", read_len=150):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=200_000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = []
    for i in range(n_reads):
        s = int(rng.integers(0, genome.size - read_len))
        seq = acgt[genome[s:s + read_len]].tobytes()
        if qual == "fixed":
            q = b"I" * read_len
        else:
            q = bytes((33 + rng.integers(2, 41, size=read_len)).astype(np.uint8))
        parts.append(b"@read%09d/1 lane:%d\n" % (i, i % 8) + seq + b"\n+\n" + q + b"\n")
    return b"".join(parts)


@pytest.fixture(scope="module")
def ctx():
    c = vgmi.Context(0, buffer_mib=16)
    yield c
    c.close()


@pytest.mark.parametrize("n_reads,qual,level", [(300, "fixed", 6), (20_000, "fixed", 1), (20_000, "fixed", 6), (20_000, "random", 6),
                                                 (20_000, "random", 9), (200_000, "fixed", 4), (200_000, "random", 6)])
def test_device_gunzip_equals_zlib(ctx, n_reads, qual, level):
    text = _fastq_text(n_reads, 7 + n_reads, qual)
    comp = gzip.compress(text, level)
    got, consumed, member_end, why = ctx.gunzip(comp, len(text) + 4096)
    assert why == 0, why
    assert member_end and got == text
    assert consumed == len(comp) - 8          # everything but the trailer


def test_device_gunzip_stored_and_fixed_blocks_and_garbage(ctx):
    # incompressible bytes (stored blocks), a tiny member (fixed codes), and a damaged stream: a prefix, never wrong bytes
    rng = np.random.default_rng(3)
    noise = bytes(rng.integers(0, 256, size=300_000, dtype=np.uint8))
    for data in (noise, b"ACGT" * 10, _fastq_text(5000, 1)):
        comp = gzip.compress(data, 6)
        got, consumed, member_end, why = ctx.gunzip(comp, len(data) + 4096)
        assert got == data[:len(got)]
        if member_end:
            assert got == data
    # a flipped byte in the middle: the device hands back a prefix of what zlib makes of the same (damaged) bytes, block by block
    text = _fastq_text(50_000, 2)
    for at in (0.5, 0.25, 0.9):
        comp = bytearray(gzip.compress(text, 6))
        comp[int(len(comp) * at)] ^= 0x55
        got, consumed, member_end, why = ctx.gunzip(bytes(comp), len(text) + 65536)
        d = zlib.decompressobj(-15)
        want = b""
        try:
            for i in range(10, len(comp), 1 << 16):
                want += d.decompress(bytes(comp[i:i + (1 << 16)]))
        except zlib.error:
            pass
        assert len(got) > 0 and got == want[:len(got)]


def _graph_ctx():
    import os
    from conftest import get_cohort
    from varigraph_amd import host
    cohort = get_cohort("cohort_snp")
    g = host.Graph(os.path.join(cohort.dir, "graph.bin.gz"))
    c = vgmi.Context(0, buffer_mib=16)
    g.upload(c)
    return g, c, cohort


def test_gzip_stream_through_the_device_parser_and_counters():
    """vgmi_fastq_commit_gzip: an ordinary gzip FASTQ file inflated, parsed and counted on the device, whole and in pieces (what a piece
    leaves untaken is presented again), two members back to back, and a stream the device must give up in the middle -- the counters
    of what it took are those of the text's prefix it reports."""
    import oracle_lib as o
    g, c, cohort = _graph_ctx()
    try:
        hap = cohort.haplotypes()[1]
        rng = np.random.default_rng(5)
        reads = [hap[s:s + 150].tobytes() for s in rng.integers(0, len(hap) - 150, size=60_000)]
        text = b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * 150 + b"\n" for i, r in enumerate(reads))
        block = np.frombuffer(b"".join(r + b"\n" for r in reads), dtype=np.uint8)
        t = o.Table(cohort.graph.keys)
        t.count_block(block, cohort.k)
        want = t.counts()
        comp = gzip.compress(text, 6)
        for piece in (None, 1 << 20, 300_000):
            c.counts_reset()
            r = c.fastq_gzip(comp, piece=piece)
            got, _, _ = c.counts_finish()
            assert r["stop"] == 1 and r["reason"] == 0 and not r["stopped"], (piece, r["stop"], r["reason"])
            assert r["device_text_bytes"] == len(text) and r["taken"] == len(comp)
            assert (r["n_records"], r["consumed"], r["tail"]) == (len(reads), len(text), b"")
            assert np.array_equal(got, want), piece
        # two members: gzread runs them together
        half = len(reads) // 2
        cut = text.index(b"@r%d\n" % half)
        two = gzip.compress(text[:cut], 6) + gzip.compress(text[cut:], 1)
        c.counts_reset()
        r = c.fastq_gzip(two)
        got, _, _ = c.counts_finish()
        assert r["stop"] == 1 and r["n_records"] == len(reads) and np.array_equal(got, want)
        # garbage behind the member is ignored; a damaged stream ends the device path with what was whole
        c.counts_reset()
        r = c.fastq_gzip(comp + b"not gzip at all" * 10)
        got, _, _ = c.counts_finish()
        assert r["stop"] == 1 and r["n_records"] == len(reads) and np.array_equal(got, want)
        bad = bytearray(comp)
        for k in range(len(bad) // 2, len(bad) // 2 + 64):
            bad[k] = 0xFF
        c.counts_reset()
        r = c.fastq_gzip(bytes(bad))
        c.counts_finish()
        # (the damaged stretch either fails -- stop 2, the host decoder goes on from the text so far -- or decodes, as it would in zlib,
        # into bytes that end in a "last" block: stop 1.  Whole files against the host path: tests/test_gpu_ingest.py.)
        # (round 5: ... and is then caught at the member's trailer -- stop 2, reason 11, the host decoder is asked what the file is)
        assert r["stop"] in (1, 2) and 0 < r["device_text_bytes"]
        if r["stop"] == 2 and r["reason"] != 11:
            assert r["reason"] != 0 and r["device_text_bytes"] < len(text) and text[:r["consumed"]].count(b"\n") == 4 * r["n_records"]
        assert r["stop"] == 2      # zlib's CRC-32 catches 64 overwritten bytes: so must the device
    finally:
        c.close()
        g.close()


def test_gzip_stream_scratch_grows_in_the_middle_of_a_member(monkeypatch):
    """a piece larger than any before it re-allocates the decoder's scratch in the middle of a member; the 32 KiB of
    text in front of the piece and the member's running CRC must survive that.  VGMI_GZ_RESERVE=0 sizes the scratch piece by piece (the
    default sizes it once for the largest piece the stream can stage), on a fresh context whose stream pool is empty; pieces grow 40-fold
    and every one of them reaches back into the text of the one before (FASTQ of one haplotype: far matches everywhere)."""
    import oracle_lib as o
    monkeypatch.setenv("VGMI_GZ_RESERVE", "0")
    g, c, cohort = _graph_ctx()
    try:
        hap = cohort.haplotypes()[1]
        rng = np.random.default_rng(9)
        reads = [hap[s:s + 150].tobytes() for s in rng.integers(0, len(hap) - 150, size=120_000)]
        text = b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * 150 + b"\n" for i, r in enumerate(reads))
        block = np.frombuffer(b"".join(r + b"\n" for r in reads), dtype=np.uint8)
        t = o.Table(cohort.graph.keys)
        t.count_block(block, cohort.k)
        want = t.counts()
        comp = gzip.compress(text, 6)
        assert len(comp) > 5_000_000
        for sizes in ([150_000, 600_000, 6_000_000], [100_000, 200_000, 400_000, 800_000, 1_600_000, 8_000_000]):
            c.counts_reset()
            r = c.fastq_gzip(comp, piece=sizes)
            got, _, _ = c.counts_finish()
            assert r["stop"] == 1 and r["reason"] == 0 and not r["stopped"], (sizes, r["stop"], r["reason"])
            assert r["device_text_bytes"] == len(text) and (r["n_records"], r["consumed"]) == (len(reads), len(text))
            assert np.array_equal(got, want), sizes
    finally:
        c.close()
        g.close()


def test_gzip_member_trailer_is_checked_on_the_device(ctx):
    """CRC-32 and ISIZE of the resolved text against the member's trailer -- what zlib checks behind gzread.  A
    clean member passes (whole and in pieces: the remainder is carried across pieces), a trailer with another CRC or another length is
    reason 11, and so is a member whose DEFLATE data decodes to its last block but to other text."""
    text = _fastq_text(30_000, 11, "random")
    comp = gzip.compress(text, 6)
    got, consumed, member_end, why = ctx.gunzip(comp, len(text) + 4096)
    assert (got, member_end, why) == (text, True, 0)
    for at, name in ((-8, "crc"), (-4, "isize")):
        bad = bytearray(comp)
        bad[at] ^= 0x01
        got, consumed, member_end, why = ctx.gunzip(bytes(bad), len(text) + 4096)
        assert got == text and member_end and why == 11, name
    # a literal changed inside a stored block: the stream still ends in its last block, with one other byte of text
    rng = np.random.default_rng(4)
    noise = bytes(rng.integers(0, 256, size=200_000, dtype=np.uint8))
    stored = bytearray(gzip.compress(noise, 0))
    stored[len(stored) // 2] ^= 0x40
    got, consumed, member_end, why = ctx.gunzip(bytes(stored), len(noise) + 4096)
    if member_end:
        assert why == 11 and got != noise and len(got) == len(noise)


def test_gzip_stream_members_one_after_another_and_a_bad_trailer():
    """Members that end inside the staged bytes are followed in the same call (gzread runs members together): three large ones in one
    piece; dozens of tiny ones go to the host decoder (reason 12) with the text counted so far reported exactly; a member whose
    trailer does not match stops the stream there (reason 11) and the members behind it are not read, as behind zlib's error."""
    import oracle_lib as o
    g, c, cohort = _graph_ctx()
    try:
        hap = cohort.haplotypes()[1]
        rng = np.random.default_rng(6)
        reads = [hap[s:s + 150].tobytes() for s in rng.integers(0, len(hap) - 150, size=45_000)]
        rec = [b"@r%d\n" % i + r + b"\n+\n" + b"I" * 150 + b"\n" for i, r in enumerate(reads)]
        text = b"".join(rec)

        def counts(n_reads):
            t = o.Table(cohort.graph.keys)
            t.count_block(np.frombuffer(b"".join(r + b"\n" for r in reads[:n_reads]), dtype=np.uint8), cohort.k)
            return t.counts()

        three = b"".join(gzip.compress(b"".join(rec[i:i + 15_000]), lvl) for i, lvl in ((0, 6), (15_000, 1), (30_000, 9)))
        for piece in (None, 700_000):
            c.counts_reset()
            r = c.fastq_gzip(three, piece=piece)
            got, _, _ = c.counts_finish()
            assert r["stop"] == 1 and r["reason"] == 0 and r["n_records"] == len(reads) and r["taken"] == len(three), (piece, r["stop"], r["reason"])
            assert np.array_equal(got, counts(len(reads))), piece
        tiny = b"".join(gzip.compress(b"".join(rec[i:i + 50]), 6) for i in range(0, 3000, 50))
        c.counts_reset()
        r = c.fastq_gzip(tiny)
        got, _, _ = c.counts_finish()
        assert r["stop"] == 2 and r["reason"] == 12 and 0 < r["n_records"] < 3000 and r["n_records"] % 50 == 0
        assert r["consumed"] == sum(len(x) for x in rec[:r["n_records"]]) and np.array_equal(got, counts(r["n_records"]))
        bad = bytearray(three)
        first = len(gzip.compress(b"".join(rec[:15_000]), 6))
        bad[first - 6] ^= 0x80           # the first member's CRC-32
        c.counts_reset()
        r = c.fastq_gzip(bytes(bad))
        got, _, _ = c.counts_finish()
        assert r["stop"] == 2 and r["reason"] == 11 and r["n_records"] == 15_000 and r["taken"] == first
        assert np.array_equal(got, counts(15_000))
    finally:
        c.close()
        g.close()
