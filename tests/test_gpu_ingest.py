"""Device-side FASTQ parsing (vgmi_fastq_*, csrc/vgmi_fastq.hip) against the host reader -- the literal restatement of
kseq_read (include/kseq.h:192-232 as used by src/fastq_kmer.cpp:97-105) that test_host_cpu.py pins -- and the oracle.

Every file below goes through FastqKmerHip twice, device parser and VGH_HOST_PARSE=1: counters, mReadBase and read count
must be identical, also when tiny staging buffers (VGMI_FASTQ_CHUNK_KB) put every kind of record across a chunk boundary,
and when the file is not regular four-line FASTQ at all (the device parser must hand over at the right byte)."""
import gzip
import os

import numpy as np
import pytest

import oracle_lib as o
from conftest import get_cohort
from varigraph_amd import host, synth, vgmi

pytestmark = pytest.mark.gpu


def _reads(n, seed, lo=30, hi=160):
    rng = np.random.default_rng(seed)
    cohort = get_cohort("cohort_snp")
    hap = cohort.haplotypes()[1]
    out = []
    for _ in range(n):
        ln = int(rng.integers(lo, hi))
        s = int(rng.integers(0, len(hap) - ln))
        r = bytearray(hap[s:s + ln].tobytes())
        if rng.random() < 0.1:
            r[int(rng.integers(0, ln))] = ord("N")
        if rng.random() < 0.05:
            r = bytearray(bytes(r).lower())
        out.append(bytes(r))
    return out


def _fastq(reads, qual_char=b"I", names=None, trailing_newline=True, line_end=b"\n"):
    parts = []
    for i, r in enumerate(reads):
        nm = names[i] if names else b"r%d extra comment" % i
        q = qual_char * len(r) if len(qual_char) == 1 else (qual_char * len(r))[:len(r)]
        parts.append(b"@" + nm + line_end + r + line_end + b"+" + line_end + q + line_end)
    t = b"".join(parts)
    return t if trailing_newline else t[:-len(line_end)]


def _cases():
    rd = _reads(400, 1)
    c = {}
    c["regular"] = _fastq(rd)
    c["no_trailing_newline"] = _fastq(rd, trailing_newline=False)
    c["quality_starts_with_at_plus_gt"] = _fastq(rd, qual_char=b"@+>I")
    c["names_with_at_and_tabs"] = _fastq(rd, names=[b"@@r%d\t@x +y" % i for i in range(len(rd))])
    c["crlf"] = _fastq(rd, line_end=b"\r\n")
    c["one_crlf_line_in_the_middle"] = _fastq(rd[:200]) + _fastq(rd[200:201], line_end=b"\r\n") + _fastq(rd[201:])
    wrapped = b"".join(b"@w%d\n" % i + r[:40] + b"\n" + r[40:] + b"\n+\n" + b"I" * 40 + b"\n" + b"I" * (len(r) - 40) + b"\n"
                       for i, r in enumerate(rd[:50]) if len(r) > 80)
    c["wrapped_lines_after_regular_ones"] = _fastq(rd[:100]) + wrapped + _fastq(rd[100:])
    c["fasta"] = b"".join(b">s%d\n" % i + r + b"\n" for i, r in enumerate(rd))
    c["fasta_after_fastq"] = _fastq(rd[:150]) + b"".join(b">s%d\n" % i + r + b"\n" for i, r in enumerate(rd[150:]))
    c["blank_lines_between_records"] = _fastq(rd[:100]) + b"\n\n" + _fastq(rd[100:200]) + b"\n" + _fastq(rd[200:])
    c["quality_too_short_stops_the_file"] = _fastq(rd[:120]) + b"@bad\n" + rd[120] + b"\n+\nIII\n" + _fastq(rd[121:])
    c["quality_too_long_stops_the_file"] = _fastq(rd[:120]) + b"@bad\n" + rd[120] + b"\n+\n" + b"I" * (len(rd[120]) + 3) + b"\n" + _fastq(rd[121:])
    c["truncated_inside_last_quality"] = _fastq(rd)[:-20]
    c["truncated_inside_last_sequence"] = _fastq(rd[:399]) + b"@last\n" + rd[399][:17]
    c["junk_before_first_record"] = b"# made by hand\n\n" + _fastq(rd)
    c["nul_byte_in_a_sequence"] = _fastq(rd[:77]) + b"@n\nACGTACGTACGTACGTACGTACGTACGTACGT\0ACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\n" + b"I" * 69 + b"\n" + _fastq(rd[77:])
    c["sequence_line_starting_with_gt"] = _fastq(rd[:30]) + b"@x\n>CGTACGT\n+\nIIIIIIII\n" + _fastq(rd[30:])
    c["single_record"] = _fastq(rd[:1])
    c["only_a_header"] = b"@lonely"
    return c


@pytest.fixture(scope="module")
def graph_ctx():
    cohort = get_cohort("cohort_snp")
    g = host.Graph(os.path.join(cohort.dir, "graph.bin.gz"))
    ctx = vgmi.Context(0, buffer_mib=16)
    g.upload(ctx)
    yield g, ctx, cohort
    ctx.close()
    g.close()


def _count(g, ctx, paths, host_parse, chunk_kb=None):
    old = {k: os.environ.get(k) for k in ("VGH_HOST_PARSE", "VGMI_FASTQ_CHUNK_KB")}
    os.environ["VGH_HOST_PARSE"] = "1" if host_parse else "0"
    if chunk_kb:
        os.environ["VGMI_FASTQ_CHUNK_KB"] = str(chunk_kb)
    else:
        os.environ.pop("VGMI_FASTQ_CHUNK_KB", None)
    try:
        cov, _node, _hist, st = g.sample_count(ctx, paths, threads=4, require_depth=False)
        return {"cov": cov, "read_base": st["read_base"], "n_reads": st["n_reads"]}
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("gz", [False, True], ids=["plain", "gzip"])
def test_named_pipe_input_equals_file_input(gz, graph_ctx, tmp_path):
    """The reference reads whatever gzopen opens -- a named pipe, `<(zcat x.gz)` -- in one pass (include/kseq.h:59-72,
    src/fastq_kmer.cpp:74-78).  A non-seekable input takes the host reader from its first byte; same counters as the file."""
    import gzip as _gz
    import threading
    g, ctx, cohort = graph_ctx
    text = _fastq(_reads(3000, 77))
    data = _gz.compress(text, 4) if gz else text
    p = tmp_path / ("x.fq.gz" if gz else "x.fq")
    p.write_bytes(data)
    want = _count(g, ctx, [str(p)], host_parse=False)
    fifo = str(tmp_path / "pipe.fq")
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write(data)
    t = threading.Thread(target=feed)
    t.start()
    try:
        got = _count(g, ctx, [fifo], host_parse=False)
    finally:
        t.join()
    assert want["n_reads"] == 3000 and got["n_reads"] == 3000
    assert np.array_equal(got["cov"], want["cov"]) and got["read_base"] == want["read_base"]


@pytest.mark.parametrize("name", sorted(_cases()))
def test_device_parser_equals_host_reader(name, graph_ctx, tmp_path):
    g, ctx, cohort = graph_ctx
    text = _cases()[name]
    p = tmp_path / "x.fq"
    p.write_bytes(text)
    try:
        want = _count(g, ctx, [str(p)], host_parse=True)
    except vgmi.VgmiError as e:   # the reference aborts on this input (empty sequence): so must the device path, with the same message
        for chunk_kb in (None, 4):
            with pytest.raises(vgmi.VgmiError) as e2:
                _count(g, ctx, [str(p)], host_parse=False, chunk_kb=chunk_kb)
            assert str(e2.value) == str(e)
        return
    # the host reader itself against the oracle on the same block
    block, _n, rb = host.fastx_read_all(str(p))
    t = o.Table(cohort.graph.keys)
    if len(block):
        t.count_block(block, cohort.k)
    assert np.array_equal(want["cov"], t.counts())
    assert want["read_base"] == rb
    for chunk_kb in (None, 4, 5, 64):
        got = _count(g, ctx, [str(p)], host_parse=False, chunk_kb=chunk_kb)
        assert np.array_equal(got["cov"], want["cov"]), (name, chunk_kb)
        assert got["read_base"] == want["read_base"], (name, chunk_kb)
        assert got["n_reads"] == want["n_reads"], (name, chunk_kb)


def test_device_parser_reports_where_it_stopped(graph_ctx):
    """The stream-level contract of vgmi_fastq_close: records and bases taken, the byte the host reader resumes at."""
    g, ctx, cohort = graph_ctx
    rd = _reads(50, 3)
    reg = _fastq(rd)
    ctx.counts_reset()
    r = ctx.fastq_text(reg, piece=1000)
    assert (r["n_records"], r["n_bases"], r["consumed"], r["stopped"], r["tail"]) == (50, sum(map(len, rd)), len(reg), False, b"")
    ctx.counts_reset()
    r = ctx.fastq_text(reg[:-1], piece=777)          # unterminated last line: that record is the host reader's
    last = len(_fastq(rd[:49]))
    assert (r["n_records"], r["consumed"], r["stopped"]) == (49, last, False) and r["tail"] == reg[last:-1]
    bad = _fastq(rd[:20]) + b">fasta\nACGT\n" + _fastq(rd[20:])
    ctx.counts_reset()
    r = ctx.fastq_text(bad, piece=4096)
    assert (r["n_records"], r["consumed"], r["stopped"], r["tail"]) == (20, len(_fastq(rd[:20])), True, b"")
    ctx.counts_reset()
    ctx.counts_finish()


def test_empty_sequence_is_an_error_on_both_paths(graph_ctx, tmp_path):
    g, ctx, _ = graph_ctx
    p = tmp_path / "e.fq"
    p.write_bytes(_fastq(_reads(10, 5)) + b"@e\n\n+\n\n" + _fastq(_reads(10, 6)))
    for hp in (True, False):
        with pytest.raises(Exception, match="empty read"):
            _count(g, ctx, [str(p)], host_parse=hp)


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf"])
def test_cohort_files_device_parser_all_containers(kind, graph_ctx, tmp_path):
    """The committed cohort FASTQs (gzip), re-packed as plain / gzip / block gzip: reference-dumped counters."""
    g, ctx, cohort = graph_ctx
    paths = []
    for m in (1, 2):
        raw = gzip.open(os.path.join(cohort.dir, f"reads_{m}.fq.gz"), "rb").read()
        if kind == "plain":
            p = tmp_path / f"r{m}.fq"
            p.write_bytes(raw)
        elif kind == "gzip":
            p = tmp_path / f"r{m}.fq.gz"
            with gzip.open(p, "wb", compresslevel=3) as f:
                f.write(raw)
        else:
            q = tmp_path / f"r{m}.fq"
            q.write_bytes(raw)
            p = tmp_path / f"r{m}.fq.bgz"
            synth.bgzf_compress_file(str(q), str(p))
        paths.append(str(p))
    for chunk_kb in (None, 16):
        got = _count(g, ctx, paths, host_parse=False, chunk_kb=chunk_kb)
        assert np.array_equal(got["cov"], cohort.ref_c_in_graph_order())
        assert got["read_base"] == cohort.ref_read_base


# ---- block gzip inflated on the device (csrc/vgmi_inflate.hip) -----------------------------------------------------
def _bgzf(tmp_path, name, text, level, block=0xff00):
    src = tmp_path / (name + ".fq")
    src.write_bytes(text)
    dst = tmp_path / (name + ".fq.gz")
    synth.bgzf_compress_file(str(src), str(dst), level=level, block=block)
    return dst


def _count_env(g, ctx, paths, env, chunk_kb=None):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _count(g, ctx, paths, host_parse=env.get("VGH_HOST_PARSE") == "1", chunk_kb=chunk_kb)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("level,block", [(0, 0xff00), (1, 0xff00), (6, 0xff00), (9, 0xff00), (4, 3000), (6, 70)])
def test_bgzf_inflated_on_the_device(level, block, graph_ctx, tmp_path):
    """Stored, fixed and dynamic DEFLATE blocks, full-size and tiny members: counters equal the all-host path
    (inflate workers + host parser) and the oracle; mixed qualities make the streams literal/match mixed."""
    g, ctx, cohort = graph_ctx
    rd = _reads(3000 if block > 100 else 60, 21 + level)
    rng = np.random.default_rng(level)
    quals = [bytes(rng.integers(33, 74, size=len(r), dtype=np.uint8)) for r in rd]
    text = b"".join(b"@q%d\n" % i + r + b"\n+\n" + q + b"\n" for i, (r, q) in enumerate(zip(rd, quals)))
    p = _bgzf(tmp_path, "a", text, level, block)
    want = _count_env(g, ctx, [str(p)], {"VGH_HOST_PARSE": "1"})
    t = o.Table(cohort.graph.keys)
    t.count_block(np.frombuffer(b"".join(r + b"\n" for r in rd), dtype=np.uint8), cohort.k)
    assert np.array_equal(want["cov"], t.counts())
    for chunk_kb in (None, 200, 96):
        got = _count_env(g, ctx, [str(p)], {"VGH_HOST_PARSE": "0"}, chunk_kb=chunk_kb)
        assert np.array_equal(got["cov"], want["cov"]) and got["read_base"] == want["read_base"] and got["n_reads"] == want["n_reads"]
    host_inflate = _count_env(g, ctx, [str(p)], {"VGH_HOST_PARSE": "0", "VGH_HOST_INFLATE": "1"})
    assert np.array_equal(host_inflate["cov"], want["cov"])
    # commits cut to whole "rounds" of wavefronts (a round of 3 072 members on the device; of 5 and 16 here): what a commit leaves
    # comes again with the next bytes
    for rnd in ("5", "16"):
        for chunk_kb in (96, 200):
            got = _count_env(g, ctx, [str(p)], {"VGH_HOST_PARSE": "0", "VGMI_BGZF_ROUND_MEMBERS": rnd}, chunk_kb=chunk_kb)
            assert np.array_equal(got["cov"], want["cov"]) and got["read_base"] == want["read_base"] and got["n_reads"] == want["n_reads"]


@pytest.mark.parametrize("damage", ["flip_in_third_member", "truncated_mid_member", "gzip_member_after_bgzf", "crc_field", "no_eof_marker",
                                    "fastq_text_behind_last_member", "short_garbage_behind_last_member"])
def test_bgzf_damage_and_mixtures_equal_the_host_decoder(damage, graph_ctx, tmp_path):
    """What the device cannot vouch for goes to the host decoder at the right byte: same counters as the all-host path
    (which delivers what decoded cleanly before the damage, like the reference's gzread loop)."""
    g, ctx, cohort = graph_ctx
    rd = _reads(2500, 77)
    text = _fastq(rd)
    p = _bgzf(tmp_path, "d", text, 5)
    raw = bytearray(p.read_bytes())
    # member boundaries
    offs, pos = [], 0
    while pos < len(raw):
        offs.append(pos)
        pos += (raw[pos + 16] | raw[pos + 17] << 8) + 1
    assert len(offs) >= 5
    if damage == "flip_in_third_member":
        raw[offs[2] + 200] ^= 0x5A
    elif damage == "truncated_mid_member":
        raw = raw[: offs[3] + 1000]
    elif damage == "gzip_member_after_bgzf":
        raw = raw[: offs[3]] + gzip.compress(text[-30000:], 6)
    elif damage == "crc_field":
        raw[offs[2] - 8] ^= 1
    elif damage == "no_eof_marker":
        raw = raw[: offs[-1]]
    elif damage == "fastq_text_behind_last_member":
        # bytes behind the last member that are no gzip header: gzread (and the host block-gzip source) ignore them; handed
        # to a parser as text they would be a read (ByteSource::open_at went transparent at an offset)
        raw = raw + b"\n@x\n" + rd[0][:60] + b"\n+\n" + b"I" * 60 + b"\n"
    elif damage == "short_garbage_behind_last_member":
        raw = raw + b"@x\nACGT\n+\nII"
    q = tmp_path / "dmg.fq.gz"
    q.write_bytes(bytes(raw))
    want = _count_env(g, ctx, [str(q)], {"VGH_HOST_PARSE": "1"})
    for chunk_kb in (None, 160):
        got = _count_env(g, ctx, [str(q)], {"VGH_HOST_PARSE": "0"}, chunk_kb=chunk_kb)
        assert np.array_equal(got["cov"], want["cov"]), damage
        assert got["read_base"] == want["read_base"] and got["n_reads"] == want["n_reads"], damage
    assert want["n_reads"] > 0


def test_bgzf_stream_contract(graph_ctx, tmp_path):
    """vgmi_fastq_commit_bgzf / _bgzf_status directly: a clean file is inflated and parsed entirely on the device (nothing
    left for the host decoder); a damaged member is named by its compressed offset and nothing behind it is parsed."""
    g, ctx, cohort = graph_ctx
    rd = _reads(2000, 5)
    text = _fastq(rd)
    p = _bgzf(tmp_path, "c", text, 6)
    comp = p.read_bytes()
    for piece in (None, 150_000):
        ctx.counts_reset()
        r = ctx.fastq_bgzf(comp, piece=piece)
        assert not r["inflate_failed"] and not r["stopped"]
        assert r["taken"] == r["good_compressed_bytes"] == len(comp)
        assert (r["n_records"], r["n_bases"], r["consumed"], r["tail"]) == (2000, sum(map(len, rd)), len(text), b"")
        got, _, _ = ctx.counts_finish()
        t = o.Table(cohort.graph.keys)
        t.count_block(np.frombuffer(b"".join(x + b"\n" for x in rd), dtype=np.uint8), cohort.k)
        assert np.array_equal(got, t.counts())
    offs, pos = [], 0
    while pos < len(comp):
        offs.append(pos)
        pos += (comp[pos + 16] | comp[pos + 17] << 8) + 1
    bad = bytearray(comp)
    bad[offs[3] + 300] ^= 0xFF
    ctx.counts_reset()
    r = ctx.fastq_bgzf(bytes(bad))
    assert r["inflate_failed"] and r["good_compressed_bytes"] == offs[3] and r["reason"] in (1, 2, 3, 4, 5)
    assert r["consumed"] + len(r["tail"]) == 3 * 0xff00      # the text of the three members in front of it, no more
    ctx.counts_finish()


@pytest.mark.parametrize("damage", ["clean", "truncated", "flipped_bit", "two_members_and_garbage", "bad_crc_then_a_member", "sixty_small_members"])
def test_gzip_files_through_several_inflate_threads(damage, graph_ctx, tmp_path, monkeypatch):
    """An ordinary gzip FASTQ file read with threads to spare is inflated by several of them (par_gunzip.cpp: small spans here,
    so that the file has dozens of seams).  Counters, read count and base count must be those of the one-thread decoder --
    also where a damaged stream ends -- on the device-parser path and on the host-parser path."""
    import gzip
    g, ctx, cohort = graph_ctx
    rng = np.random.default_rng(5)
    haps = cohort.haplotypes()
    reads = []
    for i in range(30_000):
        h = haps[i % len(haps)]
        p = int(rng.integers(0, h.size - 150))
        reads.append(h[p:p + 150].tobytes())
    text = b"".join(b"@A00:1:%d 1:N:0\n%s\n+\n%s\n" % (i, r, b"F" * 150) for i, r in enumerate(reads))
    if damage == "two_members_and_garbage":
        comp = gzip.compress(text[:4_000_000], 4) + gzip.compress(text[4_000_000:], 6) + b"\0garbage behind the last member"
    elif damage == "bad_crc_then_a_member":
        # zlib hands the first member's text over and reports its CRC at the end: the data ends there, the second member is never read
        cut = text.index(b"@A00:1:20000 ")
        first = bytearray(gzip.compress(text[:cut], 4))
        first[-7] ^= 0x20
        comp = bytes(first) + gzip.compress(text[cut:], 6)
    elif damage == "sixty_small_members":
        step = len(text) // 60
        cuts = [0] + [text.index(b"\n@A00:1:", i * step) + 1 for i in range(1, 60)] + [len(text)]
        comp = b"".join(gzip.compress(text[a:b], 5) for a, b in zip(cuts[:-1], cuts[1:]))
    else:
        comp = gzip.compress(text, 4)
    if damage == "truncated":
        comp = comp[:len(comp) * 2 // 3]
    elif damage == "flipped_bit":
        b = bytearray(comp)
        b[len(b) // 2] ^= 0x10
        comp = bytes(b)
    path = tmp_path / ("reads_%s.fq.gz" % damage)
    path.write_bytes(comp)
    monkeypatch.setenv("VGH_PARGZ_SPAN_KB", "64")
    got = {}
    for par in ("0", "1"):
        monkeypatch.setenv("VGH_PAR_GUNZIP", par)
        for host_parse in (False, True):
            got[(par, host_parse)] = _count(g, ctx, [str(path)], host_parse)
    ref = got[("0", True)]
    assert ref["n_reads"] > 5_000 and int(ref["cov"].astype(np.int64).sum()) > 0
    if damage in ("clean", "sixty_small_members"):
        assert ref["n_reads"] == 30_000
    if damage == "bad_crc_then_a_member":
        assert ref["n_reads"] == 20_000
    for key, r in got.items():
        assert r["n_reads"] == ref["n_reads"] and r["read_base"] == ref["read_base"], key
        assert np.array_equal(r["cov"], ref["cov"]), key
