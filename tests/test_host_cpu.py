"""CPU tests of the C++ host side (libvghost.so) and of the C-ABI surface."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import graphbin_py
from conftest import GOLDEN, ROOT, block_from_seqs, get_cohort, read_fastq_seqs
from varigraph_amd import host, vgmi


def test_cabi_exports_every_declared_symbol():
    """libvgmi.so / libvghost.so load and export every function include/*.h declares."""
    for header, libpath in (("vgmi.h", vgmi.LIB_PATH), ("vghost.h", host.LIB_PATH)):
        txt = open(os.path.join(ROOT, "include", header)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        declared = set(re.findall(r"\b(vg[mh]i?_[a-z0-9_]+)\s*\(", txt))
        assert len(declared) >= 10
        vgmi.lib(); host.lib()
        out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
        exported = set(re.findall(r" T (vg[mh]i?_[a-z0-9_]+)", out))
        missing = declared - exported
        assert not missing, f"{header}: not exported: {sorted(missing)}"
    assert set(vgmi.SYMBOLS) <= exported | set(re.findall(r" T (vgmi_[a-z0-9_]+)", subprocess.run(
        ["nm", "-D", "--defined-only", vgmi.LIB_PATH], capture_output=True, text=True).stdout))


def test_no_gpu_fails_loudly():
    """Without a device the product path must refuse to run (no CPU fallback)."""
    if vgmi.lib().vgmi_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vgmi.VgmiError) as e:
        vgmi.Context(0)
    assert e.value.code == vgmi.E_NO_DEVICE


def test_bloom_params_match_reference_kats():
    import json
    for c in json.load(open(os.path.join(GOLDEN, "kats.json")))["bloom_size"]:
        assert vgmi.bloom_params(c["n"], c["p"]) == (c["m"], c["n_hash"])


def test_graph_loader_matches_python_parser(cohort):
    g = cohort.graph
    a = host.load_graph(os.path.join(cohort.dir, "graph.bin.gz"))
    assert (a["k"], a["vcf_ploidy"], a["hap_num"], a["genome_size"]) == (g.k, g.vcf_ploidy, g.hap_num, g.genome_size)
    assert np.array_equal(a["keys"], g.keys)
    assert np.array_equal(a["f"], g.f)
    assert np.array_equal(a["bitvec"], g.bitvec)


def test_graph2node_matches_reference_dump(cohort):
    """Node order and per-node k-mer order (incl. the >128 std::sort-by-frequency truncation)
    equal what the reference holds after ConstructIndex::graph2node."""
    a = host.load_graph(os.path.join(cohort.dir, "graph.bin.gz"))
    ref_nodes = cohort.ref_nodes
    assert len(ref_nodes) == len(a["node_start"])
    n_big = 0
    for i, (name, start, kh) in enumerate(ref_nodes):
        assert a["chr_names"][a["node_chr"][i]] == name
        assert a["node_start"][i] == start
        lo, hi = int(a["node_off"][i]), int(a["node_off"][i + 1])
        assert np.array_equal(a["keys"][a["node_key_index"][lo:hi]], kh), (name, start)
        n_big += (hi - lo) == 128
    if cohort.meta["name"] == "cohort_sv":
        assert n_big > 0, "the SV cohort must exercise the >128 k-mer branch"


def test_hom_flags_match_oracle_histogram(cohort):
    """flag & (c != 0) histogram == reference get_hom_kmer histogram."""
    a = host.load_graph(os.path.join(cohort.dir, "graph.bin.gz"))
    c = cohort.ref_c_in_graph_order()
    sel = (a["hom_flag"] != 0) & (c != 0)
    hist = np.bincount(c[sel], minlength=256)
    assert hist.tolist() == cohort.meta["hist"]


def test_fastx_reader_matches_fixture(cohort):
    fq = [os.path.join(cohort.dir, f"reads_{m}.fq.gz") for m in (1, 2)]
    if not os.path.exists(fq[0]):
        pytest.skip("reads are regenerated for this cohort")
    tot = 0
    for p in fq:
        block, n, rb = host.fastx_read_all(p)
        seqs = read_fastq_seqs(p)
        assert n == len(seqs)
        assert np.array_equal(block, block_from_seqs(seqs))
        tot += rb
    assert tot == cohort.ref_read_base


def test_fastx_reader_edge_cases(tmp_path):
    """kseq_read semantics: FASTA, multi-line, CRLF, truncated quality stops the file, junk before
    the first header, '@' inside quality."""
    def rd(data, gz=False):
        p = tmp_path / ("x.fq.gz" if gz else "x.fq")
        (gzip.open if gz else open)(p, "wb").write(data)
        b, n, rb = host.fastx_read_all(str(p))
        return bytes(b), n, rb
    assert rd(b">a desc\nACGT\nTTGA\n>b\nGG\n") == (b"ACGTTTGA\nGG\n", 2, 10)
    assert rd(b"@r1\nACGT\n+\nIIII\n@r2\nTTTT\n+r2\nIIII\n", gz=True) == (b"ACGT\nTTTT\n", 2, 8)
    assert rd(b"junk\n@r1\r\nACGT\r\n+\r\nIIII\r\n") == (b"ACGT\n", 1, 4)
    # quality starting with '@' must not be taken for a header
    assert rd(b"@r1\nACGT\n+\n@III\n@r2\nGGCC\n+\nIIII\n") == (b"ACGT\nGGCC\n", 2, 8)
    # truncated last record (-2): the reference stops there and keeps what it had
    assert rd(b"@r1\nACGT\n+\nIIII\n@r2\nGGCC\n+\nII\n") == (b"ACGT\n", 1, 4)
    assert rd(b"@r1\nACGT\n+\nIIII\n@r2\nGGCC\n") == (b"ACGT\nGGCC\n", 2, 8)   # FASTA-style tail is accepted by kseq
    assert rd(b"") == (b"", 0, 0)
    # multi-line FASTQ
    assert rd(b"@r1\nAC\nGT\n+\nII\nII\n") == (b"ACGT\n", 1, 4)


def test_hash64_inverse_roundtrip(tmp_path):
    """vg_hash64_inv (product, used by the table build) inverts the reference hash64 for every k."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdint>
#include "vgmi_device.h"
int main() {
    uint64_t x = 88172645463325252ULL;
    for (int k = 1; k <= 28; ++k) {
        uint64_t mask = (1ULL << (2 * k)) - 1;
        for (int i = 0; i < 20000; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint64_t v = x & mask;
            if (vg_hash64_inv(vg_hash64(v, mask), mask) != v) { printf("FAIL k=%d v=%llx\n", k, (unsigned long long)v); return 1; }
            if (vg_hash64(vg_hash64_inv(v, mask), mask) != v) { printf("FAIL2 k=%d\n", k); return 1; }
        }
    }
    // KAT from the reference: ACGTACGTTGCAAGCTTAGCGATCGAT, k=27 -> 0x2df5c044b3f1eb1b
    printf("%llx\n", (unsigned long long)vg_hash64_inv(0x2df5c044b3f1eb1bULL >> 8, (1ULL << 54) - 1));
    return 0;
}''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "varigraph_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    # canonical 2-bit value of the k-mer (min of forward / reverse complement)
    seq = "ACGTACGTTGCAAGCTTAGCGATCGAT"
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    fwd = 0
    for ch in seq:
        fwd = (fwd << 2) | code[ch]
    rc = 0
    for ch in reversed(seq):
        rc = (rc << 2) | (3 - code[ch])
    assert int(out.stdout.strip(), 16) == min(fwd, rc)


def test_synth_generator_is_stable():
    """The seeded generator must keep producing the block the c1 fixture was made from."""
    c = get_cohort("c1")
    c.regenerate_block()  # asserts the md5


def _mbf_fasta(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mg", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    fa = str(tmp_path / "mbf.fa")
    mg.mbf_fasta(fa)
    return fa


def test_reference_bloom_seed_draw():
    """BloomFilter::_init_seeds restated with explicit entropy == seeds of the det reference build."""
    import json
    for c in json.load(open(os.path.join(GOLDEN, "mbf.json"))):
        got = host.bloom_reference_seeds(c["random_device_value"], c["n_hash"])
        assert [f"{int(x):x}" for x in got] == c["seeds"]


def test_oracle_whole_reference_bloom_matches_make_mbf(tmp_path):
    """oracle Bloom over the FASTA == the reference's build_fasta_index + make_mbf (sha256 of the filter)."""
    import hashlib
    import json
    import oracle_lib as o
    fa = _mbf_fasta(tmp_path)
    recs = {}
    name = None
    size = 0
    for ln in open(fa, "rb").read().split(b"\n"):
        if ln.startswith(b">"):
            name = ln[1:].split()[0]
            if name in recs:
                name = (name, "dup")          # emplace keeps the first sequence of a name
            recs[name] = b""
        elif name is not None:
            recs[name] += ln
            size += len(ln)
    for c in json.load(open(os.path.join(GOLDEN, "mbf.json"))):
        assert size == c["genome_size"]
        filt = np.zeros(c["m"], dtype=np.uint8)
        seeds = np.array([int(s, 16) for s in c["seeds"]], dtype=np.uint64)
        for nm, seq in recs.items():
            if isinstance(nm, tuple):
                continue
            o.bloom_add_seq(filt, seeds, np.frombuffer(seq, dtype=np.uint8), c["k"])
        assert hashlib.sha256(filt.tobytes()).hexdigest() == c["sha256"]
        assert int(filt.sum()) == c["sum"] and int((filt != 0).sum()) == c["nonzero"]


def test_hand_scheduled_registers_untouched_by_compiler():
    """The k = 27 kernels keep in-flight vector-memory data in fixed VGPRs (v120..v127) that the compiler must not
    know about: check on the generated gfx950 ISA that only the hand-written instructions touch them,
    that the compiler's own allocation stays below, and that the kernels have no scratch traffic."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "check_hot_vgprs.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # count27s_kernel (12-mer grid: path table / hash table), count27_kernel: LDS filter + compact, global filter + compact / 16-byte slots
    assert r.stdout.count("0 scratch accesses") == 12, r.stdout


def _bgzf_bytes(data, block=0xff00, level=6, eof_marker=True):
    """What bgzip writes: gzip members of <= 64 KiB with the 'BC' extra field (SAM spec 4.1)."""
    import struct
    import zlib
    out = []
    for o in range(0, len(data), block):
        d = data[o:o + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = c.compress(d) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(cd) + 25) + cd +
                   struct.pack("<II", zlib.crc32(d), len(d)))
    if eof_marker:
        out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return out


def test_byte_sources_deliver_what_gzread_would(tmp_path):
    """Ingest row: plain / gzip stream / block gzip inputs give the same records; concatenated members are
    followed, trailing garbage is ignored, a damaged or truncated stream ends the file where gzread would return -1."""
    rng = np.random.default_rng(7)
    n = 40000
    lens = rng.integers(30, 151, size=n)
    ends = np.cumsum(lens)
    pool = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(ends[-1]))].tobytes()
    seqs = [pool[e - l:e] for e, l in zip(ends, lens)]
    fq = b"".join(b"@r%d x\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    want = block_from_seqs(seqs)

    def rd(name, payload, threads=1):
        p = tmp_path / name
        p.write_bytes(payload)
        b, cnt, rb, kind = host.fastx_read_all(str(p), decode_threads=threads, with_kind=True)
        return bytes(b), cnt, kind

    assert rd("a.fq", fq) == (want.tobytes(), n, "plain")
    assert rd("a.fq.gz", gzip.compress(fq)) == (want.tobytes(), n, "gzip")
    # concatenated members, the boundary inside a record; then bytes that are no gzip header
    cut = len(fq) // 2 + 7
    two = gzip.compress(fq[:cut]) + gzip.compress(fq[cut:])
    assert rd("two.fq.gz", two) == (want.tobytes(), n, "gzip")
    assert rd("junk.fq.gz", two + b"not gzip at all") == (want.tobytes(), n, "gzip")
    assert rd("junk1.fq.gz", two + b"\x1f") == (want.tobytes(), n, "gzip")
    # every block type and header field the decoder (csrc/host/fast_inflate.cpp) has to handle: stored, fixed and dynamic
    # Huffman blocks, literal-only and run-length streams, sync/full flush points, FEXTRA/FNAME/FCOMMENT/FHCRC
    import struct
    import zlib

    def member(d, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, fields=False):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        if flush_every:
            body = b"".join(c.compress(d[i:i + flush_every]) + c.flush(zlib.Z_SYNC_FLUSH if (i // flush_every) % 3 else zlib.Z_FULL_FLUSH)
                            for i in range(0, len(d), flush_every)) + c.flush()
        else:
            body = c.compress(d) + c.flush()
        h = b"\x1f\x8b\x08" + bytes([0x1E if fields else 0]) + b"\0\0\0\0\0\xff"
        if fields:
            h += struct.pack("<H", 7) + b"ab\x03\x00xyz" + b"reads.fq\0" + b"a comment\0"
            h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
        return h + body + struct.pack("<II", zlib.crc32(d), len(d) & 0xFFFFFFFF)

    part = fq[: len(fq) // 8]
    part = part[: part.rfind(b"\n@r") + 1]
    (tmp_path / "part.fq").write_bytes(part)
    ref_part = bytes(host.fastx_read_all(str(tmp_path / "part.fq"))[0])
    for name, payload in (("l0", member(part, 0)), ("l1", member(part, 1)), ("l9", member(part, 9)),
                          ("fixed", member(part, 6, zlib.Z_FIXED)), ("huff", member(part, 6, zlib.Z_HUFFMAN_ONLY)),
                          ("rle", member(part, 6, zlib.Z_RLE)), ("flush", member(part, 6, flush_every=777)),
                          ("fields", member(part, 6, fields=True)),
                          ("chain", member(part[:5000], 0) + member(b"") + member(part[5000:], 9, fields=True))):
        got, _, kind = rd(name + ".fq.gz", payload)
        assert kind == "gzip" and got == ref_part, name
    # a flipped bit in the trailer CRC: all data arrives (the reference's gzread also hands it over before it fails)
    bad_crc = bytearray(member(part, 6))
    bad_crc[-6] ^= 0x40
    assert rd("badcrc.fq.gz", bytes(bad_crc))[0] == ref_part
    # truncated stream: the records before the cut still arrive, nothing after
    tb, tcnt, _ = rd("trunc.fq.gz", gzip.compress(fq)[: len(gzip.compress(fq)) // 2])
    assert 0 < tcnt < n and want.tobytes().startswith(tb[: tb.rfind(b"\n", 0, len(tb) - 1) + 1][:1000])
    # block gzip, 1..8 inflate workers, with and without the EOF marker block
    blocks = _bgzf_bytes(fq)
    assert len(blocks) > 100   # several worker tasks
    for t in (1, 2, 3, 8):
        assert rd("b.fq.gz", b"".join(blocks), t) == (want.tobytes(), n, "bgzf")
    assert rd("b2.fq.gz", b"".join(_bgzf_bytes(fq, block=1000, eof_marker=False)), 4) == (want.tobytes(), n, "bgzf")
    # an ordinary gzip member after some BGZF blocks (cat a.gz b.gz): the stream decoder takes over
    k = len(blocks) // 2
    part = sum(0xff00 for _ in range(k))
    mixed = b"".join(blocks[:k]) + gzip.compress(fq[part:])
    assert rd("mixed.fq.gz", mixed, 4) == (want.tobytes(), n, "bgzf")
    # a block with a wrong CRC: everything before it, nothing after it
    bad = bytearray(blocks[5])
    bad[-8] ^= 0xFF
    got, cnt, _ = rd("bad.fq.gz", b"".join(blocks[:5]) + bytes(bad) + b"".join(blocks[6:]), 4)
    ref_b, ref_cnt, _ = rd("bad_ref.fq", fq[: 5 * 0xff00])
    # the plain prefix may end in a partial record that kseq still accepts FASTA-style; compare complete records
    assert cnt in (ref_cnt, ref_cnt - 1) and ref_b.startswith(got[: got.rfind(b"\n", 0, len(got) - 1) + 1])
    # BGZF file cut in the middle of a block
    cutb = b"".join(blocks)[: sum(len(b) for b in blocks[:7]) + 100]
    got, cnt, _ = rd("cutb.fq.gz", cutb, 4)
    assert 0 < cnt < n and want.tobytes().startswith(got[: got.rfind(b"\n", 0, len(got) - 1) + 1])
    with pytest.raises(RuntimeError, match="No such file"):
        host.fastx_read_all(str(tmp_path / "missing.fq.gz"))


def test_stl_order_map_iterates_like_unordered_map(tmp_path):
    """csrc/host/stl_order_map.hpp fixes graph.bin's k-mer record order: it must iterate exactly like libstdc++'s
    std::unordered_map<uint64_t, ...> after the same inserts (tests/native/order_map_check.cpp)."""
    import subprocess
    exe = str(tmp_path / "order_map_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "varigraph_amd", "csrc", "host"),
                        os.path.join(ROOT, "tests", "native", "order_map_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "order identical" in r.stdout, r.stdout[-2000:]


def test_graph2node_through_the_batched_lookup_equals_the_host_index(tmp_path):
    """GraphIndex::graph2node (src/construct_index.cpp:710-751, 1572-1603) with its lookups served as ONE batch -- the hook the CLI
    points at vgmi_table_lookup -- against its own host index, under AddressSanitizer + UBSan: the same node lists, including the
    nodes that keep the 128 rarest of more than 128 k-mers (cohort_sv) and a hook that declines (the host index takes over)."""
    import gzip
    import subprocess
    host_dir = os.path.join(ROOT, "varigraph_amd", "csrc", "host")
    exe = str(tmp_path / "graph2node_check")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I", host_dir,
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "graph2node_check.cpp"),
                        os.path.join(host_dir, "graph_index.cpp"), "-L", os.path.join(ROOT, "varigraph_amd"), "-lvgmi",
                        "-Wl,-rpath," + os.path.join(ROOT, "varigraph_amd"), "-lz", "-lpthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    truncated = 0
    for cohort in ("c1", "cohort_snp", "cohort_sv", "cohort_tetra", "cohort_k22"):
        g = tmp_path / "graph.bin"
        g.write_bytes(gzip.open(os.path.join(GOLDEN, cohort, "graph.bin.gz"), "rb").read())
        r = subprocess.run([exe, str(g), "3"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("identical"), (cohort, r.stdout[-500:], r.stderr[-1500:])
        truncated += int(r.stdout.split(" nodes of 128 entries")[0].split()[-1])
    assert truncated > 0


def test_vcf_number_formatting_equals_the_stream(tmp_path):
    """csrc/host/fixed1.hpp writes GQ / GPP / CAK without an ostringstream: the same characters as `<< std::fixed << std::setprecision(1)`
    for six million floats (every tie and near-tie of the first decimals, every exponent, denormals, infinities, NaN)."""
    import subprocess
    exe = str(tmp_path / "fixed1_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "varigraph_amd", "csrc", "host"),
                        os.path.join(ROOT, "tests", "native", "fixed1_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith(" 0 different"), r.stdout[-2000:]


def test_vcf_writer_emits_block_gzip_independent_of_thread_count(tmp_path):
    """The CLI's output writer: valid BGZF (every member carries its size in the 'BC' field, EOF marker at the end), the
    content is the text, and the bytes do not depend on how many workers deflated the blocks."""
    import struct
    rng = np.random.default_rng(9)
    lines = [b"chr1\t%d\t.\tA\tC\t.\tPASS\t.\tGT:GQ:GPP:NAK:CAK:UK\t0/1:%d.0:0.99:%d,%d:5.5,4.2:%d\n" % (i * 37, rng.integers(0, 99), rng.integers(1, 60),
             rng.integers(1, 60), rng.integers(0, 50)) for i in range(60000)]
    for text in (b"".join(lines), b"", b"x", b"".join(lines)[:0xff00], b"".join(lines)[:0xff01]):
        outs = []
        for threads in (1, 3, 8):
            p = tmp_path / f"o{threads}.vcf.gz"
            host.write_vcf_gz(str(p), text, threads)
            outs.append(p.read_bytes())
        assert outs[0] == outs[1] == outs[2]
        assert gzip.decompress(outs[0]) == text
        # walk the members through their BSIZE fields
        raw, o, total = outs[0], 0, 0
        while o < len(raw):
            assert raw[o:o + 4] == b"\x1f\x8b\x08\x04" and raw[o + 12:o + 16] == b"BC\x02\x00"
            bsize = struct.unpack("<H", raw[o + 16:o + 18])[0] + 1
            isize = struct.unpack("<I", raw[o + bsize - 4:o + bsize])[0]
            assert isize <= 0xff00
            total += isize
            o += bsize
        assert o == len(raw) and total == len(text)
        assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


@pytest.mark.parametrize("name", ["cohort_snp", "cohort_sv"])
def test_reads_index_dump_is_the_reference_file(name, tmp_path):
    """vgh_reads_index_save writes the bytes FastqKmer::save_index writes for the same counters (sha256 of the reference's own
    file, tests/golden/make_reads_index_golden.py); vgh_reads_index_load reads them back, also with the records reversed (the
    reference loads them into a map), and refuses a k-mer the graph does not hold."""
    import hashlib
    import json
    from conftest import get_cohort
    cohort = get_cohort(name)
    want = json.load(open(os.path.join(os.path.dirname(cohort.dir), "reads_index.json")))[name]
    g = host.Graph(os.path.join(cohort.dir, "graph.bin.gz"))
    try:
        cov = cohort.ref_c_in_graph_order()
        path = str(tmp_path / "reads.idx")
        g.reads_index_save(path, cov, want["read_base"])
        data = open(path, "rb").read()
        assert len(data) == want["bytes"] and hashlib.sha256(data).hexdigest() == want["sha256"]
        back, rb = g.reads_index_load(path)
        assert rb == want["read_base"] and np.array_equal(back, cov)
        rec = (len(data) - 8) // cov.size
        records = [data[8 + i * rec:8 + (i + 1) * rec] for i in range(cov.size)]
        open(path, "wb").write(data[:8] + b"".join(reversed(records)))
        back, rb = g.reads_index_load(path)
        assert np.array_equal(back, cov)
        open(path, "wb").write(data[:8] + b"\x01" * 8 + records[0][8:])
        with pytest.raises(RuntimeError, match="not in the graph"):
            g.reads_index_load(path)
    finally:
        g.close()
