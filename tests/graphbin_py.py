"""Independent (pure python) parser of the reference's graph.bin, for tests only.
Layout: SURVEY.md Appendix A; writer src/construct_index.cpp:760-902, reader :911-1105."""
import gzip
import struct

import numpy as np


class GraphBin:
    pass


def load(path):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        b = f.read()
    o = 0

    def rd(fmt):
        nonlocal o
        v = struct.unpack_from("<" + fmt, b, o)
        o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal o
        n = rd("I")
        s = b[o:o + n]
        o += n
        return s

    g = GraphBin()
    g.graph_base_num = rd("Q")
    g.k = rd("I")
    g.vcf_ploidy = rd("I")
    g.vcf_head = rstr()
    g.chr_len = {}
    g.genome_size = 0
    g.vcf_info = {}
    for _ in range(rd("I")):
        name = rstr().decode()
        ln = rd("I")
        g.chr_len[name] = ln
        g.genome_size += ln
        sites = {}
        for _ in range(rd("I")):
            start = rd("I")
            sites[start] = [rstr() for _ in range(rd("I"))]
        g.vcf_info[name] = sites
    g.hap_num = rd("H")
    g.hap_names = {}
    for _ in range(g.hap_num):
        idx = rd("H")
        g.hap_names[idx] = rstr().decode()
    g.nodes = {}  # chr -> list of (start, seqs, gts, kmer hashes)
    for _ in range(rd("I")):
        name = rstr().decode()
        lst = []
        for _ in range(rd("I")):
            start = rd("I")
            seqs = [rstr() for _ in range(rd("I"))]
            ngt = rd("I")
            gts = np.frombuffer(b, dtype="<u2", count=ngt, offset=o).copy()
            o += 2 * ngt
            nk = rd("I")
            kh = np.frombuffer(b, dtype="<u8", count=nk, offset=o).copy()
            o += 8 * nk
            lst.append((start, seqs, gts, kh))
        g.nodes[name] = lst
    g.read_base = rd("Q")
    keys, cs, fs, bvs = [], [], [], []
    while o < len(b):
        key, c, f_, bl = struct.unpack_from("<QBBQ", b, o)
        o += 18
        keys.append(key); cs.append(c); fs.append(f_)
        bvs.append(np.frombuffer(b, dtype=np.int8, count=bl, offset=o))
        o += bl
    g.keys = np.array(keys, dtype=np.uint64)
    g.c = np.array(cs, dtype=np.uint8)
    g.f = np.array(fs, dtype=np.uint8)
    g.bitlen = len(bvs[0]) if bvs else 0
    g.bitvec = np.stack(bvs).astype(np.int8) if bvs else np.zeros((0, 0), dtype=np.int8)
    return g


def variant_nodes(g):
    """(chr, start, kmerHash[]) of every variant node in map order (chr lexicographic, start ascending)."""
    out = []
    for name in sorted(g.nodes):
        for start, seqs, gts, kh in sorted(g.nodes[name], key=lambda t: t[0]):
            if len(gts) == 1:
                continue
            out.append((name, start, kh))
    return out


def load_nodes_dump_gz(path):
    """<out>.nodes written by oracle/ref_harness.cpp `sample`: node k-mer lists after graph2node."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        b = f.read()
    o = 0
    out = []
    while o < len(b):
        n = struct.unpack_from("<I", b, o)[0]; o += 4
        name = b[o:o + n].decode(); o += n
        start, cnt = struct.unpack_from("<II", b, o); o += 8
        kh = np.frombuffer(b, dtype="<u8", count=cnt, offset=o).copy(); o += 8 * cnt
        out.append((name, start, kh))
    return out


def load_counts_dump(path):
    """OUT of ref_harness count/sample: readBase, genomeSize, n, n*(key,c,f)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        b = f.read()
    read_base, genome_size, n = struct.unpack_from("<QQQ", b, 0)
    rec = np.frombuffer(b, dtype=np.dtype([("key", "<u8"), ("c", "u1"), ("f", "u1")]), count=n, offset=24)
    return read_base, genome_size, rec["key"].copy(), rec["c"].copy(), rec["f"].copy()
