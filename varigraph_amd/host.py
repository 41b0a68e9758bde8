"""ctypes binding of include/vghost.h (libvghost.so): graph.bin reader, FASTA/Q reader and the
sample-counting driver (C++17 host side).  No fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

from . import vgmi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvghost.so")
_lib = None


class GraphInfo(C.Structure):
    _fields_ = [("graph_base_num", C.c_uint64), ("genome_size", C.c_uint64), ("n_keys", C.c_uint64),
                ("bitlen", C.c_uint64), ("n_variant_nodes", C.c_uint64), ("n_node_entries", C.c_uint64),
                ("k", C.c_uint32), ("vcf_ploidy", C.c_uint32), ("hap_num", C.c_uint32), ("n_chromosomes", C.c_uint32)]


class SampleStats(C.Structure):
    _fields_ = [("read_base", C.c_uint64), ("n_reads", C.c_uint64), ("read_depth", C.c_float),
                ("hap_kmer_coverage", C.c_float), ("max_coverage", C.c_uint8), ("hom_coverage", C.c_uint8),
                ("seconds_total", C.c_double), ("seconds_kernel", C.c_double)]


class GenotypeConfig(C.Structure):
    _fields_ = [("sample_type", C.c_char_p), ("sample_ploidy", C.c_uint32), ("haploid_num", C.c_uint32),
                ("chr_len_thread", C.c_uint32), ("transition", C.c_char_p), ("sv_only", C.c_int), ("threads", C.c_uint32),
                ("min_gq", C.c_float)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    vgmi.lib()  # libvgmi.so first (libvghost links against it)
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m varigraph_amd.build`")
    l = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    l.vgh_last_error.restype = C.c_char_p
    l.vgh_graph_load.restype = C.c_int; l.vgh_graph_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    l.vgh_graph_free.restype = None; l.vgh_graph_free.argtypes = [vp]
    l.vgh_graph_get_info.restype = C.c_int; l.vgh_graph_get_info.argtypes = [vp, C.POINTER(GraphInfo)]
    for name in ("keys", "f", "bitvec", "hom_flag", "node_off", "node_key_index", "node_start", "node_chr"):
        fn = getattr(l, "vgh_graph_" + name)
        fn.restype = vp
        fn.argtypes = [vp]
    l.vgh_graph_chr_name.restype = C.c_char_p; l.vgh_graph_chr_name.argtypes = [vp, C.c_uint32]
    l.vgh_graph_upload.restype = C.c_int; l.vgh_graph_upload.argtypes = [vp, vp]
    l.vgh_fastx_read_all.restype = C.c_int64
    l.vgh_fastx_read_all.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
    l.vgh_fastx_read_all_mt.restype = C.c_int64
    l.vgh_fastx_read_all_mt.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64),
                                        C.c_char_p]
    l.vgh_free.restype = None; l.vgh_free.argtypes = [vp]
    l.vgh_sample_count.restype = C.c_int
    l.vgh_sample_count.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp,
                                   C.POINTER(SampleStats)]
    l.vgh_bloom_reference_seeds.restype = C.c_int
    l.vgh_bloom_reference_seeds.argtypes = [C.c_uint32, C.c_uint32, vp]
    l.vgh_make_mbf.restype = C.c_int
    l.vgh_make_mbf.argtypes = [vp, C.c_char_p, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    l.vgh_coverage_stats.restype = C.c_int
    l.vgh_coverage_stats.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(SampleStats)]
    l.vgh_genotype_config_default.restype = None; l.vgh_genotype_config_default.argtypes = [C.POINTER(GenotypeConfig)]
    l.vgh_genotyper_create.restype = C.c_int; l.vgh_genotyper_create.argtypes = [vp, C.POINTER(vp)]
    l.vgh_genotyper_free.restype = None; l.vgh_genotyper_free.argtypes = [vp]
    l.vgh_genotype.restype = C.c_int
    l.vgh_genotype.argtypes = [vp, vp, C.c_float, C.c_char_p, C.POINTER(GenotypeConfig), C.POINTER(vp), C.POINTER(C.c_size_t)]
    _lib = l
    return l


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class Graph:
    """A loaded graph.bin (host arrays stay in the C++ object; numpy copies on request)."""

    def __init__(self, path):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.vgh_graph_load(os.fsencode(path), C.byref(h))
        if rc:
            raise RuntimeError(self._l.vgh_last_error().decode())
        self._h = h
        info = GraphInfo()
        self._l.vgh_graph_get_info(h, C.byref(info))
        self.info = {f[0]: getattr(info, f[0]) for f in GraphInfo._fields_}

    def close(self):
        if getattr(self, "_h", None):
            self._l.vgh_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def arrays(self):
        i = self.info
        l, h = self._l, self._h
        n, nv, ne = i["n_keys"], i["n_variant_nodes"], i["n_node_entries"]
        return {
            "k": i["k"], "vcf_ploidy": i["vcf_ploidy"], "hap_num": i["hap_num"], "genome_size": i["genome_size"],
            "keys": _arr(l.vgh_graph_keys(h), n, np.uint64),
            "f": _arr(l.vgh_graph_f(h), n, np.uint8),
            "bitvec": _arr(l.vgh_graph_bitvec(h), n * i["bitlen"], np.int8).reshape(n, i["bitlen"]),
            "hom_flag": _arr(l.vgh_graph_hom_flag(h), n, np.uint8),
            "node_off": _arr(l.vgh_graph_node_off(h), nv + 1, np.uint64),
            "node_key_index": _arr(l.vgh_graph_node_key_index(h), ne, np.uint32),
            "node_start": _arr(l.vgh_graph_node_start(h), nv, np.uint32),
            "node_chr": _arr(l.vgh_graph_node_chr(h), nv, np.uint32),
            "chr_names": [l.vgh_graph_chr_name(h, c).decode() for c in range(i["n_chromosomes"])],
        }

    def upload(self, ctx):
        rc = self._l.vgh_graph_upload(self._h, ctx._h)
        if rc:
            raise vgmi.VgmiError(rc, self._l.vgh_last_error().decode())
        ctx.n_keys = self.info["n_keys"]
        ctx.n_node_entries = self.info["n_node_entries"]

    def reads_index_save(self, path, cov, read_base):
        """FastqKmer::save_index: the reference's per-sample dump of the k-mer table."""
        cov = np.ascontiguousarray(cov, dtype=np.uint8)
        assert cov.size == self.info["n_keys"]
        self._l.vgh_reads_index_save.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p]
        if self._l.vgh_reads_index_save(self._h, vgmi._ptr(cov), int(read_base), os.fsencode(path)):
            raise RuntimeError(self._l.vgh_last_error().decode())

    def reads_index_load(self, path):
        cov = np.empty(self.info["n_keys"], dtype=np.uint8)
        rb = C.c_uint64()
        self._l.vgh_reads_index_load.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_uint64)]
        if self._l.vgh_reads_index_load(self._h, os.fsencode(path), vgmi._ptr(cov), C.byref(rb)):
            raise RuntimeError(self._l.vgh_last_error().decode())
        return cov, rb.value

    def sample_count(self, ctx, fastq_paths, threads=4, sample_ploidy=2, use_depth=False, require_depth=True):
        """FastqKmerHip::build_fastq_index + coverage statistics for one sample.  require_depth=False: a sample too thin
        for the coverage peak (the reference exits with "Failed to retrieve depth information") still returns its
        counters."""
        i = self.info
        cov = np.empty(i["n_keys"], dtype=np.uint8)
        cov_node = np.empty(i["n_node_entries"], dtype=np.uint8)
        hist = np.zeros(256, dtype=np.uint64)
        st = SampleStats()
        arr = (C.c_char_p * len(fastq_paths))(*[os.fsencode(p) for p in fastq_paths])
        rc = self._l.vgh_sample_count(self._h, ctx._h, arr, len(fastq_paths), threads, sample_ploidy, int(use_depth),
                                      vgmi._ptr(cov), vgmi._ptr(cov_node), vgmi._ptr(hist), C.byref(st))
        if rc and (require_depth or b"Failed to retrieve depth" not in self._l.vgh_last_error()):
            raise vgmi.VgmiError(rc, self._l.vgh_last_error().decode())
        stats = {f[0]: getattr(st, f[0]) for f in SampleStats._fields_}
        return cov, cov_node, hist, stats


def coverage_stats(hist, read_base, genome_size, sample_ploidy=2, use_depth=False):
    """ReadDepth_ / hapKmerCoverage_ / peaks from a masked coverage histogram (host arithmetic of the reference)."""
    hist = np.ascontiguousarray(hist, dtype=np.uint64)
    st = SampleStats()
    rc = lib().vgh_coverage_stats(hist.ctypes.data_as(C.c_void_p), int(read_base), int(genome_size), sample_ploidy,
                                  1 if use_depth else 0, C.byref(st))
    if rc:
        raise RuntimeError(lib().vgh_last_error().decode())
    return {f[0]: getattr(st, f[0]) for f in SampleStats._fields_}


class Genotyper:
    """Host HMM of `genotype` (include/vghost.h): keeps the per-node k-mer lists across samples like the reference."""

    def __init__(self, graph):
        self._l = lib()
        self._graph = graph   # keeps the C++ graph alive
        h = C.c_void_p()
        rc = self._l.vgh_genotyper_create(graph._h, C.byref(h))
        if rc:
            raise RuntimeError(self._l.vgh_last_error().decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.vgh_genotyper_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, cov, hap_kmer_coverage, sample_name, sample_type="het", sample_ploidy=2, haploid_num=15,
            granularity_bp=1000000, transition="rec", sv_only=False, threads=4, min_gq=0.0):
        """cov: c per key in graph.bin record order.  Returns the VCF text (bytes)."""
        cov = np.ascontiguousarray(cov, dtype=np.uint8)
        assert cov.size == self._graph.info["n_keys"]
        cfg = GenotypeConfig()
        self._l.vgh_genotype_config_default(C.byref(cfg))
        st, tr = sample_type.encode(), transition.encode()
        cfg.sample_type, cfg.transition = st, tr
        cfg.sample_ploidy, cfg.haploid_num, cfg.chr_len_thread = sample_ploidy, haploid_num, granularity_bp
        cfg.sv_only, cfg.threads, cfg.min_gq = int(sv_only), threads, min_gq
        out, n = C.c_void_p(), C.c_size_t()
        rc = self._l.vgh_genotype(self._h, cov.ctypes.data_as(C.c_void_p), C.c_float(hap_kmer_coverage),
                                  sample_name.encode(), C.byref(cfg), C.byref(out), C.byref(n))
        if rc:
            raise RuntimeError(self._l.vgh_last_error().decode())
        try:
            return C.string_at(out, n.value)
        finally:
            self._l.vgh_free(out)


def load_graph(path):
    """graph.bin -> dict of numpy arrays (keys, f, bitvec, hom_flag, node CSR, ...)."""
    g = Graph(path)
    try:
        return g.arrays()
    finally:
        g.close()


def write_vcf_gz(path, text, threads=1):
    """`text` (bytes) as block gzip, the way `varigraph-mi genotype` writes <sample>.varigraph.vcf.gz."""
    l = lib()
    l.vgh_write_vcf_gz.restype = C.c_int
    l.vgh_write_vcf_gz.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_uint32]
    if l.vgh_write_vcf_gz(os.fsencode(path), text, len(text), threads):
        raise RuntimeError(l.vgh_last_error().decode())


def fastx_read_all(path, decode_threads=1, with_kind=False):
    """All records of a FASTA/Q(.gz) file as a '\\n'-joined block (kseq_read semantics).  with_kind: also how the
    bytes were decoded ("plain" | "gzip" | "bgzf", csrc/host/byte_source.hpp)."""
    l = lib()
    p, n, rb = C.c_void_p(), C.c_size_t(), C.c_uint64()
    kind = C.create_string_buffer(8)
    cnt = l.vgh_fastx_read_all_mt(os.fsencode(path), decode_threads, C.byref(p), C.byref(n), C.byref(rb), kind)
    if cnt < 0:
        raise RuntimeError(l.vgh_last_error().decode())
    try:
        block = _arr(p.value, n.value, np.uint8)
    finally:
        l.vgh_free(p)
    if with_kind:
        return block, int(cnt), rb.value, kind.value.decode()
    return block, int(cnt), rb.value


def bloom_reference_seeds(random_device_value, n_hash):
    out = np.zeros(n_hash, dtype=np.uint64)
    rc = lib().vgh_bloom_reference_seeds(random_device_value, n_hash, vgmi._ptr(out))
    if rc:
        raise RuntimeError("vgh_bloom_reference_seeds")
    return out


def make_mbf(ctx, fasta_path, k, seeds=None, random_device_value=0):
    """build_fasta_index + make_mbf with the Bloom filter on the device; returns (genome_size, m, n_hash)."""
    l = lib()
    gs, m, nh = C.c_uint64(), C.c_uint64(), C.c_uint32()
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    rc = l.vgh_make_mbf(ctx._h, os.fsencode(fasta_path), k, vgmi._ptr(seeds), 0 if seeds is None else seeds.size,
                        random_device_value, C.byref(gs), C.byref(m), C.byref(nh))
    if rc:
        raise vgmi.VgmiError(rc, l.vgh_last_error().decode())
    ctx._bloom_m = m.value
    return gs.value, m.value, nh.value
