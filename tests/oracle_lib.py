"""ctypes binding of oracle/liboracle.so -- the CPU checker.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    l = C.CDLL(LIB_PATH)
    vp, u64, u32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t
    l.vgo_hash64.restype = u64; l.vgo_hash64.argtypes = [u64, u64]
    l.vgo_sketch.restype = sz; l.vgo_sketch.argtypes = [vp, sz, u32, vp]
    l.vgo_table_new.restype = vp; l.vgo_table_new.argtypes = [vp, sz]
    l.vgo_table_free.restype = None; l.vgo_table_free.argtypes = [vp]
    l.vgo_table_find.restype = C.c_int64; l.vgo_table_find.argtypes = [vp, u64]
    l.vgo_count_read.restype = C.c_int64; l.vgo_count_read.argtypes = [vp, vp, sz, u32]
    l.vgo_count_block.restype = C.c_int64; l.vgo_count_block.argtypes = [vp, vp, sz, u32, C.POINTER(u64)]
    l.vgo_table_counts.restype = None; l.vgo_table_counts.argtypes = [vp, vp]
    l.vgo_table_reset.restype = None; l.vgo_table_reset.argtypes = [vp]
    l.vgo_bloom_size.restype = u64; l.vgo_bloom_size.argtypes = [u64, C.c_double]
    l.vgo_bloom_num_hashes.restype = u32; l.vgo_bloom_num_hashes.argtypes = [u64, u64]
    l.vgo_murmur_sum.restype = u64; l.vgo_murmur_sum.argtypes = [u64, u64]
    l.vgo_bloom_add.restype = None; l.vgo_bloom_add.argtypes = [vp, u64, vp, u32, u64]
    l.vgo_bloom_add_seq.restype = C.c_int64; l.vgo_bloom_add_seq.argtypes = [vp, u64, vp, u32, vp, sz, u32]
    l.vgo_bloom_count.restype = C.c_uint8; l.vgo_bloom_count.argtypes = [vp, u64, vp, u32, u64]
    l.vgo_bloom_find.restype = C.c_int; l.vgo_bloom_find.argtypes = [vp, u64, vp, u32, u64]
    l.vgo_hom_hist.restype = None; l.vgo_hom_hist.argtypes = [vp, vp, vp, sz, sz, u32, u32, vp]
    l.vgo_hom_peak.restype = C.c_int; l.vgo_hom_peak.argtypes = [vp, C.c_float, vp, vp]
    l.vgo_read_depth.restype = C.c_float; l.vgo_read_depth.argtypes = [u64, u64]
    l.vgo_hap_kmer_cov.restype = C.c_float; l.vgo_hap_kmer_cov.argtypes = [C.c_uint8, u32, C.c_float]
    l.vgo_use_depth_cov.restype = C.c_uint8; l.vgo_use_depth_cov.argtypes = [C.c_float]
    _lib = l
    return l


def _p(a):
    return C.c_void_p(a.ctypes.data)


def hash64(key, k):
    return lib().vgo_hash64(key, (1 << (2 * k)) - 1)


def sketch(seq, k):
    """All keys the reference emits for one sequence, in order (None where it would assert)."""
    s = np.frombuffer(seq if isinstance(seq, (bytes, bytearray)) else bytes(seq), dtype=np.uint8)
    out = np.empty(max(len(s), 1), dtype=np.uint64)
    buf = np.ascontiguousarray(s) if len(s) else np.zeros(1, dtype=np.uint8)
    n = lib().vgo_sketch(_p(buf), len(s), k, _p(out))
    if n == C.c_size_t(-1).value:
        return None
    return out[:n].copy()


def sketch_block_positions(block, k):
    """Per-byte key array for a '\\n'-joined read block: key of the k-mer ending at byte i or ~0.
    Built from vgo_sketch per read with the emission positions recomputed by the same state machine."""
    block = np.ascontiguousarray(block, dtype=np.uint8)
    out = np.full(block.size, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nl = np.flatnonzero(block == 10)
    start = 0
    for e in nl.tolist():
        seq = block[start:e]
        if len(seq):
            keys = sketch(seq.tobytes(), k)
            pos = _emit_positions(seq, k)
            assert len(pos) == len(keys)
            out[start + np.asarray(pos, dtype=np.int64)] = keys
        start = e + 1
    return out


def _emit_positions(seq, k):
    """Positions at which src/kmer.cpp:126-146 emits (pure python, small inputs only)."""
    tbl = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3, 85: 3, 117: 3, 0: 0, 1: 1, 2: 2, 3: 3}
    mask = (1 << (2 * k)) - 1
    sh = 2 * (k - 1)
    fwd = rc = 0
    l = 0
    pos = []
    for i, ch in enumerate(seq.tolist()):
        c = tbl.get(ch, 4)
        if c < 4:
            fwd = ((fwd << 2) | c) & mask
            rc = (rc >> 2) | ((3 ^ c) << sh)
            if fwd == rc:
                continue
            l += 1
            if l >= k:
                pos.append(i)
        else:
            l = 0
    return pos


class Table:
    def __init__(self, keys):
        self.keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._h = lib().vgo_table_new(_p(self.keys), self.keys.size)
        if not self._h:
            raise ValueError("duplicate key")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vgo_table_free(self._h)
            self._h = None

    def count_block(self, block, k):
        block = np.ascontiguousarray(block, dtype=np.uint8)
        rb = C.c_uint64(0)
        hits = lib().vgo_count_block(self._h, _p(block), block.size, k, C.byref(rb))
        return hits, rb.value

    def counts(self):
        out = np.empty(self.keys.size, dtype=np.uint8)
        lib().vgo_table_counts(self._h, _p(out))
        return out

    def reset(self):
        lib().vgo_table_reset(self._h)


def bloom_add_seq(filt, seeds, seq, k):
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    return lib().vgo_bloom_add_seq(_p(filt), filt.size, _p(seeds), seeds.size, _p(seq), seq.size, k)


def bloom_query(filt, seeds, keys):
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    mn = np.array([lib().vgo_bloom_count(_p(filt), filt.size, _p(seeds), seeds.size, int(x)) for x in keys], dtype=np.uint8)
    nz = np.array([lib().vgo_bloom_find(_p(filt), filt.size, _p(seeds), seeds.size, int(x)) for x in keys], dtype=np.uint8)
    return mn, nz


def hom_hist(c, f, bitvec, hap_num, vcf_ploidy):
    c = np.ascontiguousarray(c, dtype=np.uint8)
    f = np.ascontiguousarray(f, dtype=np.uint8)
    bitvec = np.ascontiguousarray(bitvec, dtype=np.int8)
    hist = np.zeros(256, dtype=np.uint64)
    lib().vgo_hom_hist(_p(c), _p(f), _p(bitvec), bitvec.shape[1], c.size, hap_num, vcf_ploidy, _p(hist))
    return hist


def hom_peak(hist, read_depth):
    hist = np.ascontiguousarray(hist, dtype=np.uint64)
    mx, hm = C.c_uint8(), C.c_uint8()
    rc = lib().vgo_hom_peak(_p(hist), read_depth, C.byref(mx), C.byref(hm))
    if rc:
        return None
    return mx.value, hm.value
