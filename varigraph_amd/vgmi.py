"""ctypes binding of include/vgmi.h (libvgmi.so).  No fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvgmi.so")

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_STATE, E_DUPLICATE_KEY, E_BAD_KEY, E_EMPTY_READ, E_NOMEM = range(-1, -9, -1)


class VgmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vgmi error {code}: {msg}")
        self.code = code


def load_library():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m varigraph_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch wheels bundle their own libamdhip64; load torch FIRST so the process ends up with one
    # HIP runtime (torch is only plumbing here: device buffers and torch.distributed)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, sz, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int
    sig = {
        "vgmi_device_count": (i32, []),
        "vgmi_create": (i32, [i32, sz, C.POINTER(vp)]),
        "vgmi_destroy": (None, [vp]),
        "vgmi_last_error": (C.c_char_p, [vp]),
        "vgmi_stream": (vp, [vp]),
        "vgmi_device_of": (i32, [vp, C.POINTER(i32)]),
        "vgmi_device_memory": (i32, [vp, C.POINTER(sz), C.POINTER(sz)]),
        "vgmi_table_upload": (i32, [vp, vp, sz, u32]),
        "vgmi_table_image_bytes": (i32, [vp, C.POINTER(sz)]),
        "vgmi_table_export": (i32, [vp, vp, sz]),
        "vgmi_table_import": (i32, [vp, vp, sz]),
        "vgmi_table_clone": (i32, [vp, vp]),
        "vgmi_rccl_unique_id": (i32, [vp]),
        "vgmi_table_broadcast": (i32, [vp, i32, i32, vp]),
        "vgmi_comm_create": (i32, [i32, i32, i32, vp, C.POINTER(vp)]),
        "vgmi_table_broadcast_comm": (i32, [vp, vp]),
        "vgmi_table_snapshot": (i32, [vp]),
        "vgmi_comm_destroy": (None, [vp]),
        "vgmi_table_info": (i32, [vp, C.POINTER(sz), C.POINTER(u32), C.POINTER(sz), C.POINTER(sz)]),
        "vgmi_xtable_info": (i32, [vp, C.POINTER(sz), C.POINTER(sz)]),
        "vgmi_ctable_info": (i32, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "vgmi_table_lookup": (i32, [vp, vp, sz, vp]),
        "vgmi_nodes_upload": (i32, [vp, vp, vp, sz]),
        "vgmi_flags_upload": (i32, [vp, vp]),
        "vgmi_counts_reset": (i32, [vp]),
        "vgmi_reads_submit": (i32, [vp, vp, sz, vp, sz]),
        "vgmi_reads_submit_device": (i32, [vp, vp, sz, vp, sz]),
        "vgmi_read_base": (i32, [vp, C.POINTER(u64)]),
        "vgmi_counts_finish": (i32, [vp, vp, vp, vp]),
        "vgmi_counts_finish_device": (i32, [vp, vp, vp, vp]),
        "vgmi_count_kernel_ms": (i32, [vp, C.POINTER(C.c_float), C.POINTER(u64)]),
        "vgmi_counts_export_device": (i32, [vp, vp]),
        "vgmi_counts_import_device": (i32, [vp, vp]),
        "vgmi_fastq_open": (i32, [vp, C.POINTER(vp)]),
        "vgmi_fastq_acquire": (i32, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "vgmi_fastq_text_capacity": (i32, [vp, C.POINTER(sz)]),
        "vgmi_fastq_commit": (i32, [vp, sz]),
        "vgmi_fastq_commit_bgzf": (i32, [vp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(i32)]),
        "vgmi_fastq_bgzf_status": (i32, [vp, C.POINTER(i32), C.POINTER(u64), C.POINTER(u32)]),
        "vgmi_fastq_commit_gzip": (i32, [vp, sz, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(i32)]),
        "vgmi_fastq_gzip_status": (i32, [vp, C.POINTER(u64), C.POINTER(u32)]),
        "vgmi_gunzip_buffer": (i32, [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(i32), C.POINTER(u32)]),
        "vgmi_fastq_close": (i32, [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(i32), vp, sz, C.POINTER(sz)]),
        "vgmi_sketch_keys": (i32, [vp, vp, sz, vp, sz, u32, vp]),
        "vgmi_bloom_params": (i32, [u64, C.c_double, C.POINTER(u64), C.POINTER(u32)]),
        "vgmi_bloom_create": (i32, [vp, u64, u32, vp]),
        "vgmi_bloom_add_seq": (i32, [vp, vp, u64, u32]),
        "vgmi_bloom_add_seq_device": (i32, [vp, vp, u64, u32]),
        "vgmi_hmm_recursion": (i32, [vp, u32, u32, vp, u32, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, u32, vp]),
        "vgmi_hmm_calls": (i32, [vp, u32, u32, vp, u32, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]),
        "vgmi_hmm_calls_part": (i32, [vp, u32, u32, vp, u32, vp, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, u32,
                                      vp, vp, vp, vp, vp, vp]),
        "vgmi_hmm_entries_upload": (i32, [vp, vp, sz]),
        "vgmi_hmm_sample_upload": (i32, [vp, vp, sz]),
        "vgmi_hmm_emissions": (i32, [vp, u32, u32, vp, vp, vp, C.c_uint64, u32, C.c_float, C.c_double, C.c_double, vp, C.c_uint64, vp, vp, vp, vp, vp,
                                      C.POINTER(vp)]),
        "vgmi_hmm_emissions_ploidy": (i32, [vp, u32, u32, u32, vp, vp, C.c_uint64, u32, C.c_float, C.c_double, C.c_double, vp, C.c_uint64, vp, vp, vp, vp, vp,
                                      C.POINTER(vp)]),
        "vgmi_hmm_part_set_rows": (i32, [vp, C.c_uint64, vp, vp]),
        "vgmi_hmm_part_fix_rows": (i32, [vp, C.c_uint64, vp, vp, vp, vp]),
        "vgmi_hmm_plan_create": (i32, [vp, u32, u32, vp, u32, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, u32, vp, vp, vp, vp, C.POINTER(vp)]),
        "vgmi_hmm_part_calls_plan": (i32, [vp, vp, vp, vp]),
        "vgmi_hmm_plan_free": (None, [vp]),
        "vgmi_hmm_part_calls": (i32, [vp, u32, vp, u32, vp, vp, vp, C.c_uint64, vp, vp, u32, vp, vp, vp, vp, vp, vp]),
        "vgmi_hmm_part_fetch": (i32, [vp, vp]),
        "vgmi_hmm_part_free": (None, [vp]),
        "vgmi_bloom_fetch": (i32, [vp, vp]),
        "vgmi_bloom_save_file": (i32, [vp, C.c_char_p]),
        "vgmi_bloom_load_file": (i32, [vp, C.c_char_p]),
        "vgmi_bloom_load": (i32, [vp, vp]),
        "vgmi_bloom_query": (i32, [vp, vp, sz, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


_lib = None
SYMBOLS = None


def lib():
    global _lib, SYMBOLS
    if _lib is None:
        _lib, SYMBOLS = load_library()
    return _lib


def _ptr(a):
    """numpy array / torch tensor / int / None -> void*"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if isinstance(a, (bytes, bytearray)):
        return C.cast(C.c_char_p(bytes(a)), C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def bloom_params(n, p):
    m, nh = C.c_uint64(), C.c_uint32()
    lib().vgmi_bloom_params(n, p, C.byref(m), C.byref(nh))
    return m.value, nh.value


# ---- tooling: the seeded synthetic workloads (libvgsynth.so, varigraph_amd/synthlib.py -- NOT part of the product library)
def synth_reference(seed, length):
    from . import synthlib
    return synthlib.reference(seed, length)


def synth_snp_keys(ref, pos, alts, k=27):
    from . import synthlib
    return synthlib.snp_keys(ref, pos, alts, k)


def synth_reads_host(seed, first_read, n_reads, read_len, haps):
    from . import synthlib
    return synthlib.reads_host(seed, first_read, n_reads, read_len, haps)


class Context:
    """One vgmi context (one GPU)."""

    def __init__(self, device=0, buffer_mib=100):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.vgmi_create(device, buffer_mib, C.byref(h))
        if rc:
            raise VgmiError(rc, (self._l.vgmi_last_error(None) or b"").decode())
        self._h = h
        self.n_keys = 0
        self.n_node_entries = 0

    def close(self):
        if getattr(self, "_h", None):
            self._l.vgmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise VgmiError(rc, (self._l.vgmi_last_error(self._h) or b"").decode())

    @property
    def stream(self):
        return self._l.vgmi_stream(self._h)

    # ---- table
    def table_upload(self, keys, k):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._chk(self._l.vgmi_table_upload(self._h, _ptr(keys), keys.size, k))
        self.n_keys = keys.size
        self.n_node_entries = 0

    def table_image_bytes(self):
        n = C.c_size_t()
        self._chk(self._l.vgmi_table_image_bytes(self._h, C.byref(n)))
        return n.value

    def table_export(self, dev_tensor):
        self._chk(self._l.vgmi_table_export(self._h, _ptr(dev_tensor), dev_tensor.numel()))

    def table_import(self, dev_tensor):
        self._chk(self._l.vgmi_table_import(self._h, _ptr(dev_tensor), dev_tensor.numel()))
        self.n_keys = self.table_info()["n_keys"]
        self.n_node_entries = 0

    def table_clone_from(self, src):
        """Adopt a device-to-device copy of `src`'s table image (one process, several contexts / devices)."""
        self._chk(self._l.vgmi_table_clone(self._h, src._h))
        self.n_keys = self.table_info()["n_keys"]
        self.n_node_entries = 0

    def device_memory(self):
        f, t = C.c_size_t(), C.c_size_t()
        self._chk(self._l.vgmi_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def table_info(self):
        n, k, s, f = C.c_size_t(), C.c_uint32(), C.c_size_t(), C.c_size_t()
        self._chk(self._l.vgmi_table_info(self._h, C.byref(n), C.byref(k), C.byref(s), C.byref(f)))
        return {"n_keys": n.value, "k": k.value, "n_slots": s.value, "filter_bits": f.value}

    def table_lookup(self, keys):
        """Index of every key in the uploaded key array, 0xFFFFFFFF where the table does not hold it (graph2node's find, batched)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty(keys.size, dtype=np.uint32)
        self._chk(self._l.vgmi_table_lookup(self._h, _ptr(keys) if keys.size else None, keys.size, _ptr(out) if keys.size else None))
        return out

    def xtable_info(self):
        n, o = C.c_size_t(), C.c_size_t()
        self._chk(self._l.vgmi_xtable_info(self._h, C.byref(n), C.byref(o)))
        return {"n_lines": n.value, "overflow_pairs": o.value}

    def ctable_info(self):
        v = [C.c_size_t() for _ in range(5)]
        self._chk(self._l.vgmi_ctable_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("n_buckets", "n_entries", "n_unitigs", "moved_entries", "overflow_kmers"), (x.value for x in v)))

    def nodes_upload(self, node_off, key_index):
        node_off = np.ascontiguousarray(node_off, dtype=np.uint64)
        key_index = np.ascontiguousarray(key_index, dtype=np.uint32)
        self._chk(self._l.vgmi_nodes_upload(self._h, _ptr(node_off), _ptr(key_index), node_off.size - 1))
        self.n_node_entries = int(node_off[-1])

    def flags_upload(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert flags.size == self.n_keys
        self._chk(self._l.vgmi_flags_upload(self._h, _ptr(flags)))

    # ---- per sample
    def counts_reset(self):
        self._chk(self._l.vgmi_counts_reset(self._h))

    def reads_submit(self, block, n_reads, read_off=None):
        block = np.ascontiguousarray(block, dtype=np.uint8)
        if read_off is not None:
            read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        self._chk(self._l.vgmi_reads_submit(self._h, _ptr(block), block.size, _ptr(read_off), n_reads))

    def reads_submit_device(self, dev_block, n_bytes, n_reads, dev_read_off=None):
        self._chk(self._l.vgmi_reads_submit_device(self._h, _ptr(dev_block), n_bytes, _ptr(dev_read_off), n_reads))

    def read_base(self):
        v = C.c_uint64()
        self._chk(self._l.vgmi_read_base(self._h, C.byref(v)))
        return v.value

    def counts_finish(self, nodes=False, hist=False):
        cov = np.empty(self.n_keys, dtype=np.uint8)
        cov_node = np.empty(self.n_node_entries, dtype=np.uint8) if nodes else None
        h = np.zeros(256, dtype=np.uint64) if hist else None
        self._chk(self._l.vgmi_counts_finish(self._h, _ptr(cov), _ptr(cov_node), _ptr(h)))
        return cov, cov_node, h

    def counts_finish_device(self, dev_cov=None, dev_cov_node=None, dev_hist=None):
        self._chk(self._l.vgmi_counts_finish_device(self._h, _ptr(dev_cov), _ptr(dev_cov_node), _ptr(dev_hist)))

    def counts_export_device(self, dev_u32):
        self._chk(self._l.vgmi_counts_export_device(self._h, _ptr(dev_u32)))

    def counts_import_device(self, dev_u32):
        self._chk(self._l.vgmi_counts_import_device(self._h, _ptr(dev_u32)))

    def count_kernel_ms(self):
        ms, n = C.c_float(), C.c_uint64()
        self._chk(self._l.vgmi_count_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def fastq_text(self, text, piece=None):
        """Device-side FASTQ parser over one stream of file text (bytes), committed in pieces of `piece` bytes (default:
        whole staging buffers).  Returns dict(n_records, n_bases, consumed, stopped, tail)."""
        fq = C.c_void_p()
        self._chk(self._l.vgmi_fastq_open(self._h, C.byref(fq)))
        mv = memoryview(text)
        pos = 0
        try:
            while True:
                buf, cap = C.c_void_p(), C.c_size_t()
                self._chk(self._l.vgmi_fastq_acquire(fq, C.byref(buf), C.byref(cap)))
                n = min(cap.value if piece is None else min(piece, cap.value), len(mv) - pos)
                C.memmove(buf, bytes(mv[pos:pos + n]), n)
                self._chk(self._l.vgmi_fastq_commit(fq, n))
                pos += n
                if pos >= len(mv):
                    break
        finally:
            nr, nb, cons, st, tl = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int(), C.c_size_t()
            tail = C.create_string_buffer(1 << 20)
            rc = self._l.vgmi_fastq_close(fq, C.byref(nr), C.byref(nb), C.byref(cons), C.byref(st), tail, 1 << 20, C.byref(tl))
        self._chk(rc)
        return {"n_records": nr.value, "n_bases": nb.value, "consumed": cons.value, "stopped": bool(st.value),
                "tail": tail.raw[:tl.value]}

    def fastq_bgzf(self, comp, piece=None):
        """Block-gzip bytes through the device inflate + parser.  Returns the dict of fastq_text plus inflate_failed,
        good_compressed_bytes, taken (compressed bytes handed over as whole members)."""
        fq = C.c_void_p()
        self._chk(self._l.vgmi_fastq_open(self._h, C.byref(fq)))
        comp = bytes(comp)
        pos, carry, total_taken, n_call = 0, b"", 0, 0
        failed, good, reason = C.c_int(), C.c_uint64(), C.c_uint32()
        try:
            while True:
                buf, cap = C.c_void_p(), C.c_size_t()
                self._chk(self._l.vgmi_fastq_acquire(fq, C.byref(buf), C.byref(cap)))
                if isinstance(piece, (list, tuple)):      # a size per call (the last one from there on)
                    want = piece[min(n_call, len(piece) - 1)]
                else:
                    want = piece
                n_call += 1
                room = (cap.value if want is None else min(want, cap.value)) - len(carry)
                new = comp[pos:pos + max(room, 0)]
                pos += len(new)
                data = carry + new
                C.memmove(buf, data, len(data))
                taken, n_text, nb = C.c_size_t(), C.c_size_t(), C.c_int()
                self._chk(self._l.vgmi_fastq_commit_bgzf(fq, len(data), C.byref(taken), C.byref(n_text), C.byref(nb)))
                total_taken += taken.value
                carry = data[taken.value:]
                if nb.value or (pos >= len(comp) and (taken.value == 0 or not carry)):
                    break
            self._chk(self._l.vgmi_fastq_bgzf_status(fq, C.byref(failed), C.byref(good), C.byref(reason)))
        finally:
            nr, nbs, cons, st, tl = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int(), C.c_size_t()
            tail = C.create_string_buffer(1 << 20)
            rc = self._l.vgmi_fastq_close(fq, C.byref(nr), C.byref(nbs), C.byref(cons), C.byref(st), tail, 1 << 20, C.byref(tl))
        self._chk(rc)
        return {"n_records": nr.value, "n_bases": nbs.value, "consumed": cons.value, "stopped": bool(st.value),
                "tail": tail.raw[:tl.value], "inflate_failed": bool(failed.value), "good_compressed_bytes": good.value,
                "reason": reason.value, "taken": total_taken}

    def fastq_gzip(self, comp, piece=None):
        """Ordinary gzip bytes through the device inflate + parser (vgmi_fastq_commit_gzip).  Returns the dict of fastq_text plus stop
        (1 data over, 2 the device gave up), device_text_bytes, reason, taken (compressed bytes used up)."""
        fq = C.c_void_p()
        self._chk(self._l.vgmi_fastq_open(self._h, C.byref(fq)))
        comp = bytes(comp)
        pos, carry, total_taken, stop, n_call = 0, b"", 0, 0, 0
        dtext, reason = C.c_uint64(), C.c_uint32()
        try:
            while True:
                buf, cap = C.c_void_p(), C.c_size_t()
                self._chk(self._l.vgmi_fastq_acquire(fq, C.byref(buf), C.byref(cap)))
                if isinstance(piece, (list, tuple)):      # a size per call (the last one from there on)
                    want = piece[min(n_call, len(piece) - 1)]
                else:
                    want = piece
                n_call += 1
                room = (cap.value if want is None else min(want, cap.value)) - len(carry)
                new = comp[pos:pos + max(room, 0)]
                pos += len(new)
                data = carry + new
                C.memmove(buf, data, len(data))
                taken, n_text, st = C.c_size_t(), C.c_size_t(), C.c_int()
                self._chk(self._l.vgmi_fastq_commit_gzip(fq, len(data), int(pos >= len(comp)), C.byref(taken), C.byref(n_text), C.byref(st)))
                total_taken += taken.value
                carry = data[taken.value:]
                stop = st.value
                if stop or (pos >= len(comp) and taken.value == 0):
                    break
            self._chk(self._l.vgmi_fastq_gzip_status(fq, C.byref(dtext), C.byref(reason)))
        finally:
            nr, nbs, cons, sp, tl = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int(), C.c_size_t()
            tail = C.create_string_buffer(1 << 20)
            rc = self._l.vgmi_fastq_close(fq, C.byref(nr), C.byref(nbs), C.byref(cons), C.byref(sp), tail, 1 << 20, C.byref(tl))
        self._chk(rc)
        return {"n_records": nr.value, "n_bases": nbs.value, "consumed": cons.value, "stopped": bool(sp.value), "tail": tail.raw[:tl.value],
                "stop": stop, "device_text_bytes": dtext.value, "reason": reason.value, "taken": total_taken}

    def gunzip(self, comp, cap):
        """An ordinary gzip member through the device pipeline (vgmi_gunzip_buffer): (text bytes, compressed bytes consumed, member
        end reached, reason the device stopped early)."""
        comp = bytes(comp)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n_out, cons, fin, why = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_uint32()
        self._chk(self._l.vgmi_gunzip_buffer(self._h, comp, len(comp), _ptr(out), cap, C.byref(n_out), C.byref(cons), C.byref(fin), C.byref(why)))
        return out[:n_out.value].tobytes(), cons.value, bool(fin.value), why.value

    def sketch_keys(self, block, n_reads, k, read_off=None):
        block = np.ascontiguousarray(block, dtype=np.uint8)
        if read_off is not None:
            read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        out = np.empty(block.size, dtype=np.uint64)
        self._chk(self._l.vgmi_sketch_keys(self._h, _ptr(block), block.size, _ptr(read_off), n_reads, k, _ptr(out)))
        return out

    # ---- bloom
    def bloom_create(self, m, n_hash, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.size == n_hash
        self._chk(self._l.vgmi_bloom_create(self._h, m, n_hash, _ptr(seeds)))
        self._bloom_m = m

    def bloom_add_seq(self, seq, k):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self._chk(self._l.vgmi_bloom_add_seq(self._h, _ptr(seq), seq.size, k))

    def bloom_add_seq_device(self, dev_seq, length, k):
        self._chk(self._l.vgmi_bloom_add_seq_device(self._h, _ptr(dev_seq), length, k))

    def hmm_recursion(self, keep, obs, row, restart, pow_tables, uniform, chains, ploidy):
        """Forward / backward recursion of the HMM on the device (vgmi_hmm_recursion).  keep: (windows, n_gt, n_gt) uint8;
        obs: (rows, n_gt) longdouble; row, restart: per step; pow_tables: (steps, 2, ploidy + 1) longdouble; uniform:
        longdouble scalar; chains: list of (first_step, n_steps, keep_index).  Returns (steps, n_gt) longdouble."""
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        obs = np.ascontiguousarray(obs, dtype=np.longdouble)
        row = np.ascontiguousarray(row, dtype=np.uint32)
        restart = np.ascontiguousarray(restart, dtype=np.uint8)
        pow_tables = np.ascontiguousarray(pow_tables, dtype=np.longdouble)
        uni = np.ascontiguousarray([uniform], dtype=np.longdouble)
        n_gt = obs.shape[1]
        ch = np.zeros((len(chains), 3), dtype=np.uint64)
        for i, (f, n, k) in enumerate(chains):
            ch[i] = (f, n, k)       # keep_index | pad << 32 in the third word (little endian)
        out = np.zeros((row.size, n_gt), dtype=np.longdouble)
        assert obs.itemsize == 16 and pow_tables.shape == (row.size, 2, ploidy + 1)
        self._chk(self._l.vgmi_hmm_recursion(self._h, n_gt, ploidy, _ptr(keep), keep.shape[0], _ptr(obs), obs.shape[0], _ptr(row),
                                              _ptr(restart), _ptr(pow_tables), row.size, _ptr(uni), _ptr(ch), len(chains), _ptr(out)))
        return out

    def hmm_calls(self, keep, obs, row, restart, pow_tables, uniform, chains, ploidy, gid, order, fwd_step, bwd_step):
        """hmm_recursion followed by the posterior on the device (vgmi_hmm_calls): returns (prob, winner, alpha_beta)."""
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        obs = np.ascontiguousarray(obs, dtype=np.longdouble)
        row = np.ascontiguousarray(row, dtype=np.uint32)
        restart = np.ascontiguousarray(restart, dtype=np.uint8)
        pow_tables = np.ascontiguousarray(pow_tables, dtype=np.longdouble)
        uni = np.ascontiguousarray([uniform], dtype=np.longdouble)
        gid = np.ascontiguousarray(gid, dtype=np.uint8)
        order = np.ascontiguousarray(order, dtype=np.uint8)
        fwd_step = np.ascontiguousarray(fwd_step, dtype=np.uint64)
        bwd_step = np.ascontiguousarray(bwd_step, dtype=np.uint64)
        n_gt = obs.shape[1]
        ch = np.zeros((len(chains), 3), dtype=np.uint64)
        for i, (f, n, k) in enumerate(chains):
            ch[i] = (f, n, k)
        ab = np.zeros((row.size, n_gt), dtype=np.longdouble)
        prob = np.zeros(obs.shape[0], dtype=np.longdouble)
        winner = np.zeros(obs.shape[0], dtype=np.uint32)
        self._chk(self._l.vgmi_hmm_calls(self._h, n_gt, ploidy, _ptr(keep), keep.shape[0], _ptr(obs), obs.shape[0], _ptr(row), _ptr(restart),
                                          _ptr(pow_tables), row.size, _ptr(uni), _ptr(ch), len(chains), _ptr(gid), _ptr(order), _ptr(fwd_step),
                                          _ptr(bwd_step), _ptr(prob), _ptr(winner), _ptr(ab)))
        return prob, winner, ab

    def hmm_emissions(self, entries, cov_node, used, pos_a, pos_b, top_mask, bit_len, ave, lower, upper, tables, entry_begin, entry_count, gt0,
                      fixes=None, calls=None, pos=None):
        """vgmi_hmm_entries_upload + _sample_upload + _emissions + _part_fetch: returns (obs (rows, n_gt) longdouble, n_kept, flags).
        fixes = (rows, off, j, mask): vgmi_hmm_part_fix_rows before the fetch.  calls = dict(ploidy, keep, row, restart, pow, uniform,
        chains, gid, order, fwd, bwd): the part's recursion and posterior both ways -- vgmi_hmm_part_calls with host arrays and
        vgmi_hmm_plan_create + vgmi_hmm_part_calls_plan -- returned as a fourth item ((prob, winner), (prob, winner))."""
        entries = np.ascontiguousarray(entries, dtype=np.uint64)
        cov_node = np.ascontiguousarray(cov_node, dtype=np.uint8)
        used = np.ascontiguousarray(used, dtype=np.uint8)
        if pos is None:
            pos_a = np.ascontiguousarray(pos_a, dtype=np.uint8)
            pos_b = np.ascontiguousarray(pos_b, dtype=np.uint8)
        tables = np.ascontiguousarray(tables, dtype=np.longdouble)
        entry_begin = np.ascontiguousarray(entry_begin, dtype=np.uint64)
        entry_count = np.ascontiguousarray(entry_count, dtype=np.uint32)
        gt0 = np.ascontiguousarray(gt0, dtype=np.uint16)
        ploidy = 2
        if pos is not None:      # genotypes of 2 .. 4 haplotypes: pos (n_gt, ploidy) -> vgmi_hmm_emissions_ploidy
            pos = np.ascontiguousarray(pos, dtype=np.uint8)
            ploidy = pos.shape[1]
        assert tables.size == (ploidy + 1) * 256
        n_rows, n_gt = entry_begin.size, (pos.shape[0] if pos is not None else pos_a.size)
        self._chk(self._l.vgmi_hmm_entries_upload(self._h, _ptr(entries), entries.size))
        self._chk(self._l.vgmi_hmm_sample_upload(self._h, _ptr(cov_node), cov_node.size))
        n_kept = np.zeros(max(n_rows, 1), dtype=np.uint32)
        flags = np.zeros(max(n_rows, 1), dtype=np.uint8)
        part = C.c_void_p()
        if pos is not None:
            self._chk(self._l.vgmi_hmm_emissions_ploidy(self._h, n_gt, ploidy, used.size, _ptr(used), _ptr(pos), int(top_mask), bit_len, float(ave), float(lower),
                                                         float(upper), _ptr(tables), n_rows, _ptr(entry_begin), _ptr(entry_count), _ptr(gt0), _ptr(n_kept),
                                                         _ptr(flags), C.byref(part)))
        else:
            self._chk(self._l.vgmi_hmm_emissions(self._h, n_gt, used.size, _ptr(used), _ptr(pos_a), _ptr(pos_b), int(top_mask), bit_len, float(ave), float(lower),
                                                  float(upper), _ptr(tables), n_rows, _ptr(entry_begin), _ptr(entry_count), _ptr(gt0), _ptr(n_kept), _ptr(flags),
                                                  C.byref(part)))
        both = None
        try:
            if fixes is not None:
                f_rows, f_off, f_j, f_m = (np.ascontiguousarray(a, dtype=t) for a, t in zip(fixes, (np.uint64, np.uint32, np.uint32, np.uint16)))
                self._chk(self._l.vgmi_hmm_part_fix_rows(part, f_rows.size, _ptr(f_rows), _ptr(f_off), _ptr(f_j), _ptr(f_m)))
            obs = np.zeros((n_rows, n_gt), dtype=np.longdouble)
            self._chk(self._l.vgmi_hmm_part_fetch(part, _ptr(obs)))
            if calls is not None:
                keep = np.ascontiguousarray(calls["keep"], dtype=np.uint8)
                row = np.ascontiguousarray(calls["row"], dtype=np.uint32)
                restart = np.ascontiguousarray(calls["restart"], dtype=np.uint8)
                pw = np.ascontiguousarray(calls["pow"], dtype=np.longdouble)
                uni = np.ascontiguousarray([calls["uniform"]], dtype=np.longdouble)
                ch = np.zeros((len(calls["chains"]), 3), dtype=np.uint64)
                for i, (f, n, k) in enumerate(calls["chains"]):
                    ch[i] = (f, n, k)
                gid = np.ascontiguousarray(calls["gid"], dtype=np.uint8)
                order = np.ascontiguousarray(calls["order"], dtype=np.uint8)
                fwd = np.ascontiguousarray(calls["fwd"], dtype=np.uint64)
                bwd = np.ascontiguousarray(calls["bwd"], dtype=np.uint64)
                out = []
                p1, w1 = np.zeros(n_rows, dtype=np.longdouble), np.zeros(n_rows, dtype=np.uint32)
                self._chk(self._l.vgmi_hmm_part_calls(part, calls["ploidy"], _ptr(keep), 1, _ptr(row), _ptr(restart), _ptr(pw), row.size, _ptr(uni), _ptr(ch),
                                                       len(calls["chains"]), _ptr(gid), _ptr(order), _ptr(fwd), _ptr(bwd), _ptr(p1), _ptr(w1)))
                plan = C.c_void_p()
                self._chk(self._l.vgmi_hmm_plan_create(self._h, n_gt, calls["ploidy"], _ptr(keep), 1, n_rows, _ptr(row), _ptr(restart), _ptr(pw), row.size, _ptr(uni),
                                                        _ptr(ch), len(calls["chains"]), _ptr(gid), _ptr(order), _ptr(fwd), _ptr(bwd), C.byref(plan)))
                try:
                    for _ in range(2):      # a plan serves any number of calls
                        p2, w2 = np.zeros(n_rows, dtype=np.longdouble), np.zeros(n_rows, dtype=np.uint32)
                        self._chk(self._l.vgmi_hmm_part_calls_plan(part, plan, _ptr(p2), _ptr(w2)))
                        out.append((p2, w2))
                finally:
                    self._l.vgmi_hmm_plan_free(plan)
                both = ((p1, w1), out[0], out[1])
        finally:
            self._l.vgmi_hmm_part_free(part)
        if both is not None:
            return obs, n_kept[:n_rows], flags[:n_rows], both
        return obs, n_kept[:n_rows], flags[:n_rows]

    def hmm_calls_part(self, keep, obs, row, restart, pow_tables, uniform, chains, ploidy, gid, order, fwd_step, bwd_step, rows, steps,
                       prob, winner):
        """vgmi_hmm_calls_part: the arrays are the whole run's (global rows / steps), the call works on rows [rows[0], rows[1]) and
        steps [steps[0], steps[1]) and fills those rows of prob / winner (numpy arrays of the whole run); keep holds the
        matrices of this part's chains only.  Calls on different parts may run at the same time."""
        n_gt = obs.shape[1]
        uni = np.ascontiguousarray([uniform], dtype=np.longdouble)
        ch = np.zeros((len(chains), 3), dtype=np.uint64)
        for i, (f, n, k) in enumerate(chains):
            ch[i] = (f, n, k)
        for a, t in ((keep, np.uint8), (obs, np.longdouble), (row, np.uint32), (restart, np.uint8), (pow_tables, np.longdouble),
                     (gid, np.uint8), (order, np.uint8), (fwd_step, np.uint64), (bwd_step, np.uint64), (prob, np.longdouble),
                     (winner, np.uint32)):
            assert a.dtype == t and a.flags.c_contiguous
        self._chk(self._l.vgmi_hmm_calls_part(self._h, n_gt, ploidy, _ptr(keep), keep.shape[0], _ptr(obs), rows[0], rows[1], _ptr(row),
                                               _ptr(restart), _ptr(pow_tables), steps[0], steps[1], _ptr(uni), _ptr(ch), len(chains), _ptr(gid),
                                               _ptr(order), _ptr(fwd_step), _ptr(bwd_step), _ptr(prob), _ptr(winner)))

    def bloom_save_file(self, path):
        self._chk(self._l.vgmi_bloom_save_file(self._h, os.fsencode(path)))

    def bloom_load_file(self, path):
        self._chk(self._l.vgmi_bloom_load_file(self._h, os.fsencode(path)))
        self._bloom_m = os.path.getsize(path) - 12 - 8 * int.from_bytes(open(path, 'rb').read(12)[8:12], 'little')

    def bloom_fetch(self):
        out = np.empty(self._bloom_m, dtype=np.uint8)
        self._chk(self._l.vgmi_bloom_fetch(self._h, _ptr(out)))
        return out

    def bloom_load(self, filt):
        filt = np.ascontiguousarray(filt, dtype=np.uint8)
        assert filt.size == self._bloom_m
        self._chk(self._l.vgmi_bloom_load(self._h, _ptr(filt)))

    def bloom_query(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        mn = np.empty(keys.size, dtype=np.uint8)
        nz = np.empty(keys.size, dtype=np.uint8)
        self._chk(self._l.vgmi_bloom_query(self._h, _ptr(keys), keys.size, _ptr(mn), _ptr(nz)))
        return mn, nz

    # ---- tooling
    def synth_reads_device(self, seed, first_read, n_reads, read_len, dev_hap_cat, hap_off, dev_out):
        """(libvgsynth.so, on this context's device and main stream)"""
        from . import synthlib
        dev = C.c_int()
        self._chk(self._l.vgmi_device_of(self._h, C.byref(dev)))
        synthlib.reads_device(dev.value, self.stream, seed, first_read, n_reads, read_len, dev_hap_cat, hap_off, dev_out)
