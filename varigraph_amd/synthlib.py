"""ctypes binding of libvgsynth.so (csrc/bench/vgsynth.h): the seeded synthetic workloads of bench.py, tools/ and the tests.

BENCH / TEST TOOLING -- its own library, built next to the product's (build.build_synth) and never loaded by it: the reference ships no
data and no generator, so the workloads of BASELINE.json are produced here."""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvgsynth.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            from . import build
            build.build_synth()
        # torch wheels bundle their own libamdhip64; load torch FIRST so the process ends up with one HIP runtime (as vgmi.load_library
        # does): a process that loaded this library before torch found "No HIP GPUs are available" afterwards
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        l = C.CDLL(LIB_PATH)
        i32, u32, u64, vp = C.c_int, C.c_uint32, C.c_uint64, C.c_void_p
        for name, args in {"vgs_reads_device": [i32, vp, u64, u64, u64, u32, vp, vp, u32, vp], "vgs_reads_host": [u64, u64, u64, u32, vp, vp, u32, vp],
                           "vgs_reference_host": [u64, u64, vp], "vgs_snp_keys_host": [vp, u64, vp, vp, u64, u32, vp]}.items():
            fn = getattr(l, name)
            fn.restype = i32
            fn.argtypes = args
        _lib = l
    return _lib


def _p(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a)


def _chk(rc, what):
    if rc:
        raise RuntimeError(f"libvgsynth: {what} failed ({'invalid argument' if rc == -1 else 'a HIP call failed'})")


def reference(seed, length):
    out = np.empty(length, dtype=np.uint8)
    _chk(lib().vgs_reference_host(seed, length, _p(out)), "vgs_reference_host")
    return out


def snp_keys(ref, pos, alts, k=27):
    """Unsorted, possibly repeated keys of the k-mers covering SNP sites (csrc/bench/vgsynth.h)."""
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    alts = np.ascontiguousarray(alts, dtype=np.uint8)
    out = np.empty(2 * k * pos.size, dtype=np.uint64)
    _chk(lib().vgs_snp_keys_host(_p(ref), ref.size, _p(pos), _p(alts), pos.size, k, _p(out)), "vgs_snp_keys_host")
    return out


def reads_host(seed, first_read, n_reads, read_len, haps):
    """haps: list of uint8 arrays (ASCII haplotypes). Returns the '\\n'-joined read block (uint8)."""
    cat = np.ascontiguousarray(np.concatenate(haps))
    off = np.zeros(len(haps) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(h) for h in haps])
    out = np.empty(n_reads * (read_len + 1), dtype=np.uint8)
    _chk(lib().vgs_reads_host(seed, first_read, n_reads, read_len, _p(cat), _p(off), len(haps), _p(out)), "vgs_reads_host")
    return out


def reads_device(device, stream, seed, first_read, n_reads, read_len, dev_hap_cat, hap_off, dev_out):
    hap_off = np.ascontiguousarray(hap_off, dtype=np.uint64)
    _chk(lib().vgs_reads_device(device, stream, seed, first_read, n_reads, read_len, _p(dev_hap_cat), _p(hap_off), hap_off.size - 1, _p(dev_out)),
         "vgs_reads_device")
