#!/usr/bin/env python3
"""Ordinary gzip FASTQ -> text on the device (vgmi_gunzip_buffer: host memory to host memory, PCIe both ways), alone, for profiling."""
import gzip, json, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from varigraph_amd import vgmi
    rng = np.random.default_rng(1)
    L = 150
    genome = rng.integers(0, 4, size=1_000_000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    starts = rng.integers(0, genome.size - L, size=n_reads)
    rows = acgt[genome[starts[:, None] + np.arange(L)[None, :]]]
    m = np.empty((n_reads, 14 + L + 3 + L + 1), dtype=np.uint8)
    m[:, 0], m[:, 1] = ord("@"), ord("r")
    idx = np.arange(n_reads, dtype=np.int64)
    for d in range(9):
        m[:, 2 + d] = (idx // 10 ** (8 - d)) % 10 + ord("0")
    m[:, 11], m[:, 12], m[:, 13] = ord("/"), ord("1"), 10
    m[:, 14:14 + L] = rows
    m[:, 14 + L], m[:, 15 + L], m[:, 16 + L] = 10, ord("+"), 10
    m[:, 17 + L:17 + 2 * L] = ord("I")
    m[:, 17 + 2 * L] = 10
    text = m.tobytes()
    t0 = time.perf_counter()
    comp = gzip.compress(text, level)
    t_c = time.perf_counter() - t0
    ctx = vgmi.Context(0, buffer_mib=16)
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        got, cons, fin, why = ctx.gunzip(comp, len(text) + 4096)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    t0 = time.perf_counter()
    ref = zlib.decompress(comp, 31)
    t_z = time.perf_counter() - t0
    print(json.dumps({"n_reads": n_reads, "level": level, "text_bytes": len(text), "compressed_bytes": len(comp), "identical": got == text, "member_end": fin,
                      "reason": why, "device_seconds_best_incl_pcie_and_malloc": best, "text_gb_per_s": len(text) / best / 1e9,
                      "reads_per_s": n_reads / best, "zlib_one_thread_seconds": t_z, "python_gzip_compress_s": t_c}))
    ctx.close()
if __name__ == "__main__":
    main()
