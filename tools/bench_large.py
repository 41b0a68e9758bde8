#!/usr/bin/env python3
"""Large-table performance probe (BASELINE configs 3-5 class): synthetic SNP graph key set built
by varigraph_amd.synth.snp_kmer_keys (NOT by the reference construct), reads generated on the device.
Reports kernel ms and reads/s of the count kernel for the given genome/variant sizes."""
import argparse
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=60_000_000)
    ap.add_argument("--variants", type=int, default=500_000)
    ap.add_argument("--reads", type=int, default=24_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", type=int, default=0, help="verify against the oracle on this many reads")
    ap.add_argument("--k", type=int, default=27, help="27, or 19 .. 25 (round 5: the context table with flanks of k - 16 bases; VGMI_CTABLE_K=0: the generic kernel)")
    args = ap.parse_args()
    import torch
    from varigraph_amd import synth, vgmi
    t0 = time.time()
    keys, (ref, hap1) = synth.snp_graph(args.genome, args.variants, k=args.k)
    print(f"graph: {len(keys)} keys built in {time.time() - t0:.1f}s", file=sys.stderr)
    ctx = vgmi.Context(0, buffer_mib=64)
    t0 = time.time()
    ctx.table_upload(keys, args.k)
    t_upload = time.time() - t0
    info = ctx.table_info()
    cat = np.concatenate([ref, hap1])
    off = np.array([0, len(ref), 2 * len(ref)], dtype=np.uint64)
    d_cat = torch.from_numpy(cat).cuda()
    n_reads = args.reads
    d_block = torch.empty(n_reads * 151, dtype=torch.uint8, device="cuda")
    chunk = 8_000_000
    for first in range(0, n_reads, chunk):
        n = min(chunk, n_reads - first)
        ctx.synth_reads_device(99, first, n, 150, d_cat, off, d_block[first * 151:])
    d_cov = torch.empty(len(keys), dtype=torch.uint8, device="cuda")
    d_off = (torch.arange(n_reads + 1, dtype=torch.int64, device="cuda") * 151) if args.k % 2 == 0 else None      # even k: the reads' offsets
    res = []
    for _ in range(args.steps + 1):
        ctx.counts_reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        ctx.reads_submit_device(d_block, n_reads * 151, n_reads, d_off)
        ctx.counts_finish_device(d_cov, None, None)
        dt = time.perf_counter() - t
        ms, _ = ctx.count_kernel_ms()
        res.append((dt, ms))
    cov = d_cov.cpu().numpy()
    hits = int(cov.astype(np.int64).sum())
    best = min(r[1] for r in res[1:])
    out = {"k": args.k, "genome": args.genome, "variants": args.variants, "n_keys": int(len(keys)), "table_slots": info["n_slots"],
           "filter_bits": info["filter_bits"], "reads": n_reads, "kernel_ms": best, "reads_per_s": n_reads / best * 1e3,
           "counted_hits_clamped": hits, "hits_per_read_lower_bound": hits / n_reads, "table_upload_s": t_upload,
           "context_table": ctx.ctable_info(), "grid_table": ctx.xtable_info(), "all_kernel_ms": [r[1] for r in res]}
    if args.check:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import oracle_lib
        m = args.check
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, m * 151, m, d_off[: m + 1] if d_off is not None else None)
        c2, _, _ = ctx.counts_finish()
        t = oracle_lib.Table(keys)
        t.count_block(d_block[: m * 151].cpu().numpy(), args.k)
        out["oracle_match"] = bool(np.array_equal(c2, t.counts()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
