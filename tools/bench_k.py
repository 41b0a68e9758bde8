#!/usr/bin/env python3
"""Count-kernel rate for k != 27 (the reference runs one loop for every k <= 28, src/kmer.cpp:110-149, main.cpp:187):
the C2-shaped workload (1 Mb reference, 1 k SNPs, device-generated 2x150 bp reads) with graphs built for other k,
oracle-checked on a prefix in the same run."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="21,25,27,22,28")
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--genome", type=int, default=1_000_000)
    ap.add_argument("--variants", type=int, default=1000)
    ap.add_argument("--check", type=int, default=200_000)
    args = ap.parse_args()
    import torch
    from varigraph_amd import synth, vgmi
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib
    ctx = vgmi.Context(0, buffer_mib=64)
    out = []
    for k in [int(x) for x in args.ks.split(",")]:
        keys, haps = synth.snp_graph(args.genome, args.variants, k=k)
        ctx.table_upload(keys, k)
        cat = np.concatenate(haps)
        off = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.uint64)
        d_cat = torch.from_numpy(cat).cuda()
        n = args.reads
        d_block = torch.empty(n * 151, dtype=torch.uint8, device="cuda")
        for first in range(0, n, 8_000_000):
            m = min(8_000_000, n - first)
            ctx.synth_reads_device(7, first, m, 150, d_cat, off, d_block[first * 151:])
        d_off = (torch.arange(n + 1, dtype=torch.int64, device="cuda") * 151) if k % 2 == 0 else None
        best = None
        for _ in range(3):
            ctx.counts_reset()
            ctx.reads_submit_device(d_block, n * 151, n, d_off)
            ctx.counts_finish_device(None, None, None)
            ms, _ = ctx.count_kernel_ms()
            best = ms if best is None or ms < best else best
        m = args.check
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, m * 151, m, d_off[: m + 1] if d_off is not None else None)
        got, _, _ = ctx.counts_finish()
        t = oracle_lib.Table(keys)
        t.count_block(d_block[: m * 151].cpu().numpy(), k)
        row = {"k": k, "n_keys": int(keys.size), "reads": n, "kernel_ms": best, "reads_per_s": n / best * 1e3,
               "stream_gb_per_s": n * 151 / best / 1e6, "oracle_match": bool(np.array_equal(got, t.counts()))}
        print(json.dumps(row), flush=True)
        out.append(row)
        del d_block, d_cat


if __name__ == "__main__":
    main()
