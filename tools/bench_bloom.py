#!/usr/bin/env python3
"""K3 (construct-side counting Bloom update, make_mbf) at scale: a random reference sequence resident in HBM,
BloomFilter(n = G - k + 1, p = 0.01) geometry, 7 Murmur3 positions per emitted k-mer.  Reports k-mers/s and the
SURVEY 8d accounting (15 B per reference k-mer: 1 base + 7 byte read-modify-writes)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=60_000_000)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--query", type=int, default=0, help="also time K4 (BloomFilter::count / ::find) on this many random keys")
    ap.add_argument("--piece", type=int, default=0, help="add the sequence in pieces of this many bases (chromosomes), 0 = one call")
    args = ap.parse_args()
    import torch
    from varigraph_amd import vgmi
    G, k = args.genome, args.k
    m, nh = vgmi.bloom_params(G - k + 1, 0.01)
    ctx = vgmi.Context(0)
    seeds = np.arange(1, nh + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    gen = torch.Generator(device="cuda").manual_seed(7)
    codes = torch.randint(0, 4, (G,), generator=gen, device="cuda", dtype=torch.uint8)
    seq = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[codes.long()] if G <= 200_000_000 else None
    if seq is None:   # avoid the int64 index copy for multi-Gb sequences
        seq = codes
        seq.mul_(0).add_(65)  # placeholder, replaced below chunk-wise
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
        for a in range(0, G, 100_000_000):
            c = torch.randint(0, 4, (min(100_000_000, G - a),), generator=gen, device="cuda")
            seq[a:a + c.numel()] = lut[c]
    torch.cuda.synchronize()
    res = []
    for _ in range(args.steps + 1):
        ctx.bloom_create(m, nh, seeds)
        torch.cuda.synchronize()
        t = time.perf_counter()
        if args.piece:
            for a in range(0, G, args.piece):
                ctx.bloom_add_seq_device(seq[a:a + args.piece], min(args.piece, G - a), k)
        else:
            ctx.bloom_add_seq_device(seq, G, k)
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t)
    best = min(res[1:])
    n_kmers = G - k + 1
    out = {"genome": G, "k": k, "piece": args.piece, "bloom_bytes": m, "n_hash": nh, "seconds": best, "kmers_per_s": n_kmers / best,
           "accounting_GBps": 15.0 * n_kmers / best / 1e9, "filter_updates_per_s": nh * n_kmers / best}
    if args.query:
        qk = (np.random.default_rng(3).integers(0, 1 << (2 * k), size=args.query, dtype=np.uint64) << np.uint64(8)) | np.uint64(k)
        t = time.perf_counter()
        mn, nz = ctx.bloom_query(qk)
        out["query_keys_per_s_pcie_inclusive"] = args.query / (time.perf_counter() - t)
        out["query_all_nonzero"] = int(nz.sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
