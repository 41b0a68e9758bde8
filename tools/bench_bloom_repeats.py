#!/usr/bin/env python3
"""K3 (Bloom build) on random and on repetitive references: tandem repeats put one k-mer at many of a wave's positions.
VGMI_DBG=128 switches the per-wave privatisation off for comparison."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varigraph_amd import vgmi
rng = np.random.default_rng(1)
G = 200_000_000
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
seqs = {
    "random": acgt[rng.integers(0, 4, size=G)],
    "tandem300": np.tile(acgt[rng.integers(0, 4, size=300)], G // 300),
    "polyA+AT": np.concatenate([np.full(G // 2, 65, np.uint8), np.tile(np.frombuffer(b"AT", np.uint8), G // 4)]),
    "satellite171 5% + random": None,
}
sat = np.tile(acgt[rng.integers(0, 4, size=171)], (G // 20) // 171)
r = acgt[rng.integers(0, 4, size=G - sat.size)]
seqs["satellite171 5% + random"] = np.concatenate([r[: r.size // 2], sat, r[r.size // 2:]])
ctx = vgmi.Context(0, buffer_mib=64)
for name, s in seqs.items():
    n = s.size - 27 + 1
    m, nh = vgmi.bloom_params(n, 0.01)
    seeds = np.arange(1, nh + 1, dtype=np.uint64) * 7919
    best = 1e9
    for _ in range(3):
        ctx.bloom_create(m, nh, seeds)
        t = time.perf_counter(); ctx.bloom_add_seq(s, 27); dt = time.perf_counter() - t
        best = min(best, dt)
    f = ctx.bloom_fetch()
    print(f"{name}: {best*1e3:.1f} ms ({n/best/1e9:.2f}e9 k-mers/s), sum {int(f.astype(np.uint64).sum())}, max {int(f.max())}")
