#!/usr/bin/env python3
"""Diagnostic: total table hits of the C2 bench workload (raw 32-bit counters), per read."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import bench
from varigraph_amd import vgmi

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = vgmi.Context(0, buffer_mib=256)
g = bench.load_graph()
ctx.table_upload(g["keys"], g["k"])
haps = bench.cohort_haplotypes()
cat = np.concatenate(haps)
hap_off = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.uint64)
d_cat = torch.from_numpy(cat).cuda()
RL = bench.READ_LEN
d_block = torch.empty(n_reads * (RL + 1), dtype=torch.uint8, device="cuda")
ctx.synth_reads_device(1000, 0, n_reads, RL, d_cat, hap_off, d_block)
ctx.counts_reset()
ctx.reads_submit_device(d_block, n_reads * (RL + 1), n_reads)
d = torch.zeros(len(g["keys"]), dtype=torch.int32, device="cuda")
ctx.counts_export_device(d)
torch.cuda.synchronize()
tot = int(d.to(torch.int64).sum())
print({"reads": n_reads, "hits": tot, "hits_per_read": tot / n_reads, "max_count": int(d.max())})
