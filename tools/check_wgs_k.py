#!/usr/bin/env python3
"""Whole-genome-class graph (3 Gb, 5 M SNPs) at another k: the counters of a prefix of reads against the oracle's emitter (vgo_sketch per
read, looked up by binary search in the sorted key list -- bench.py's c5 check, for k != 27).  Prints one JSON line per k.
  check_wgs_k.py 21,22 [reads]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import oracle_lib
    from varigraph_amd import synth, vgmi
    ks = [int(x) for x in sys.argv[1].split(",")]
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    ctx = vgmi.Context(0, buffer_mib=64)
    for k in ks:
        keys, (ref, hap1) = synth.snp_graph(3_000_000_000, 5_000_000, k=k)
        ctx.table_upload(keys, k)
        cat = np.concatenate([ref, hap1])
        off = np.array([0, len(ref), 2 * len(ref)], dtype=np.uint64)
        d_cat = torch.from_numpy(cat).cuda()
        d_block = torch.empty(m * 151, dtype=torch.uint8, device="cuda")
        ctx.synth_reads_device(4711, 0, m, 150, d_cat, off, d_block)
        d_off = (torch.arange(m + 1, dtype=torch.int64, device="cuda") * 151) if k % 2 == 0 else None
        ctx.counts_reset()
        ctx.reads_submit_device(d_block, m * 151, m, d_off)
        got, _, _ = ctx.counts_finish()
        rows = d_block.cpu().numpy().reshape(m, 151)
        emitted = np.concatenate([oracle_lib.sketch(rows[i, :150].tobytes(), k) for i in range(m)])
        emitted = emitted[emitted != np.uint64(0xFFFFFFFFFFFFFFFF)]
        pos = np.searchsorted(keys, emitted)
        pos[pos == keys.size] = 0
        idx, cnt = np.unique(pos[keys[pos] == emitted], return_counts=True)
        want = np.zeros(keys.size, dtype=np.uint8)
        want[idx] = np.minimum(cnt, 255).astype(np.uint8)
        print(json.dumps({"k": k, "graph_kmers": int(keys.size), "reads": m, "oracle_match": bool(np.array_equal(got, want)),
                          "cov_sum": int(got.astype(np.int64).sum()), "keys_nonzero": int((got != 0).sum()), "context_table": ctx.ctable_info()}), flush=True)
        del d_cat, d_block, keys, ref, hap1, cat


if __name__ == "__main__":
    main()
