#!/usr/bin/env python3
"""Plain and block-gzip FASTQ -> counters at several staging-buffer sizes (the chunk of the device-side parser / inflate)."""
import json, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,100,128,256,512").split(",")]
    from varigraph_amd import host, synth, vgmi
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"], seed=meta["cohort_seed"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])
    work = tempfile.mkdtemp(prefix="vg_chunk_")
    try:
        block = vgmi.synth_reads_host(1000, 0, n_reads, 150, haps)
        plain = synth.write_fastq_pair_fast(os.path.join(work, "s"), block, n_reads, 150)
        bgz = [synth.bgzf_compress_file(p, p + ".bgz.gz", level=4) for p in plain]
        g = host.Graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
        os.environ["VGH_HOST_PARSE"] = "0"
        for mib in sizes:
            ctx = vgmi.Context(0, buffer_mib=mib)
            g.upload(ctx)
            row = {"n_reads": n_reads, "buffer_mib": mib}
            for label, files in (("plain", plain), ("bgzf", bgz)):
                best = None
                for _ in range(4):
                    t0 = time.perf_counter()
                    g.sample_count(ctx, files, threads=8, require_depth=False)
                    dt = time.perf_counter() - t0
                    best = dt if best is None or dt < best else best
                row[label + "_reads_per_s"] = n_reads / best
            print(json.dumps(row), flush=True)
            ctx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)
if __name__ == "__main__":
    main()
