#!/usr/bin/env python3
"""Block-gzip FASTQ files -> counters (device inflate + device parser), alone, for profiling."""
import json, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from varigraph_amd import host, synth, vgmi
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"], seed=meta["cohort_seed"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])
    work = tempfile.mkdtemp(prefix="vg_bgzf_")
    try:
        block = vgmi.synth_reads_host(1000, 0, n_reads, 150, haps)
        plain = synth.write_fastq_pair_fast(os.path.join(work, "s"), block, n_reads, 150, qual=os.environ.get("VG_BENCH_QUAL", "const"))
        bgz = [synth.bgzf_compress_file(p, p + ".bgz.gz", level=level) for p in plain]
        g = host.Graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
        os.environ["VGH_HOST_PARSE"] = "0"
        for mib in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "128").split(",")]:
            ctx = vgmi.Context(0, buffer_mib=mib)
            g.upload(ctx)
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                cov, _, _, st = g.sample_count(ctx, bgz, threads=8, require_depth=False)
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            print(json.dumps({"n_reads": n_reads, "level": level, "buffer_mib": mib, "bgzf_reads_per_s": n_reads / best,
                              "compressed_bytes": sum(os.path.getsize(p) for p in bgz), "text_bytes": sum(os.path.getsize(p) for p in plain)}), flush=True)
            ctx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)
if __name__ == "__main__":
    main()
