#!/usr/bin/env python3
"""Ordinary gzip FASTQ files -> counters, device gunzip (VGH_DEVICE_GUNZIP=1) against the host's inflate threads (=0), alone."""
import gzip, json, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    threads = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,16").split(",")]
    from varigraph_amd import host, synth, vgmi
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"], seed=meta["cohort_seed"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])
    work = tempfile.mkdtemp(prefix="vg_gz_")
    try:
        block = vgmi.synth_reads_host(1000, 0, n_reads, 150, haps)
        plain = synth.write_fastq_pair_fast(os.path.join(work, "s"), block, n_reads, 150, qual=os.environ.get("VG_BENCH_QUAL", "const"))
        gz = []
        for p in plain:
            with open(p, "rb") as fi, gzip.open(p + ".gz", "wb", compresslevel=level) as fo:
                shutil.copyfileobj(fi, fo, 1 << 24)
            gz.append(p + ".gz")
        g = host.Graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
        os.environ["VGH_HOST_PARSE"] = "0"
        ctx = vgmi.Context(0, buffer_mib=256)
        g.upload(ctx)
        ref_cov = None
        for dev in ("1", "0"):
            os.environ["VGH_DEVICE_GUNZIP"] = dev
            for t in threads:
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    cov, _, _, st = g.sample_count(ctx, gz, threads=t, require_depth=False)
                    dt = time.perf_counter() - t0
                    best = dt if best is None or dt < best else best
                if ref_cov is None:
                    ref_cov = cov
                print(json.dumps({"device_gunzip": dev, "threads": t, "n_reads": n_reads, "level": level, "gzip_reads_per_s": n_reads / best,
                                  "seg_kb": os.environ.get("VGMI_GZ_SEG_KB", "32"), "identical": bool((cov == ref_cov).all()), "n_counted": int(st["n_reads"]),
                                  "compressed_bytes": sum(os.path.getsize(p) for p in gz)}), flush=True)
        ctx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)
if __name__ == "__main__":
    main()
