#!/usr/bin/env python3
"""Sample-level (end-to-end) rate: FASTQ files on disk -> counters on the host, through the C++
FastqKmerHip pipeline (inflate thread(s) -> parser thread per file -> pinned staging -> HIP) for plain, gzip and
block-gzip (BGZF) copies of the same reads, next to the unmodified reference on the same files.  SURVEY 8d metric level (ii)."""
import json
import os
import subprocess
import sys
import tempfile
import time
import gzip
import shutil

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    from varigraph_amd import host, synth, vgmi
    import numpy as np
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "c1", "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    variants, gts = synth.make_cohort(ref, meta["n_var"], n_samples=meta["n_samples"], ploidy=meta["ploidy"],
                                      seed=meta["cohort_seed"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, meta["ploidy"])
    work = tempfile.mkdtemp(prefix="vg_pipe_")
    out = {"n_reads": n_reads}
    try:
        block = vgmi.synth_reads_host(1000, 0, n_reads, 150, haps)
        plain = synth.write_fastq_pair(os.path.join(work, "s"), block, n_reads, 150, gz=False)
        gz = []
        for p in plain:
            with open(p, "rb") as fi, gzip.open(p + ".gz", "wb", compresslevel=4) as fo:
                shutil.copyfileobj(fi, fo, 1 << 24)
            gz.append(p + ".gz")
        bgz = [synth.bgzf_compress_file(p, p + ".bgz.gz", level=4) for p in plain]
        g = host.Graph(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"))
        ctx = vgmi.Context(0, buffer_mib=128)
        g.upload(ctx)
        ref_cov = None
        for parse in ("device", "host"):   # device: vgmi_fastq_* (records found on the GPU); host: parser thread per file
            os.environ["VGH_HOST_PARSE"] = "1" if parse == "host" else "0"
            for label, files, threads in (("plain_t2", plain, 2), ("plain_t8", plain, 8), ("plain_t16", plain, 16), ("gz_t2", gz, 2),
                                          ("gz_t8", gz, 8), ("gz_t16", gz, 16), ("bgzf_t4", bgz, 4), ("bgzf_t16", bgz, 16)):
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    cov, _, _, st = g.sample_count(ctx, files, threads=threads, require_depth=False)
                    dt = time.perf_counter() - t0
                    best = dt if best is None or dt < best else best
                out[f"{parse}_parse_{label}_reads_per_s"] = n_reads / best
                out[f"{parse}_parse_{label}_kernel_s"] = st["seconds_kernel"]
                if ref_cov is None:
                    ref_cov = cov
                assert np.array_equal(cov, ref_cov)
        out["plain_bytes_per_read"] = sum(os.path.getsize(p) for p in plain) / n_reads
        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
        if os.path.exists(harness):
            graph = os.path.join(work, "graph.bin")
            with open(graph, "wb") as f:
                f.write(gzip.open(os.path.join(ROOT, "tests", "golden", "c1", "graph.bin.gz"), "rb").read())
            for label, files in (("plain", plain), ("gz", gz)):
                r = subprocess.run([harness, "count", graph, "10", os.path.join(work, "c.bin")] + files,
                                   capture_output=True, text=True)
                vals = dict(ln.split(" ", 1) for ln in r.stdout.splitlines() if " " in ln)
                out[f"reference_t10_{label}_reads_per_s"] = n_reads / float(vals["build_fastq_index_s"])
            import graphbin_py  # noqa
    except ImportError:
        pass
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    main()
