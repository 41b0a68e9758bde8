#!/usr/bin/env python3
"""sha256 of the reference's FastqKmer::save_index file (src/fastq_kmer.cpp:200-238) for the committed cohorts that carry
their reads: oracle/_ref/ref_harness (the unmodified reference + oracle/ref_harness.cpp) counts the cohort's FASTQ files and
writes its own dump (VG_SAVE_READS_INDEX).  Run in the build container after `make -C oracle ref`; writes reads_index.json."""
import glob
import gzip
import hashlib
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref_harness")


def main():
    out = {}
    for name in ("cohort_snp", "cohort_sv"):
        d = os.path.join(HERE, name)
        fq = sorted(glob.glob(os.path.join(d, "reads_*.fq.gz")))
        assert len(fq) == 2, fq
        with tempfile.TemporaryDirectory() as work:
            graph = os.path.join(work, "graph.bin")
            open(graph, "wb").write(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
            idx = os.path.join(work, "reads.idx")
            subprocess.run([HARNESS, "count", graph, "2", os.path.join(work, "c.bin")] + fq, check=True, capture_output=True,
                           env=dict(os.environ, VG_SAVE_READS_INDEX=idx))
            data = open(idx, "rb").read()
            out[name] = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                         "read_base": int.from_bytes(data[:8], "little")}
    json.dump(out, open(os.path.join(HERE, "reads_index.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
