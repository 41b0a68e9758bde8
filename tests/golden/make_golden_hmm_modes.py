#!/usr/bin/env python3
"""Extra genotype modes for the host HMM tests: runs the UNMODIFIED reference (deterministic build,
oracle/_ref/varigraph_det: std::random_device pinned to 20241022) on the committed cohort_sv / cohort_snp inputs and
stores the decompressed VCFs next to the existing fixtures.  Run in the build container only (needs oracle/_ref)."""
import gzip, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLI_DET = os.path.join(ROOT, "oracle", "_ref", "varigraph_det")

MODES = {   # file suffix -> (cohort, extra CLI arguments, samples)
    "fre": ("cohort_sv", ["-m", "fre"], 1),
    "sv": ("cohort_sv", ["--sv"], 1),
    "minsupport": ("cohort_sv", ["--min-support", "30"], 1),
    "gran": ("cohort_sv", ["--granularity", "0.02"], 1),
    "fre_n5": ("cohort_snp", ["-m", "fre", "-n", "5"], 1),
    "two_n5": ("cohort_snp", ["-n", "5"], 2),      # same reads twice: the pruned node lists persist across samples
}


def main():
    for name, (cohort, extra, n_samples) in MODES.items():
        d = os.path.join(HERE, cohort)
        with tempfile.TemporaryDirectory() as w:
            graph = os.path.join(w, "graph.bin")
            open(graph, "wb").write(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
            fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
            with open(os.path.join(w, "samples.cfg"), "w") as f:
                for s in range(n_samples):
                    f.write(f"sample{s} " + " ".join(fq) + "\n")
            subprocess.run([CLI_DET, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", "4"] + extra, cwd=w,
                           check=True, capture_output=True)
            for s in range(n_samples):
                txt = gzip.open(os.path.join(w, f"sample{s}.varigraph.vcf.gz"), "rb").read()
                suffix = name if n_samples == 1 else f"{name}_s{s}"
                open(os.path.join(d, f"expected_{suffix}.vcf"), "wb").write(txt)
                print(cohort, suffix, txt.count(b"\n"), "lines")


if __name__ == "__main__":
    main()
