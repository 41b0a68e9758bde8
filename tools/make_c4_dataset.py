#!/usr/bin/env python3
"""The files of the bench's c4 block (BASELINE configs[3] at the level a user runs it) written to a directory, so that `varigraph-mi`
can be started on them directly -- under rocprofv3, which must be handed the program itself: a chr20-scale reference + cohort VCF,
graph.bin by `varigraph-mi construct`, one sample's plain FASTQ pair, and a samples.cfg naming it N times.

  python tools/make_c4_dataset.py DIR [--samples 8] [--pairs 6000000]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=6_000_000)
    ap.add_argument("--genome", type=int, default=60_000_000)
    ap.add_argument("--variants", type=int, default=500_000)
    args = ap.parse_args()
    from varigraph_amd import synth
    os.makedirs(args.dir, exist_ok=True)
    ref = synth.make_reference(args.genome)
    var, gts = synth.make_cohort(ref, args.variants, n_samples=7, ploidy=2, seed=11)
    fa, vcf = os.path.join(args.dir, "ref.fa"), os.path.join(args.dir, "in.vcf")
    synth.write_fasta(fa, "chr1", ref)
    synth.write_vcf(vcf, "chr1", len(ref), var, gts, 7, 2)
    haps = synth.sample_haplotypes(ref, var, gts, 0, 2)
    fq = synth.write_fastq_pair_device(os.path.join(args.dir, "s"), haps, args.pairs, 1000)
    cli = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    subprocess.run([cli, "construct", "-r", fa, "-v", vcf, "--save-graph", os.path.join(args.dir, "graph.bin"), "-t", "32", "--gpu", "0"], check=True, env=env)
    run = os.path.join(args.dir, "run")
    os.makedirs(run, exist_ok=True)
    open(os.path.join(run, "samples.cfg"), "w").write("".join(f"sample{i} " + " ".join(fq) + "\n" for i in range(args.samples)))
    print(run)


if __name__ == "__main__":
    main()
