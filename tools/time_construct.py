#!/usr/bin/env python3
"""Phase times of `varigraph-mi construct` (VGH_TIMING=1) on a synthetic cohort; no reference run."""
import argparse, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=10_000_000)
    ap.add_argument("--variants", type=int, default=80_000)
    ap.add_argument("--threads", type=int, default=32)
    a = ap.parse_args()
    from varigraph_amd import synth
    work = tempfile.mkdtemp(prefix="vg_ct_")
    try:
        ref = synth.make_reference(a.genome)
        variants, gts = synth.make_cohort(ref, a.variants, n_samples=7, ploidy=2, seed=11)
        fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
        synth.write_fasta(fa, "chr1", ref)
        synth.write_vcf(vcf, "chr1", len(ref), variants, gts, 7, 2)
        cli = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
        env = dict(os.environ, VGH_TIMING="1", VGH_RANDOM_DEVICE_VALUE="20241022")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "construct", "-r", fa, "-v", vcf, "--save-graph", os.path.join(work, "g.bin"), "-t", str(a.threads)],
                           capture_output=True, text=True, env=env)
        print("wall %.2f s rc=%d" % (time.perf_counter() - t0, r.returncode))
        print("\n".join(ln for ln in r.stderr.split("\n") if ln.startswith("[construct]") or "varigraph-mi]" in ln))
        print(subprocess.run(["md5sum", os.path.join(work, "g.bin")], capture_output=True, text=True).stdout.split()[0])
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
