#!/usr/bin/env python3
"""`varigraph-mi construct` against the unmodified reference's `construct` (deterministic build, random_device pinned to
20241022): same FASTA + VCF -> graph.bin compared byte for byte.  Needs a GPU (the Bloom filter lives on the device) and
oracle/_ref/varigraph_det.  Inputs: the committed cohort VCFs + their regenerated references, or a synthetic cohort."""
import argparse, hashlib, json, os, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=1_000_000)
    ap.add_argument("--variants", type=int, default=5_000)
    ap.add_argument("--indel", type=float, default=0.2)
    ap.add_argument("--sv", type=float, default=0.05)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--ploidy", type=int, default=2)
    ap.add_argument("--samples", type=int, default=7)
    ap.add_argument("--extra", default="")
    args = ap.parse_args()
    from varigraph_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "varigraph_det")
    cli = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
    work = tempfile.mkdtemp(prefix="vg_con_")
    out = dict(vars(args))
    try:
        ref = synth.make_reference(args.genome)
        variants, gts = synth.make_cohort(ref, args.variants, n_samples=args.samples, ploidy=args.ploidy, seed=21,
                                          indel_frac=args.indel, sv_frac=args.sv)
        fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
        synth.write_fasta(fa, "chr1", ref)
        synth.write_vcf(vcf, "chr1", len(ref), variants, gts, args.samples, args.ploidy)
        extra = args.extra.split() if args.extra else []
        common = ["construct", "-r", fa, "-v", vcf, "-k", str(args.k), "--vcf-ploidy", str(args.ploidy)] + extra
        env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
        digests = {}
        for name, exe, more in (("reference", ref_bin, ["-t", "16"]), ("native", cli, ["--gpu", "0", "-t", "16"])):
            g = os.path.join(work, name + ".bin")
            t0 = time.perf_counter()
            r = subprocess.run([exe] + common + ["--save-graph", g] + more, cwd=work, capture_output=True, text=True, env=env,
                               timeout=3000)
            out[name + "_s"] = time.perf_counter() - t0
            if r.returncode != 0:
                out[name + "_error"] = r.stderr[-500:]
                continue
            out[name + "_bytes"] = os.path.getsize(g)
            digests[name] = hashlib.sha256(open(g, "rb").read()).hexdigest()
            if name == "native":
                out["native_log"] = [ln for ln in r.stderr.strip().split("\n") if "varigraph-mi]" in ln][-2:]
        out["identical"] = len(digests) == 2 and digests["reference"] == digests["native"]
        if len(digests) == 2 and not out["identical"]:
            a, b = open(os.path.join(work, "reference.bin"), "rb").read(), open(os.path.join(work, "native.bin"), "rb").read()
            n = min(len(a), len(b))
            first = next((i for i in range(n) if a[i] != b[i]), n)
            out["first_difference_at"] = first
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))
    return 0 if out.get("identical") else 1


if __name__ == "__main__":
    sys.exit(main())
