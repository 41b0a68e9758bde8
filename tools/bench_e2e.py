#!/usr/bin/env python3
"""End-to-end at a chosen scale, everything from files (SURVEY 8d level iii): synthetic reference + cohort VCF +
30x paired reads of VCF sample 0 -> `construct` by the UNMODIFIED reference and by varigraph-mi (graph.bin compared byte
for byte) -> `genotype` by (a) the all-CPU reference and (b) varigraph-mi (device counting, own HMM) -> VCFs compared
byte for byte.
Needs oracle/_ref/varigraph_det (test infrastructure; it travels to the GPU box as a prebuilt binary)."""
import argparse, gzip, json, os, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=10_000_000)
    ap.add_argument("--variants", type=int, default=80_000)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=10)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--keep", default="")
    ap.add_argument("--native-only", action="store_true", help="skip the reference runs (no comparison)")
    ap.add_argument("--samples", type=int, default=1, help="the same reads as N samples (native-only runs)")
    ap.add_argument("--gpus", default="0", help="device list of the native run, e.g. 0,0,0,0")
    ap.add_argument("--ploidy", type=int, default=2, help="ploidy of the VCF samples and of the sequenced sample")
    ap.add_argument("--vcf-samples", type=int, default=7)
    ap.add_argument("--indel", type=float, default=0.0)
    ap.add_argument("--sv", type=float, default=0.0)
    ap.add_argument("--matrix", default="", help="after the ordinary runs: the native genotype again per configuration on the same files, "
                    "';'-separated lists of 't=<threads>' / 'samples=<n>' / NAME=value (environment); times and the VCF's md5 per run")
    ap.add_argument("--repeat", type=int, default=1, help="runs per --matrix configuration (all reported)")
    args = ap.parse_args()
    from varigraph_amd import synth, vgmi
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "varigraph_det")
    cli = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
    work = args.keep or tempfile.mkdtemp(prefix="vg_e2e_")
    os.makedirs(work, exist_ok=True)
    out = {"genome": args.genome, "variants": args.variants, "pairs": args.pairs, "threads": args.threads, "gz": args.gz}
    try:
        t0 = time.perf_counter()
        ref = synth.make_reference(args.genome)
        variants, gts = synth.make_cohort(ref, args.variants, n_samples=args.vcf_samples, ploidy=args.ploidy, seed=11,
                                          indel_frac=args.indel, sv_frac=args.sv)
        fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
        synth.write_fasta(fa, "chr1", ref)
        synth.write_vcf(vcf, "chr1", len(ref), variants, gts, args.vcf_samples, args.ploidy)
        haps = synth.sample_haplotypes(ref, variants, gts, 0, args.ploidy)
        block = vgmi.synth_reads_host(1000, 0, 2 * args.pairs, 150, haps)
        fq = synth.write_fastq_pair(os.path.join(work, "s"), block, 2 * args.pairs, 150, gz=args.gz)
        out["synth_s"] = time.perf_counter() - t0
        graph = os.path.join(work, "graph.bin")
        env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
        if not args.native_only:
            t0 = time.perf_counter()
            r = subprocess.run([ref_bin, "construct", "-r", fa, "-v", vcf, "--save-graph", graph, "-t", "32", "--vcf-ploidy", str(args.ploidy)], cwd=work,
                               capture_output=True, text=True)
            out["reference_construct_s"] = time.perf_counter() - t0
            if r.returncode != 0:
                out["error"] = r.stderr[-400:]
                print(json.dumps(out))
                return
            out["graph_bytes"] = os.path.getsize(graph)
        graph_native = os.path.join(work, "graph_native.bin")
        t0 = time.perf_counter()
        r = subprocess.run([cli, "construct", "-r", fa, "-v", vcf, "--save-graph", graph_native, "-t", "32", "--gpu", "0", "--vcf-ploidy", str(args.ploidy)], cwd=work,
                           capture_output=True, text=True, env=env)
        out["native_construct_s"] = time.perf_counter() - t0
        if r.returncode != 0:
            out["native_construct_error"] = r.stderr[-400:]
        else:
            out["native_construct_log"] = [ln for ln in r.stderr.strip().split("\n") if "varigraph-mi]" in ln][-2:]
            if args.native_only:
                graph = graph_native
            else:
                out["graph_identical"] = subprocess.run(["cmp", "-s", graph, graph_native]).returncode == 0
        vcfs = {}
        for name, exe, extra in (("reference_cpu", ref_bin, []), ("native_cli", cli, ["--gpus", args.gpus])):
            if args.native_only and name == "reference_cpu":
                continue
            d = os.path.join(work, name)
            os.makedirs(d, exist_ok=True)
            n_samples = args.samples if name == "native_cli" else 1
            open(os.path.join(d, "samples.cfg"), "w").write("".join(f"sample{i} " + " ".join(fq) + "\n" for i in range(n_samples)))
            t0 = time.perf_counter()
            r = subprocess.run([exe, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", str(args.threads)] + extra +
                               (["--sample-ploidy", str(args.ploidy), "--use-depth"] if args.ploidy != 2 else []),
                               cwd=d, capture_output=True, text=True, env=env)
            out[name + "_genotype_s"] = time.perf_counter() - t0
            if r.returncode != 0:
                out[name + "_error"] = r.stderr[-400:]
                continue
            out[name + "_log_tail"] = [ln for ln in r.stderr.strip().split("\n") if "varigraph-mi]" in ln or "graph_index]" in ln or "[vgmi]" in ln][-160:]
            vcfs[name] = gzip.open(os.path.join(d, "sample0.varigraph.vcf.gz"), "rb").read()
        if not args.native_only:
            out["vcf_identical"] = len(vcfs) == 2 and vcfs["reference_cpu"] == vcfs["native_cli"]
        out["vcf_records"] = vcfs.get("native_cli", b"").count(b"\n")
        if args.matrix:
            import hashlib
            out["matrix"] = []
            for cfg in [c for c in args.matrix.split(";") if c.strip()]:
                env2, threads, n_samples = dict(env), args.threads, args.samples
                for item in cfg.split(","):
                    name, _, val = item.strip().partition("=")
                    if name == "t":
                        threads = int(val)
                    elif name == "samples":
                        n_samples = int(val)
                    elif name:
                        env2[name] = val
                d = os.path.join(work, "matrix")
                shutil.rmtree(d, ignore_errors=True)
                os.makedirs(d)
                open(os.path.join(d, "samples.cfg"), "w").write("".join(f"sample{i} " + " ".join(fq) + "\n" for i in range(n_samples)))
                for _ in range(args.repeat):
                    t0 = time.perf_counter()
                    r = subprocess.run([cli, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", str(threads), "--gpus", args.gpus] +
                                       (["--sample-ploidy", str(args.ploidy), "--use-depth"] if args.ploidy != 2 else []),
                                       cwd=d, capture_output=True, text=True, env=env2)
                    dt = time.perf_counter() - t0
                    row = {"config": cfg, "threads": threads, "samples": n_samples, "genotype_s": dt, "rc": r.returncode}
                    if r.returncode == 0:
                        lines = [ln for ln in r.stderr.strip().split("\n") if "varigraph-mi]" in ln or "graph_index]" in ln]
                        row["done_in"] = next((ln.split("done in")[1].strip() for ln in reversed(lines) if "done in" in ln), None)
                        row["loaded"] = next((ln[ln.rfind("(") + 1:ln.rfind(")")] for ln in lines if "graph loaded" in ln), None)
                        row["vcf_md5"] = sorted({hashlib.md5(gzip.open(os.path.join(d, f"sample{i}.varigraph.vcf.gz"), "rb").read().replace(
                            f"sample{i}".encode(), b"S")).hexdigest() for i in range(n_samples)})
                        row["same_as_first_run"] = hashlib.md5(vcfs.get("native_cli", b"").replace(b"sample0", b"S")).hexdigest() in row["vcf_md5"] and len(row["vcf_md5"]) == 1
                        row["log"] = [ln for ln in lines if "HMM part" not in ln][-110:]
                    else:
                        row["error"] = r.stderr[-300:]
                    out["matrix"].append(row)
    finally:
        if not args.keep:
            shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
