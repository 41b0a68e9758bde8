#!/usr/bin/env python3
"""Drawn end-to-end cases through both CLIs (the unmodified reference, oracle/_ref/varigraph_det, and varigraph-mi) on one GPU box:
genome size, variant mix, cohort size and ploidy, k, construct mode, reads per sample and every genotype option are drawn from a seed;
graph.bin and every VCF must be byte-identical, or both must refuse.  Prints one line per case and the parameters of any difference.
  fuzz_cli_parity.py <first seed> <cases>"""
import gzip, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("tc", os.path.join(ROOT, "tests", "test_gpu_configs.py"))
tc = importlib.util.module_from_spec(spec); spec.loader.exec_module(tc)
from varigraph_amd import synth


def dress(path, rng):
    """A FASTQ file as sequencers and pipelines leave them (round 5): some reads with N, some in lower case, some cut short (ragged
    lengths), then plain / gzip at a drawn level / gzip of several members / block gzip.  Returns the path to put into samples.cfg."""
    import zlib
    lines = open(path, "rb").read().split(b"\n")
    n_rec = len(lines) // 4
    for r in rng.choice(n_rec, size=max(1, n_rec // 40), replace=False):
        seq, qual = bytearray(lines[4 * r + 1]), lines[4 * r + 3]
        what = int(rng.integers(0, 4))
        if what == 0:
            for pos in rng.integers(0, len(seq), size=int(rng.integers(1, 4))): seq[int(pos)] = ord("N")
        elif what == 1:
            seq = bytearray(bytes(seq).lower())
        elif what == 2:
            cut = int(rng.integers(1, len(seq)))
            seq, qual = seq[:cut], qual[:cut]
        else:
            a = int(rng.integers(0, len(seq) - 2))
            seq[a:a + 2] = b"NN"
        lines[4 * r + 1], lines[4 * r + 3] = bytes(seq), qual
    text = b"\n".join(lines)
    form = int(rng.integers(0, 5))
    if form == 0:
        open(path, "wb").write(text)
        return path
    if form == 1 or form == 2:
        level = int(rng.integers(1, 10))
        open(path + ".gz", "wb").write(gzip.compress(text, compresslevel=level))
        os.remove(path)
        return path + ".gz"
    if form == 3:      # several members, cut anywhere
        cuts = sorted(set([0, len(text)] + [int(x) for x in rng.integers(0, len(text), size=int(rng.integers(1, 6)))]))
        with open(path + ".gz", "wb") as f:
            for a, b in zip(cuts[:-1], cuts[1:]): f.write(gzip.compress(text[a:b], compresslevel=6))
        os.remove(path)
        return path + ".gz"
    open(path, "wb").write(text)
    synth.bgzf_compress_file(path, path + ".gz", level=int(rng.integers(1, 10)), block=int(rng.integers(2000, 0xff01)))
    os.remove(path)
    return path + ".gz"


FORCE = {}      # --k K / --genome G on the command line: the drawn value replaced (round 6: campaigns of k = 28 over graphs on either side of 65 536 k-mers)


def case(seed):
    rng = np.random.default_rng(seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    genome = int(pick([40_000, 90_000, 200_000, 400_000, 1_500_000, 1_500_000]))      # (the last: more than 65 536 k-mers with dense variants -- the context table at any k)
    vploidy = pick([2, 2, 2, 3, 4])
    n_samples = pick([1, 2, 3, 5, 7]) if vploidy <= 2 else pick([1, 2, 3])
    k = pick([27, 27, 27, 21, 25, 22, 28, 15, 11, 19, 23, 20, 24, 26])
    k = FORCE.get("k", k)
    genome = FORCE.get("genome", genome)
    if k < 19 and genome > 400_000: genome = 400_000      # (the reference takes minutes over 1.5 Mb of 11- and 15-mers)
    copts = pick([[], [], ["--fast"], ["--use-unique-kmers"]])
    n_var = max(5, genome // pick([300, 600, 1500]))
    indel, sv = pick([0.0, 0.1, 0.3]), pick([0.0, 0.01, 0.05])
    sploidy = pick([2, 2, 2, 3, 4])
    gopts = ["-g", pick(["het", "het", "hom"]), "-m", pick(["rec", "fre"]), "--sample-ploidy", str(sploidy), "-n", str(pick([1, 2, 4, 7, 15, 15, 40]))]
    if rng.random() < 0.3: gopts += ["--sv"]
    if rng.random() < 0.4: gopts += ["--use-depth"]
    if rng.random() < 0.3: gopts += ["--min-support", str(pick([5, 30, 80]))]
    if rng.random() < 0.5: gopts += ["--granularity", str(pick([0.003, 0.02, 0.1]))]
    n_reads_samples = pick([1, 1, 2, 3])
    pairs = int(pick([3_000, 15_000, 40_000]))
    desc = f"seed {seed}: genome {genome}, {n_var} variants (indel {indel}, sv {sv}), cohort {n_samples} x ploidy {vploidy}, k {k}, construct {copts}, {n_reads_samples} samples x {pairs} pairs, genotype {' '.join(gopts)}"
    work = tempfile.mkdtemp(prefix="fuzz_")
    try:
        ref = synth._ACGT[rng.integers(0, 4, size=genome)]
        variants, gts = synth.make_cohort(ref, n_var, n_samples=n_samples, ploidy=vploidy, seed=int(seed) + 1, indel_frac=indel, sv_frac=sv)
        fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
        synth.write_fasta(fa, "chr1", ref); synth.write_vcf(vcf, "chr1", len(ref), variants, gts, n_samples, vploidy)
        graphs, rcs = {}, {}
        for name, exe, more in (("native", tc.CLI, ["--gpu", "0"]), ("cpu", tc.REF, [])):
            graphs[name] = os.path.join(work, f"g_{name}.bin")
            try:
                r = subprocess.run([exe, "construct", "-r", fa, "-v", vcf, "--save-graph", graphs[name], "-t", "6", "-k", str(k), "--vcf-ploidy", str(vploidy)] + copts + more,
                                   cwd=work, capture_output=True, text=True, env=tc.ENV, timeout=150)
            except subprocess.TimeoutExpired:
                return f"TIMEOUT of {name} (construct)", desc
            rcs[name] = r.returncode
        if (rcs["native"] == 0) != (rcs["cpu"] == 0): return "CONSTRUCT STATUS DIFFERS " + str(rcs), desc
        if rcs["cpu"] != 0: return "both refuse construct", desc
        if open(graphs["native"], "rb").read() != open(graphs["cpu"], "rb").read(): return "GRAPH DIFFERS", desc
        cfg = ""
        for i in range(n_reads_samples):
            who = int(rng.integers(0, n_samples))
            haps = synth.sample_haplotypes(ref, variants, gts, who, vploidy)
            fq = tc._write_fastq(os.path.join(work, f"s{i}"), haps, pairs, seed=int(seed) * 10 + i)
            fq = [dress(f, rng) for f in fq]
            cfg += f"ind{i} " + " ".join(fq) + "\n"
        outs, codes = {}, {}
        for name, exe, more in (("native", tc.CLI, ["--gpu", "0"]), ("cpu", tc.REF, [])):
            d = os.path.join(work, name); os.makedirs(d)
            open(os.path.join(d, "samples.cfg"), "w").write(cfg)
            try:
                r = subprocess.run([exe, "genotype", "--load-graph", graphs["cpu"], "-s", "samples.cfg", "-t", "6"] + gopts + more, cwd=d, capture_output=True, text=True, env=tc.ENV, timeout=150)
            except subprocess.TimeoutExpired:
                return f"TIMEOUT of {name}", desc
            codes[name] = r.returncode
            if r.returncode != 0:
                codes[name + "_stderr"] = r.stderr.strip().split("\n")[-1][:300]
            if r.returncode == 0:
                outs[name] = [gzip.open(os.path.join(d, f"ind{i}.varigraph.vcf.gz"), "rb").read() for i in range(n_reads_samples)]
        if (codes["native"] == 0) != (codes["cpu"] == 0): return "GENOTYPE STATUS DIFFERS " + str(codes), desc
        if codes["cpu"] != 0: return "both refuse genotype", desc
        for i in range(n_reads_samples):
            if outs["native"][i] != outs["cpu"][i]:
                a, b = outs["native"][i].split(b"\n"), outs["cpu"][i].split(b"\n")
                first = next((j for j, (x, y) in enumerate(zip(a, b)) if x != y), -1)
                return f"VCF {i} DIFFERS at line {first}: {a[first][:160] if first >= 0 else ''} | {b[first][:160] if first >= 0 else ''}", desc
        return f"identical ({sum(o.count(10) for o in outs['cpu'])} VCF lines)", desc
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    first, n = int(sys.argv[1]), int(sys.argv[2])
    for i, a in enumerate(sys.argv):
        if a in ("--k", "--genome"): FORCE[a[2:]] = int(sys.argv[i + 1])
    bad = 0
    for s in range(first, first + n):
        t0 = time.time()
        verdict, desc = case(s)
        flag = verdict.isupper() or "DIFFERS" in verdict or "TIMEOUT of native" in verdict      # (the reference's own pool loses a wake-up now and then: skipped, not a difference)
        bad += flag
        print(("!! " if flag else "ok ") + verdict + f" [{time.time() - t0:.1f} s] -- " + desc, flush=True)
    print(f"{n} cases, {bad} differences")
    sys.exit(1 if bad else 0)
