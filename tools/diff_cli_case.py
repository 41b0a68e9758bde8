#!/usr/bin/env python3
"""One case of tests/test_gpu_configs.py::test_other_cohort_ploidies_and_kmer_lengths_identical by hand, with the lines that differ:
  diff_cli_case.py <vcf_ploidy> <k> <sample_ploidy> <n>
runs the reference, `varigraph-mi genotype` and `varigraph-mi` with the HMM on the host (VGH_HMM_DEVICE=0) on the same files and prints
the first VCF lines that differ (how the round found the tally of an unselected called haplotype)."""
import os, sys, subprocess, gzip, shutil, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import importlib.util
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
spec = importlib.util.spec_from_file_location("tc", os.path.join(ROOT, "tests", "test_gpu_configs.py"))
tc = importlib.util.module_from_spec(spec); spec.loader.exec_module(tc)
from varigraph_amd import synth
vcf_ploidy, k, sample_ploidy, n = map(int, sys.argv[1:5])
work = tempfile.mkdtemp(prefix="dbg_")
ref = synth.make_reference(200_000)
n_s = 3 if vcf_ploidy >= 3 else 5
variants, gts = synth.make_cohort(ref, 300, n_samples=n_s, ploidy=vcf_ploidy, seed=3, indel_frac=0.1, sv_frac=0.01)
fa, vcf = os.path.join(work, "ref.fa"), os.path.join(work, "in.vcf")
synth.write_fasta(fa, "chr1", ref); synth.write_vcf(vcf, "chr1", len(ref), variants, gts, n_s, vcf_ploidy)
g = os.path.join(work, "g.bin")
subprocess.run([tc.REF, "construct", "-r", fa, "-v", vcf, "--save-graph", g, "-t", "8", "-k", str(k), "--vcf-ploidy", str(vcf_ploidy)], cwd=work, capture_output=True, env=tc.ENV)
sp = min(sample_ploidy, vcf_ploidy) if vcf_ploidy < sample_ploidy else sample_ploidy
haps = synth.sample_haplotypes(ref, variants, gts, 0, vcf_ploidy)[: max(1, sp)]
fq = tc._write_fastq(os.path.join(work, "s"), haps, 30_000, seed=5)
outs = {}
for name, exe, more in (("native", tc.CLI, ["--gpu", "0"]), ("cpu", tc.REF, []), ("native_host", tc.CLI, ["--gpu", "0"])):
    d = os.path.join(work, name); os.makedirs(d)
    open(os.path.join(d, "samples.cfg"), "w").write("s " + " ".join(fq) + "\n")
    env = dict(tc.ENV)
    if name == "native_host": env["VGH_HMM_DEVICE"] = "0"
    r = subprocess.run([exe, "genotype", "--load-graph", g, "-s", "samples.cfg", "-t", "6", "--sample-ploidy", str(sample_ploidy), "-n", str(n)] + more, cwd=d, capture_output=True, text=True, env=env)
    print(name, r.returncode, r.stderr[-300:] if r.returncode else "")
    outs[name] = gzip.open(os.path.join(d, "s.varigraph.vcf.gz"), "rb").read().split(b"\n")
a, b, c = outs["native"], outs["cpu"], outs["native_host"]
print(len(a), len(b), len(c), "host==cpu:", c == b)
nd = 0
for x, y in zip(a, b):
    if x != y:
        print("NATIVE:", x[:300].decode()); print("CPU   :", y[:300].decode()); nd += 1
        if nd >= 4: break
shutil.rmtree(work)
