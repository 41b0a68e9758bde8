#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REAL reference (oracle/_ref, built from /root/reference by
oracle/Makefile).  Runs only in the build container.  Everything written here is data: inputs and
the reference's outputs for them.  No reference source text is stored.

    python tests/golden/make_golden.py
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref")
HARNESS = os.path.join(REF, "ref_harness")
CLI_DET = os.path.join(REF, "varigraph_det")

from varigraph_amd import synth, vgmi  # noqa: E402


def run(args, stdin=None, cwd=None):
    r = subprocess.run(args, input=stdin, capture_output=True, cwd=cwd)
    if r.returncode != 0:
        sys.stderr.write(r.stderr.decode()[-2000:])
        raise SystemExit(f"{args[0]} failed")
    return r.stdout.decode()


def kat_strings(rng):
    acgt = "ACGT"
    def rnd(n, alphabet=acgt):
        return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=n))
    s = []
    s.append("ACGTACGTTGCAAGCTTAGCGATCGAT")
    s.append("acgtacgttgcaagcttagcgatcgatACGTTTGACCA")          # lower case
    s.append("ACGUACGUUGCAAGCUUAGCGAUCGAUacguuu")               # U / u
    s.append(rnd(150))
    s.append(rnd(60) + "N" + rnd(60))                          # one N
    s.append(rnd(30) + "NN" + rnd(26) + "n" + rnd(27) + "N" + rnd(28))
    s.append(rnd(26))                                          # shorter than 27
    s.append(rnd(27))
    s.append(rnd(28))
    s.append("A" * 60)
    s.append("ACGT" * 20)                                      # even-k palindromes everywhere
    s.append("AATT" + "GAATTC" * 6 + "N" + "GAATTC" * 4 + rnd(20))  # palindromes straddling an N
    s.append("GCGC" * 5 + "N" + "ATAT" * 5 + "ACGT" + rnd(40))
    s.append(rnd(200, "ACGTacgtNnU"))
    s.append("N" * 40)
    s.append(rnd(10) + "\x00\x01\x02\x03" * 8 + rnd(10))       # raw 0..3 bytes map to themselves
    s.append(rnd(40) + "-*." + rnd(40) + "R" + rnd(30))        # other IUPAC / punctuation
    for _ in range(12):
        s.append(rnd(int(rng.integers(1, 200)), "ACGTACGTACGTACGTACGTNacgtU"))
    return s


def make_kats():
    rng = np.random.default_rng(7)
    out = {"hash64": [], "sketch": [], "bloom_size": [], "murmur": []}
    # hash64
    for k in (1, 5, 6, 8, 21, 27, 28):
        vals = [0, 1, (1 << (2 * k)) - 1] + [int(x) & ((1 << (2 * k)) - 1) for x in rng.integers(0, 1 << 62, size=6)]
        res = run([HARNESS, "hash64", str(k)], "\n".join(f"{v:x}" for v in vals).encode()).split()
        out["hash64"].append({"k": k, "in": [f"{v:x}" for v in vals], "out": res})
    # emitter traces
    strings = kat_strings(rng)
    assert all("\n" not in s for s in strings)
    for k in (1, 5, 6, 8, 21, 27, 28):
        lines = run([HARNESS, "sketch", str(k)], ("\n".join(strings) + "\n").encode("latin1")).splitlines()
        assert len(lines) == len(strings), (len(lines), len(strings))
        traces = []
        for s, ln in zip(strings, lines):
            tok = ln.split()
            assert int(tok[0]) == len(tok) - 1
            traces.append({"seq_hex": s.encode("latin1").hex(), "keys": tok[1:]})
        out["sketch"].append({"k": k, "traces": traces})
    # bloom sizing (n -> m, n_hash)
    for n in (100, 974, 1000, 99974, 100000, 999974, 59999974, 2999999974):
        m, nh = run([HARNESS, "bloomsize", str(n), "0.01"]).split()
        out["bloom_size"].append({"n": n, "p": 0.01, "m": int(m), "n_hash": int(nh)})
    # murmur sum KATs (key, seed64) -> h1+h2
    pairs = [(0x2df5c044b3f1eb1b, s) for s in (0x1234567800000001, 0xdeadbeefcafef00d, 3, 4, 5, 6, 0xffffffffffffffff)]
    pairs += [(int(a), int(b)) for a, b in zip(rng.integers(0, 1 << 62, size=8), rng.integers(0, 1 << 62, size=8))]
    res = run([HARNESS, "murmur"], "\n".join(f"{a:x} {b:x}" for a, b in pairs).encode()).split()
    out["murmur"] = [{"key": f"{a:x}", "seed": f"{b:x}", "sum": r} for (a, b), r in zip(pairs, res)]
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("kats.json", {k: len(v) for k, v in out.items()})


def make_bloom():
    """Full filter for a 1 kb random reference (with an N run), k=27 and k=11, seeds fixed."""
    rng = np.random.default_rng(11)
    ref = synth.make_reference(1000, seed=5).tobytes().decode()
    ref2 = ref[:400] + "NNNNN" + ref[400:700].lower() + "N" + ref[700:]
    cases = []
    for name, k, seqs, seeds in (
        ("k27", 27, [ref], [1, 2, 3, 4, 5, 6, 7]),
        ("k11_multi", 11, [ref2, ref[100:300], "ACGT" * 30],
         [0x1234567800000001, 0xdeadbeefcafef00d, 3, 4, 5, 6, 0xffffffffffffffff]),
    ):
        n = sum(len(s) for s in seqs) - k + 1
        tmp = tempfile.mktemp()
        qkeys = []
        # query keys: every key of the first sequence + random keys
        tr = run([HARNESS, "sketch", str(k)], (seqs[0] + "\n").encode()).split()[1:]
        qkeys = tr[:50] + [f"{(int(x) << 8) | k:x}" for x in rng.integers(0, 1 << (2 * k - 1), size=30)]
        stdin = "\n".join(seqs) + "\nQ\n" + "\n".join(qkeys) + "\n"
        res = run([HARNESS, "bloom", str(k), str(n), "0.01", ",".join(f"{s:x}" for s in seeds), tmp], stdin.encode())
        dump = open(tmp, "rb").read()
        os.unlink(tmp)
        m = int.from_bytes(dump[:8], "little")
        nh = int.from_bytes(dump[8:12], "little")
        filt = np.frombuffer(dump, dtype=np.uint8, offset=12 + 8 * nh)
        assert filt.size == m
        q = [ln.split() for ln in res.splitlines()]
        np.save(os.path.join(HERE, f"bloom_{name}_filter.npy"), filt)
        cases.append({"name": name, "k": k, "n": n, "m": m, "n_hash": nh, "seeds": [f"{s:x}" for s in seeds],
                      "seqs": seqs, "query_keys": qkeys, "query_count": [int(a) for a, b in q],
                      "query_find": [int(b) for a, b in q], "filter_sum": int(filt.sum()),
                      "filter_nonzero": int((filt != 0).sum())})
        print("bloom", name, m, nh, int(filt.sum()))
    with open(os.path.join(HERE, "bloom.json"), "w") as f:
        json.dump(cases, f)


def mbf_fasta(path):
    """Two-record FASTA (multi-line second record with lower case, an N run and a duplicate name)."""
    ref = synth.make_reference(100_000, seed=synth.REF_SEED + 1)
    ref2 = synth.make_reference(30_000, seed=99)
    with open(path, "wb") as f:
        f.write(b">chr1 first\n" + ref.tobytes() + b"\n>chr2\n" + ref2[:15000].tobytes() + b"\n" +
                ref2[15000:].tobytes().lower() + b"NNNN\n>chr1 duplicate name: counted in the size, sequence ignored\n" +
                ref2[:500].tobytes() + b"\n")


def make_mbf_fixture():
    """Whole-reference counting Bloom filter from the reference's own build_fasta_index + make_mbf
    (det build: seeds are what _init_seeds draws when random_device returns 20241022)."""
    import hashlib
    fa = tempfile.mktemp(suffix=".fa")
    out = tempfile.mktemp()
    mbf_fasta(fa)
    cases = []
    for k in (27, 21):
        txt = run([os.path.join(REF, "ref_harness_det"), "mbf", fa, str(k), out])
        vals = dict(ln.split() for ln in txt.splitlines())
        d = open(out, "rb").read()
        m = int.from_bytes(d[:8], "little")
        nh = int.from_bytes(d[8:12], "little")
        seeds = [int.from_bytes(d[12 + 8 * i:20 + 8 * i], "little") for i in range(nh)]
        filt = np.frombuffer(d, dtype=np.uint8, offset=12 + 8 * nh)
        assert filt.size == m == int(vals["m"])
        cases.append({"k": k, "genome_size": int(vals["genome_size"]), "m": m, "n_hash": nh, "random_device_value": 20241022,
                      "seeds": [f"{x:x}" for x in seeds], "sha256": hashlib.sha256(filt.tobytes()).hexdigest(),
                      "sum": int(filt.sum()), "nonzero": int((filt != 0).sum()), "max": int(filt.max())})
        print("mbf", k, m, nh, cases[-1]["sum"])
    with open(os.path.join(HERE, "mbf.json"), "w") as f:
        json.dump(cases, f, indent=1)
    os.unlink(fa); os.unlink(out)


def genotype_modes(workdir, graph, sample_cfg_line, modes):
    out = {}
    for name, extra in modes.items():
        d = tempfile.mkdtemp(dir=workdir)
        with open(os.path.join(d, "samples.cfg"), "w") as f:
            f.write(sample_cfg_line + "\n")
        run([CLI_DET, "genotype", "--load-graph", graph, "-s", "samples.cfg", "-t", "4"] + extra, cwd=d)
        vcf = [p for p in os.listdir(d) if p.endswith(".varigraph.vcf.gz")]
        assert len(vcf) == 1
        out[name] = gzip.open(os.path.join(d, vcf[0]), "rb").read()
    return out


def make_cohort(name, ref_len, n_var, n_samples, ploidy, n_pairs, seed, indel_frac=0.0, sv_frac=0.0, k=27,
                store_reads=True, store_graph=True, modes=None, sample_ploidy=2):
    out_dir = os.path.join(HERE, name)
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    work = tempfile.mkdtemp()
    ref = synth.make_reference(ref_len, seed=synth.REF_SEED + seed)
    variants, gts = synth.make_cohort(ref, n_var, n_samples=n_samples, ploidy=ploidy, seed=seed,
                                      indel_frac=indel_frac, sv_frac=sv_frac)
    fa = os.path.join(work, "ref.fa")
    vcf = os.path.join(out_dir, "in.vcf")
    synth.write_fasta(fa, "chr1", ref)
    synth.write_vcf(vcf, "chr1", ref_len, variants, gts, n_samples, ploidy)
    graph = os.path.join(work, "graph.bin")
    run([CLI_DET, "construct", "-r", fa, "-v", vcf, "--save-graph", graph, "-k", str(k), "--vcf-ploidy", str(ploidy),
         "-t", "4"])
    haps = synth.sample_haplotypes(ref, variants, gts, 0, ploidy)
    n_reads = 2 * n_pairs
    block = vgmi.synth_reads_host(1000 + seed, 0, n_reads, 150, haps)
    fq = synth.write_fastq_pair(os.path.join(work, "reads"), block, n_reads, 150, gz=True)
    dump = os.path.join(work, "counts.bin")
    txt = run([HARNESS, "sample", graph, "4", str(sample_ploidy), "0", dump] + fq)
    meta = {"name": name, "ref_len": ref_len, "ref_seed": synth.REF_SEED + seed, "cohort_seed": seed,
            "n_var": n_var, "n_samples": n_samples, "ploidy": ploidy, "indel_frac": indel_frac, "sv_frac": sv_frac,
            "k": k, "n_pairs": n_pairs, "read_seed": 1000 + seed, "read_len": 150, "sample_ploidy": sample_ploidy}
    for ln in txt.splitlines():
        key, _, val = ln.partition(" ")
        if key == "hist":
            meta["hist"] = [int(x) for x in val.split()]
        elif key.endswith("_s"):
            continue
        else:
            meta[key] = val
    # threads must not matter: re-run with -t 1
    dump1 = os.path.join(work, "counts1.bin")
    run([HARNESS, "count", graph, "1", dump1] + fq)
    assert open(dump, "rb").read() == open(dump1, "rb").read(), "reference counts depend on thread count?!"
    # --use-depth variant of the coverage statistics
    dump2 = os.path.join(work, "counts2.bin")
    txt2 = run([HARNESS, "sample", graph, "4", str(sample_ploidy), "1", dump2] + fq)
    for ln in txt2.splitlines():
        key, _, val = ln.partition(" ")
        if key in ("hom_cov", "hap_kmer_cov_bits"):
            meta["use_depth_" + key] = val
    with gzip.open(os.path.join(out_dir, "counts.bin.gz"), "wb", compresslevel=9) as f:
        f.write(open(dump, "rb").read())
    with gzip.open(os.path.join(out_dir, "nodes.bin.gz"), "wb", compresslevel=9) as f:
        f.write(open(dump + ".nodes", "rb").read())
    if store_graph:
        with gzip.open(os.path.join(out_dir, "graph.bin.gz"), "wb", compresslevel=9) as f:
            f.write(open(graph, "rb").read())
    if store_reads:
        for p in fq:
            shutil.copy(p, os.path.join(out_dir, os.path.basename(p)))
    if modes:
        vcfs = genotype_modes(work, graph, "sample0 " + " ".join(fq), modes)
        for mname, data in vcfs.items():
            with open(os.path.join(out_dir, f"expected_{mname}.vcf"), "wb") as f:
                f.write(data)
    import hashlib
    meta["block_md5"] = hashlib.md5(block.tobytes()).hexdigest()
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    shutil.rmtree(work, ignore_errors=True)
    print(name, {k: meta[k] for k in ("n_keys", "read_base", "hom_cov", "max_cov")})


def main():
    for p in (HARNESS, CLI_DET):
        if not os.path.exists(p):
            raise SystemExit(f"{p} missing: run `make -C oracle ref` (needs /root/reference)")
    make_kats()
    make_bloom()
    make_mbf_fixture()
    modes = {"het": [], "hom": ["-g", "hom"], "use_depth": ["--use-depth"], "n5": ["-n", "5"]}
    # G4: tiny SNP cohort, 3 diploid VCF samples (7 haplotypes)
    make_cohort("cohort_snp", 100_000, 100, 3, 2, 3000, seed=1, modes=modes)
    # G4': indels + long insertions (exercises the >128 k-mer sort of graph2node)
    make_cohort("cohort_sv", 100_000, 80, 3, 2, 3000, seed=2, indel_frac=0.3, sv_frac=0.15, modes=modes)
    # even k
    make_cohort("cohort_k22", 50_000, 50, 3, 2, 1500, seed=3, k=22, modes={"het": []})
    # tetraploid
    make_cohort("cohort_tetra", 60_000, 60, 3, 4, 4000, seed=4, sample_ploidy=4,
                modes={"p4_use_depth": ["--sample-ploidy", "4", "--use-depth"]})
    # C1/C2 graph of BASELINE.json: 1 Mb, 1 k SNPs, 7 diploid samples (15 haplotypes); reads are regenerated
    make_cohort("c1", 1_000_000, 1000, 7, 2, 20000, seed=5, store_reads=False, modes=None)


if __name__ == "__main__":
    main()
