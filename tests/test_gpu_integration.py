"""End-to-end drop-in check (-m gpu): the UNMODIFIED reference compiled together with this repo's
adapter (integration/varigraph_hip.hpp -> oracle/_ref/varigraph_hip) runs `genotype` with the read
counting on the MI355X; the VCF it writes must be byte-identical (after gunzip) to the VCF the
all-CPU reference (oracle/_ref/varigraph_det, same deterministic flavour) writes ON THE SAME HOST
for the same graph, reads and options.

Same host matters: the reference's HMM is x87 `long double` and its GQ goes through log10l, whose
x87 transcendental instructions are not bit-identical across CPU vendors -- the all-CPU reference
itself prints GQ 99.0 vs 192.7 at a few sites on the GPU box's EPYC vs the build container's Xeon
(genotypes and every other field agree).  The committed expected_*.vcf (made in the build
container) are therefore compared field-wise without GQ."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _run(cmd, **kw):
    """subprocess.run with two additions.  (1) A timeout becomes a failure naming the binary and showing what it had
    printed.  (2) Binaries that contain the reference's code (oracle/_ref: the all-CPU reference and the adapter build)
    are retried: the reference's thread pool notifies its workers without holding the mutex they wait under
    (include/ThreadPool.hpp: submit() / shutdown()), so a run can, rarely, sleep forever on a lost wake-up -- seen twice
    in ~30 runs of this file on the 256-thread measurement host.  The native CLI does not use that pool and gets no
    retry."""
    has_reference_code = os.sep + os.path.join("oracle", "_ref") + os.sep in str(cmd[0])
    attempts = 6 if has_reference_code else 1
    if has_reference_code and "timeout" in kw:
        kw = dict(kw, timeout=min(kw["timeout"], 25))   # the fixtures take a second or two
    for attempt in range(attempts):
        try:
            return subprocess.run(cmd, **kw)
        except subprocess.TimeoutExpired as e:
            err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
            if attempt + 1 < attempts:
                print(f"[retry] {cmd[0]} did not finish in {e.timeout} s (attempt {attempt + 1}); stderr tail: {err[-300:]}")
                continue
            raise AssertionError(f"TIMEOUT after {e.timeout} s: {' '.join(map(str, cmd[:3]))} ...\nstderr so far:\n{err[-3000:]}") from None


pytestmark = pytest.mark.gpu


def _missing(why):
    """Every test here is a `-m gpu` test and every binary it needs is built by __graft_entry__.build() and travels to the
    GPU box with the tree (oracle/_ref is git-ignored, not gpurun-ignored): an absent binary is a broken hand-over, not a
    reason to report green without the VCF comparison."""
    pytest.fail("required binary absent on the GPU box: " + why, pytrace=False)

BIN = os.path.join(ROOT, "oracle", "_ref", "varigraph_hip")
REF = os.path.join(ROOT, "oracle", "_ref", "varigraph_det")


def _strip_gq(vcf):
    out = []
    for ln in vcf.split(b"\n"):
        if ln and not ln.startswith(b"#"):
            cols = ln.split(b"\t")
            f = cols[9].split(b":")
            f[1] = b"GQ"
            cols[9] = b":".join(f)
            ln = b"\t".join(cols)
        out.append(ln)
    return b"\n".join(out)

CASES = [
    ("cohort_snp", "het", []), ("cohort_snp", "hom", ["-g", "hom"]), ("cohort_snp", "use_depth", ["--use-depth"]),
    ("cohort_snp", "n5", ["-n", "5"]),
    ("cohort_sv", "het", []), ("cohort_sv", "hom", ["-g", "hom"]), ("cohort_sv", "use_depth", ["--use-depth"]),
    ("cohort_sv", "n5", ["-n", "5"]),
    ("cohort_k22", "het", []),
    ("cohort_tetra", "p4_use_depth", ["--sample-ploidy", "4", "--use-depth"]),
]


@pytest.mark.parametrize("cohort,mode,extra", CASES, ids=[f"{c}-{m}" for c, m, _ in CASES])
def test_genotype_vcf_identical_to_reference(cohort, mode, extra, tmp_path):
    if not (os.path.exists(BIN) and os.path.exists(REF)):
        _missing("integration binary not built (needs /root/reference at build time: make -C oracle ref)")
    d = os.path.join(GOLDEN, cohort)
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    (tmp_path / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
    r = _run([BIN, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4",
                        "--gpu", "0", "--buffer", "8"] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = gzip.open(tmp_path / "sample0.varigraph.vcf.gz", "rb").read()
    assert got.count(b"\n") > 20
    # (1) byte-identical to the all-CPU reference on this host
    cpu = tmp_path / "cpu"
    cpu.mkdir()
    (cpu / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
    r2 = _run([REF, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4"] + extra,
                        cwd=cpu, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    want_here = gzip.open(cpu / "sample0.varigraph.vcf.gz", "rb").read()
    assert got == want_here
    # (2) equal to the committed fixture from the build container in everything but GQ
    want = open(os.path.join(d, f"expected_{mode}.vcf"), "rb").read()
    assert _strip_gq(got) == _strip_gq(want)


def test_integration_binary_fails_loudly_on_bad_input(tmp_path):
    if not os.path.exists(BIN):
        _missing("integration binary not built")
    r = _run([BIN, "genotype", "--load-graph", str(tmp_path / "nope.bin"), "-s", "nope.cfg"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


# ------------------------------------------------------------------------------------------------------------------
# The native CLI (varigraph_amd/bin/varigraph-mi: own graph.bin reader, device counting, own host HMM and VCF writer;
# no reference code in the binary).  VGH_RANDOM_DEVICE_VALUE pins the haplotype sampler's seed source the way the
# deterministic reference build pins std::random_device.
CLI = os.path.join(ROOT, "varigraph_amd", "bin", "varigraph-mi")
NATIVE_CASES = CASES + [
    ("cohort_sv", "fre", ["-m", "fre"]), ("cohort_sv", "sv", ["--sv"]), ("cohort_sv", "minsupport", ["--min-support", "30"]),
    ("cohort_sv", "gran", ["--granularity", "0.02"]), ("cohort_snp", "fre_n5", ["-m", "fre", "-n", "5"]),
]


@pytest.mark.parametrize("cohort,mode,extra", NATIVE_CASES, ids=[f"{c}-{m}" for c, m, _ in NATIVE_CASES])
def test_native_cli_vcf_identical_to_reference(cohort, mode, extra, tmp_path):
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built (python -m varigraph_amd.build)")
    d = os.path.join(GOLDEN, cohort)
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    (tmp_path / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4", "--gpu", "0",
                        "--buffer", "8"] + extra, cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = gzip.open(tmp_path / "sample0.varigraph.vcf.gz", "rb").read()
    # (1) equal to the committed fixture from the build container in everything but GQ
    want = open(os.path.join(d, f"expected_{mode}.vcf"), "rb").read()
    assert _strip_gq(got) == _strip_gq(want)
    # (2) byte-identical to the all-CPU reference run on this host
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det (the unmodified reference, built by `make -C oracle ref`)")
    cpu = tmp_path / "cpu"
    cpu.mkdir()
    (cpu / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
    r2 = _run([REF, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4"] + extra,
                        cwd=cpu, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert got == gzip.open(cpu / "sample0.varigraph.vcf.gz", "rb").read()


def test_native_cli_two_samples_and_errors(tmp_path):
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    d = os.path.join(GOLDEN, "cohort_snp")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    (tmp_path / "samples.cfg").write_text("".join(f"sample{s} " + " ".join(fq) + "\n" for s in (0, 1)))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4", "-n", "5"], cwd=tmp_path,
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for s in (0, 1):
        got = gzip.open(tmp_path / f"sample{s}.varigraph.vcf.gz", "rb").read()
        want = open(os.path.join(d, f"expected_two_n5_s{s}.vcf"), "rb").read()
        assert _strip_gq(got) == _strip_gq(want)
    # loud failures: missing graph, missing read file, bad option value
    assert _run([CLI, "genotype", "--load-graph", str(tmp_path / "nope.bin"), "-s", "samples.cfg"], cwd=tmp_path,
                          capture_output=True, timeout=300).returncode != 0
    (tmp_path / "bad.cfg").write_text("s0 /nonexistent_1.fq.gz\n")
    assert _run([CLI, "genotype", "--load-graph", str(graph), "-s", "bad.cfg"], cwd=tmp_path,
                          capture_output=True, timeout=300).returncode != 0
    assert _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-g", "maybe"], cwd=tmp_path,
                          capture_output=True, timeout=300).returncode != 0


def test_native_cli_samples_of_one_and_of_three_files(tmp_path):
    """A sample line names any number of files (src/varigraph.cpp:104-146): one file (single-end), three files (a file twice: the
    coverage doubles), the usual pair -- in ONE run, each VCF byte for byte against the reference run on that sample alone."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    d = os.path.join(GOLDEN, "cohort_sv")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    one, two = (os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2))
    plain = tmp_path / "reads_2.fq"
    plain.write_bytes(gzip.open(two, "rb").read())
    samples = {"single": [one], "triple": [one, str(plain), one], "pair": [one, two], "other": [str(plain)]}
    (tmp_path / "samples.cfg").write_text("".join(f"{n} " + " ".join(f) + "\n" for n, f in samples.items()))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "6", "--gpu", "0", "--use-depth"], cwd=tmp_path,
             capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for n, f in samples.items():
        cpu = tmp_path / ("cpu_" + n)
        cpu.mkdir()
        (cpu / "samples.cfg").write_text(f"{n} " + " ".join(f) + "\n")
        r2 = _run([REF, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4", "--use-depth"], cwd=cpu, capture_output=True,
                  text=True, timeout=300)
        assert r2.returncode == 0, r2.stderr[-2000:]
        assert gzip.open(tmp_path / f"{n}.varigraph.vcf.gz", "rb").read() == gzip.open(cpu / f"{n}.varigraph.vcf.gz", "rb").read(), n


@pytest.mark.parametrize("ploidy,n", [(3, 7), (3, 15), (5, 4), (6, 4), (8, 3)])
def test_native_cli_other_sample_ploidies(ploidy, n, tmp_path):
    """`--sample-ploidy` 2..8 (main.cpp:355-357): triploid runs on the device (a stride of four powers per step), five and more
    copies stay on the host's x87 recursion -- either way byte for byte the reference's VCF, on the tetraploid cohort's graph."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    d = os.path.join(GOLDEN, "cohort_tetra")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    extra = ["--sample-ploidy", str(ploidy), "-n", str(n), "--use-depth"]
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    outs = {}
    for name, exe, more in (("native", CLI, ["--gpu", "0"]), ("cpu", REF, [])):
        w = tmp_path / name
        w.mkdir()
        (w / "samples.cfg").write_text("s " + " ".join(fq) + "\n")
        r = _run([exe, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4"] + extra + more, cwd=w, capture_output=True,
                 text=True, env=env, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        outs[name] = gzip.open(w / "s.varigraph.vcf.gz", "rb").read()
    assert outs["native"] == outs["cpu"] and outs["cpu"].count(b"\n") > 20


def test_native_cli_ragged_reads_with_n_and_lower_case(tmp_path):
    """Reads as sequencers deliver them -- trimmed to any length (some shorter than k), with N calls, soft-clipped lower case -- as
    plain, gzip and block-gzip FASTQ and as wrapped FASTA, one sample each: every VCF byte for byte the reference's on the same text."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, "cohort_snp")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    rng = np.random.default_rng(12)
    files = []
    for mate in (1, 2):
        rec = gzip.open(os.path.join(d, f"reads_{mate}.fq.gz"), "rb").read().split(b"\n")
        out = []
        for i in range(0, len(rec) - 3, 4):
            name, seq, qual = rec[i], bytearray(rec[i + 1]), rec[i + 3]
            j = i // 4
            if j % 3 == 0:
                seq = seq[: int(rng.integers(5, len(seq) + 1))]
            if j % 5 == 0 and len(seq) > 8:
                for p_ in rng.integers(0, len(seq), size=2):
                    seq[int(p_)] = ord("N")
            if j % 7 == 0:
                cut = int(rng.integers(0, len(seq)))
                seq[cut:] = bytes(seq[cut:]).lower()
            out += [name, bytes(seq), b"+", qual[: len(seq)]]
        text = b"\n".join(out) + b"\n"
        plain = tmp_path / f"r_{mate}.fq"
        plain.write_bytes(text)
        with gzip.open(tmp_path / f"r_{mate}.fq.gz", "wb", compresslevel=3) as f:
            f.write(text)
        synth.bgzf_compress_file(str(plain), str(tmp_path / f"r_{mate}.bgz.gz"), block=30000)
        fasta = tmp_path / f"r_{mate}.fa"          # the same reads as FASTA records, sequences wrapped at 60 (kseq reads both)
        fasta.write_bytes(b"".join(b">" + out[q][1:] + b"\n" + b"\n".join(out[q + 1][w:w + 60] for w in range(0, max(len(out[q + 1]), 1), 60)) + b"\n"
                                   for q in range(0, len(out), 4)))
        files.append({"plain": str(plain), "gzip": str(tmp_path / f"r_{mate}.fq.gz"), "bgzf": str(tmp_path / f"r_{mate}.bgz.gz"), "fasta": str(fasta)})
    kinds = ("plain", "gzip", "bgzf", "fasta")
    (tmp_path / "samples.cfg").write_text("".join(f"{k} {files[0][k]} {files[1][k]}\n" for k in kinds))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "6", "--gpu", "0", "--buffer", "8"], cwd=tmp_path,
             capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cpu = tmp_path / "cpu"
    cpu.mkdir()
    (cpu / "samples.cfg").write_text(f"plain {files[0]['gzip']} {files[1]['gzip']}\n")
    r2 = _run([REF, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4"], cwd=cpu, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    want = gzip.open(cpu / "plain.varigraph.vcf.gz", "rb").read()
    assert want.count(b"\n") > 30
    for k in kinds:
        got = gzip.open(tmp_path / f"{k}.varigraph.vcf.gz", "rb").read()
        assert got.replace(b"\t" + k.encode() + b"\n", b"\tplain\n") == want, k


def test_native_cli_sample_without_depth_fails_like_the_reference(tmp_path):
    """A sample of a dozen reads has no k-mer depth to speak of: the reference gives up ("Failed to retrieve depth information",
    src/varigraph.cpp:308-362) with a non-zero status, and so does `varigraph-mi` -- with or without `--use-depth`."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    d = os.path.join(GOLDEN, "cohort_snp")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    few = tmp_path / "few.fq"
    few.write_bytes(b"\n".join(gzip.open(os.path.join(d, "reads_1.fq.gz"), "rb").read().split(b"\n")[:48]) + b"\n")
    (tmp_path / "samples.cfg").write_text(f"s {few}\n")
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    for extra in ([], ["--use-depth"]):
        r1 = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "2", "--gpu", "0"] + extra, cwd=tmp_path,
                  capture_output=True, text=True, env=env, timeout=300)
        r2 = _run([REF, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "2"] + extra, cwd=tmp_path, capture_output=True,
                  text=True, timeout=300)
        assert (r1.returncode == 0) == (r2.returncode == 0), (extra, r1.returncode, r2.returncode, r1.stderr[-600:], r2.stderr[-600:])
        if r2.returncode == 0:      # (should the reference ever produce a VCF from this, so must we, and the same)
            assert False, "the reference genotyped a dozen reads: compare the VCFs here"


@pytest.mark.parametrize("extra", [["--sv"], ["--sv", "--use-depth", "-n", "7"], ["--sv", "--granularity", "0.003"]], ids=["sv", "sv-use-depth-n7", "sv-small-windows"])
def test_native_cli_sv_only_over_a_graph_without_long_alleles(extra, tmp_path):
    """`--sv` over a cohort of SNPs: no node is the HMM's business, the reference writes the header and nothing else -- and so must
    `varigraph-mi` (a part of the windows without rows used to end the run with a device error; found by tools/fuzz_cli_parity.py)."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    d = os.path.join(GOLDEN, "cohort_snp")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    outs = {}
    for name, exe, more in (("native", CLI, ["--gpu", "0"]), ("cpu", REF, [])):
        w = tmp_path / name
        w.mkdir()
        (w / "samples.cfg").write_text("a " + " ".join(fq) + "\nb " + " ".join(fq) + "\n")
        r = _run([exe, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4"] + extra + more, cwd=w, capture_output=True, text=True,
                 env=env, timeout=300)
        assert r.returncode == 0, (name, r.stderr[-1500:])
        outs[name] = [gzip.open(w / f"{n}.varigraph.vcf.gz", "rb").read() for n in ("a", "b")]
    assert outs["native"] == outs["cpu"]


def test_native_cli_several_devices_keep_sample_order(tmp_path):
    """--gpus a,b: samples are counted on several device contexts in parallel (here the same GPU twice) while the
    HMM consumes them strictly in `-s` order -- its per-node state carries over from sample to sample, so the result
    must be the one of the sequential run."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    d = os.path.join(GOLDEN, "cohort_snp")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    (tmp_path / "samples.cfg").write_text("".join(f"sample{s} " + " ".join(fq) + "\n" for s in range(4)))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4", "-n", "5", "--gpus", "0,0"],
                       cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "table: 1 build on device 0, 1 device-to-device image copies" in r.stderr   # built once, copied, never re-uploaded
    body = lambda v: [ln for ln in _strip_gq(v).split(b"\n") if ln and not ln.startswith(b"#")]
    got = [gzip.open(tmp_path / f"sample{s}.varigraph.vcf.gz", "rb").read() for s in range(4)]
    for s in (0, 1):
        want = open(os.path.join(d, f"expected_two_n5_s{s}.vcf"), "rb").read()
        assert _strip_gq(got[s]) == _strip_gq(want)
    assert body(got[2]) == body(got[1]) and body(got[3]) == body(got[1])


def test_native_cli_independent_samples_run_side_by_side(tmp_path):
    """-n >= #haplotypes: nothing can be pruned from a node's k-mer list, the samples are independent, and --gpus a,b,c
    runs one genotyping consumer per device context.  Every sample must come out as the single-sample run does."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    d = os.path.join(GOLDEN, "cohort_snp")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    (tmp_path / "samples.cfg").write_text("".join(f"sample{s} " + " ".join(fq) + "\n" for s in range(5)))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "3", "--gpus", "0,0,0"],
                       cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "table: 1 build on device 0, 2 device-to-device image copies" in r.stderr
    want = open(os.path.join(d, "expected_het.vcf"), "rb").read()
    for s in range(5):
        got = gzip.open(tmp_path / f"sample{s}.varigraph.vcf.gz", "rb").read()
        assert _strip_gq(got).replace(b"sample%d" % s, b"sample0") == _strip_gq(want).replace(b"sample0", b"sample0"), s


# ------------------------------------------------------------------------------------------------------------------
# `varigraph-mi construct`: the committed graph.bin.gz of every cohort was written by the unmodified reference
# (deterministic build, std::random_device = 20241022) from the committed in.vcf and the seeded synthetic reference.
# The native construct (Bloom filter built and queried on the device) must reproduce the file byte for byte.
@pytest.mark.parametrize("cohort", ["cohort_snp", "cohort_sv", "cohort_k22", "cohort_tetra"])
def test_native_construct_reproduces_reference_graph(cohort, tmp_path):
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, cohort)
    meta = json.load(open(os.path.join(d, "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    fa = tmp_path / "ref.fa"
    synth.write_fasta(str(fa), "chr1", ref)
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "construct", "-r", str(fa), "-v", os.path.join(d, "in.vcf"), "--save-graph", "graph.bin", "-k",
                        str(meta["k"]), "--vcf-ploidy", str(meta["ploidy"]), "--gpu", "0"], cwd=tmp_path, capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = (tmp_path / "graph.bin").read_bytes()
    want = gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read()
    assert len(got) == len(want)
    assert got == want


@pytest.mark.parametrize("opts", [["--fast"], ["--use-unique-kmers"], ["--fast", "--use-unique-kmers"], ["-k", "21", "--fast"]],
                         ids=["fast", "unique", "fast-unique", "k21-fast"])
@pytest.mark.parametrize("cohort", ["cohort_snp", "cohort_sv"])
def test_native_construct_options_reproduce_reference_graph(cohort, opts, tmp_path):
    """`construct --fast` / `--use-unique-kmers` (main.cpp:88-109, construct_index.cpp): graph.bin byte for byte the reference's,
    built here by the reference itself with the same options."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, cohort)
    meta = json.load(open(os.path.join(d, "meta.json")))
    fa = tmp_path / "ref.fa"
    synth.write_fasta(str(fa), "chr1", synth.make_reference(meta["ref_len"], seed=meta["ref_seed"]))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    base = ["construct", "-r", str(fa), "-v", os.path.join(d, "in.vcf"), "--vcf-ploidy", str(meta["ploidy"])] + opts
    r = _run([CLI] + base + ["--save-graph", "native.bin", "--gpu", "0"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run([REF] + base + ["--save-graph", "cpu.bin"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got, want = (tmp_path / "native.bin").read_bytes(), (tmp_path / "cpu.bin").read_bytes()
    assert len(got) == len(want) and got == want


def test_native_construct_soft_masked_reference_with_n_runs(tmp_path):
    """A reference as assemblies come: soft-masked (lower-case) stretches and runs of N, some of them across variant sites.  Whatever
    the reference makes of it -- a graph, or an error -- `varigraph-mi construct` makes the same."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, "cohort_sv")
    meta = json.load(open(os.path.join(d, "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"]).copy()
    n = ref.size
    pos = [int(ln.split("\t")[1]) - 1 for ln in open(os.path.join(d, "in.vcf")) if not ln.startswith("#")]
    ref[n // 10: n // 5] |= 0x20                                   # soft-masked stretch, variant sites included
    free = [p for p in range(1000, n - 1000, 997) if all(abs(p - q) > 400 for q in pos)][:6]
    for p in free:
        ref[p: p + 73] = ord("N")                                  # N runs away from the sites
    near = pos[len(pos) // 2]
    ref[near + 40: near + 45] = ord("n")                           # ... and one inside the k-mer reach of a site
    fa = tmp_path / "ref.fa"
    synth.write_fasta(str(fa), "chr1", ref)
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    base = ["construct", "-r", str(fa), "-v", os.path.join(d, "in.vcf"), "--vcf-ploidy", str(meta["ploidy"])]
    r1 = _run([CLI] + base + ["--save-graph", "native.bin", "--gpu", "0"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    r2 = _run([REF] + base + ["--save-graph", "cpu.bin"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert (r1.returncode == 0) == (r2.returncode == 0), (r1.returncode, r2.returncode, r1.stderr[-800:], r2.stderr[-800:])
    assert r2.returncode == 0          # (the reference takes this input: seven of the sites lie in the soft-masked stretch)
    if r2.returncode == 0:
        assert (tmp_path / "native.bin").read_bytes() == (tmp_path / "cpu.bin").read_bytes()
        fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
        outs = {}
        for name, exe, more in (("native", CLI, ["--gpu", "0"]), ("cpu", REF, [])):
            w = tmp_path / name
            w.mkdir()
            (w / "samples.cfg").write_text("s " + " ".join(fq) + "\n")
            r = _run([exe, "genotype", "--load-graph", str(tmp_path / "cpu.bin"), "-s", "samples.cfg", "-t", "4"] + more, cwd=w, capture_output=True,
                     text=True, env=env, timeout=600)
            assert r.returncode == 0, (name, r.stderr[-2000:])
            outs[name] = gzip.open(w / "s.varigraph.vcf.gz", "rb").read()
        assert outs["native"] == outs["cpu"]


def _odd_vcf(path, ref, flavour):
    """A cohort VCF with what real call sets hold and the fixtures do not; `flavour` picks one kind of line (or "all")."""
    acgt = b"ACGT"
    n = ref.size
    lines = []

    def other(base, k=1):
        return [bytes([c]) for c in acgt if c != base][:k]

    sites = list(range(3000, n - 3000, 1900))
    for i, p in enumerate(sites):
        b = int(ref[p])
        kind = ["multi_snp", "multi_indel", "missing", "unphased", "extra_format", "overlap", "plain"][i % 7]
        if flavour != "all" and kind not in (flavour, "plain"):
            kind = "plain"
        r_ = bytes([b]).decode()
        if kind == "multi_snp":
            a1, a2 = (x.decode() for x in other(b, 2))
            lines.append((p, r_, f"{a1},{a2}", "GT", ["0|1", "2|1", "0|2", "2|2"]))
        elif kind == "multi_indel":
            lines.append((p, r_, f"{r_}T,{r_}TTG", "GT", ["1|2", "0|1", "2|0", "0|0"]))
        elif kind == "missing":
            lines.append((p, r_, other(b)[0].decode(), "GT", [".|.", "0|1", ".", "1|1"]))
        elif kind == "unphased":
            lines.append((p, r_, other(b)[0].decode(), "GT", ["0/1", "1/1", "0|1", "1/0"]))
        elif kind == "extra_format":
            lines.append((p, r_, other(b)[0].decode(), "GT:DP:GQ", ["0|1:12:99", "1|1:7:40", "0|0:9:50", "1|0:3:10"]))
        elif kind == "overlap":
            lines.append((p, bytes(ref[p:p + 9]).decode(), r_, "GT", ["0|1", "0|0", "1|0", "0|1"]))       # a deletion ...
            lines.append((p + 4, bytes([int(ref[p + 4])]).decode(), other(int(ref[p + 4]))[0].decode(), "GT", ["1|0", "0|1", "0|0", "1|1"]))   # ... across a SNP
        else:
            lines.append((p, r_, other(b)[0].decode(), "GT", ["0|1", "1|0", "1|1", "0|0"]))
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=chr1,length=%d>\n" % n)
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0\tS1\tS2\tS3\n")
        for i, (p, r_, a, fmt, gts) in enumerate(lines):
            f.write(f"chr1\t{p + 1}\tx{i}\t{r_}\t{a}\t.\tPASS\t.\t{fmt}\t" + "\t".join(gts) + "\n")
    return len(lines)


@pytest.mark.parametrize("flavour", ["multi_snp", "multi_indel", "missing", "unphased", "extra_format", "overlap", "all"])
def test_native_construct_call_set_oddities_like_the_reference(flavour, tmp_path):
    """Multi-allelic sites, missing and unphased genotypes, FORMAT fields behind GT, a deletion across a SNP: whatever the reference
    makes of such a cohort VCF -- a graph or an error -- `varigraph-mi` makes the same, and from the same graph the same calls."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, "cohort_snp")
    meta = json.load(open(os.path.join(d, "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    fa, vcf = tmp_path / "ref.fa", tmp_path / "odd.vcf"
    synth.write_fasta(str(fa), "chr1", ref)
    assert _odd_vcf(str(vcf), ref, flavour) > 20
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    base = ["construct", "-r", str(fa), "-v", str(vcf)]
    r1 = _run([CLI] + base + ["--save-graph", "native.bin", "--gpu", "0"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    r2 = _run([REF] + base + ["--save-graph", "cpu.bin"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert (r1.returncode == 0) == (r2.returncode == 0), (flavour, r1.returncode, r2.returncode, r1.stderr[-800:], r2.stderr[-800:])
    if r2.returncode != 0:
        return
    assert (tmp_path / "native.bin").read_bytes() == (tmp_path / "cpu.bin").read_bytes(), flavour
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    outs = {}
    for name, exe, more in (("native", CLI, ["--gpu", "0"]), ("cpu", REF, [])):
        w = tmp_path / name
        w.mkdir()
        (w / "samples.cfg").write_text("s " + " ".join(fq) + "\n")
        r = _run([exe, "genotype", "--load-graph", str(tmp_path / "cpu.bin"), "-s", "samples.cfg", "-t", "4"] + more, cwd=w, capture_output=True,
                 text=True, env=env, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        outs[name] = gzip.open(w / "s.varigraph.vcf.gz", "rb").read()
    assert outs["native"] == outs["cpu"], flavour


@pytest.mark.parametrize("what", ["fasta_crlf", "vcf_crlf", "fasta_lower_case_names_and_description", "vcf_gz_fasta_gz", "fasta_no_final_newline"])
def test_native_construct_file_format_corners_like_the_reference(what, tmp_path):
    """Files as other tools leave them: CRLF line ends in the FASTA or the VCF, a description behind the sequence name, gzip (not block
    gzip) inputs, no newline at the end of the FASTA.  The same graph.bin as the reference's construct, or the same refusal."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    if not os.path.exists(REF):
        _missing("oracle/_ref/varigraph_det")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, "cohort_sv")
    meta = json.load(open(os.path.join(d, "meta.json")))
    ref = synth.make_reference(meta["ref_len"], seed=meta["ref_seed"])
    fa, vcf = tmp_path / "ref.fa", tmp_path / "in.vcf"
    synth.write_fasta(str(fa), "chr1", ref)
    vcf.write_bytes(open(os.path.join(d, "in.vcf"), "rb").read())
    fa_arg, vcf_arg = str(fa), str(vcf)
    if what == "fasta_crlf":
        fa.write_bytes(fa.read_bytes().replace(b"\n", b"\r\n"))
    elif what == "vcf_crlf":
        vcf.write_bytes(vcf.read_bytes().replace(b"\n", b"\r\n"))
    elif what == "fasta_lower_case_names_and_description":
        fa.write_bytes(fa.read_bytes().replace(b">chr1\n", b">chr1 assembled by somebody, length=%d\n" % ref.size))
    elif what == "vcf_gz_fasta_gz":
        for p_ in (fa, vcf):
            with gzip.open(str(p_) + ".gz", "wb") as f:
                f.write(p_.read_bytes())
        fa_arg, vcf_arg = str(fa) + ".gz", str(vcf) + ".gz"
    elif what == "fasta_no_final_newline":
        fa.write_bytes(fa.read_bytes().rstrip(b"\n"))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    base = ["construct", "-r", fa_arg, "-v", vcf_arg, "--vcf-ploidy", str(meta["ploidy"])]
    r1 = _run([CLI] + base + ["--save-graph", "native.bin", "--gpu", "0"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    r2 = _run([REF] + base + ["--save-graph", "cpu.bin"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
    assert (r1.returncode == 0) == (r2.returncode == 0), (what, r1.returncode, r2.returncode, r1.stderr[-800:], r2.stderr[-800:])
    if r2.returncode == 0:
        assert (tmp_path / "native.bin").read_bytes() == (tmp_path / "cpu.bin").read_bytes(), what
    if r2.returncode == 0 and what == "vcf_crlf":
        # the reference reads no variant out of a CRLF VCF and writes a graph of 53 bytes: genotyping from it must end the same way
        fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
        res = {}
        for name, exe, more in (("native", CLI, ["--gpu", "0"]), ("cpu", REF, [])):
            w = tmp_path / name
            w.mkdir()
            (w / "samples.cfg").write_text("s " + " ".join(fq) + "\n")
            r = _run([exe, "genotype", "--load-graph", str(tmp_path / "cpu.bin"), "-s", "samples.cfg", "-t", "2"] + more, cwd=w, capture_output=True,
                     text=True, env=env, timeout=300)
            out = w / "s.varigraph.vcf.gz"
            res[name] = (r.returncode == 0, gzip.open(out, "rb").read() if r.returncode == 0 and out.exists() else None)
        assert res["native"] == res["cpu"], (res["native"][0], res["cpu"][0])


def test_native_construct_then_genotype_and_errors(tmp_path):
    """construct -> genotype with nothing but this repo's binaries, against the reference's VCF; loud failures."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, "cohort_sv")
    meta = json.load(open(os.path.join(d, "meta.json")))
    fa = tmp_path / "ref.fa"
    synth.write_fasta(str(fa), "chr1", synth.make_reference(meta["ref_len"], seed=meta["ref_seed"]))
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "construct", "-r", str(fa), "-v", os.path.join(d, "in.vcf"), "--save-graph", "g.bin"], cwd=tmp_path,
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    (tmp_path / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
    r = _run([CLI, "genotype", "--load-graph", "g.bin", "-s", "samples.cfg", "-t", "4"], cwd=tmp_path,
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = gzip.open(tmp_path / "sample0.varigraph.vcf.gz", "rb").read()
    assert _strip_gq(got) == _strip_gq(open(os.path.join(d, "expected_het.vcf"), "rb").read())
    for bad in (["-r", str(tmp_path / "nope.fa"), "-v", os.path.join(d, "in.vcf")], ["-r", str(fa), "-v", str(tmp_path / "nope.vcf")],
                ["-r", str(fa), "-v", os.path.join(d, "in.vcf"), "-k", "31"]):
        assert _run([CLI, "construct"] + bad, cwd=tmp_path, capture_output=True, timeout=300).returncode != 0


def test_native_cli_block_gzip_and_plain_inputs(tmp_path):
    """Ingest row: the same cohort with a bgzip'd VCF / reference for construct and block-gzip, plain and
    gzip FASTQ for genotype gives the committed graph and one and the same VCF (byte_source.hpp picks the decoder)."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    import json
    from varigraph_amd import synth
    d = os.path.join(GOLDEN, "cohort_snp")
    meta = json.load(open(os.path.join(d, "meta.json")))
    fa = tmp_path / "ref.fa"
    synth.write_fasta(str(fa), "chr1", synth.make_reference(meta["ref_len"], seed=meta["ref_seed"]))
    synth.bgzf_compress_file(str(fa), str(tmp_path / "ref.fa.gz"), block=4000)
    synth.bgzf_compress_file(os.path.join(d, "in.vcf"), str(tmp_path / "in.vcf.gz"), block=3000)
    env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022")
    r = _run([CLI, "construct", "-r", "ref.fa.gz", "-v", "in.vcf.gz", "--save-graph", "g.bin", "-t", "4"], cwd=tmp_path,
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "g.bin").read_bytes() == gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read()
    vcfs = {}
    for kind in ("gz", "plain", "bgzf"):
        fq = []
        for i in (1, 2):
            src = os.path.join(d, f"reads_{i}.fq.gz")
            if kind == "gz":
                fq.append(src)
                continue
            plain = tmp_path / f"reads_{i}.fq"
            plain.write_bytes(gzip.open(src, "rb").read())
            fq.append(str(plain) if kind == "plain" else synth.bgzf_compress_file(str(plain), str(tmp_path / f"reads_{i}.b.fq.gz")))
        (tmp_path / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
        r = _run([CLI, "genotype", "--load-graph", "g.bin", "-s", "samples.cfg", "-t", "6"] + (["-D"] if kind == "plain" else []),
                           cwd=tmp_path, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("[graph_index]" in r.stderr) == (kind == "plain")   # -D / --debug: single-threaded, phase times
        vcfs[kind] = gzip.open(tmp_path / "sample0.varigraph.vcf.gz", "rb").read()
    assert vcfs["plain"] == vcfs["gz"] and vcfs["bgzf"] == vcfs["gz"]
    want = open(os.path.join(d, "expected_het.vcf"), "rb").read() if os.path.exists(os.path.join(d, "expected_het.vcf")) else None
    if want is not None:
        assert _strip_gq(vcfs["gz"]) == _strip_gq(want)


def test_native_cli_hmm_on_the_device_in_parts_equals_the_host_hmm(tmp_path):
    """The same sample with the HMM's recursion and posterior on the device (the default: the windows go to the device in
    parts as they are prepared -- twenty windows of 5 kb here, four parts) and on the host (VGH_HMM_DEVICE=0): identical VCF bytes, and
    the log shows which one ran."""
    if not os.path.exists(CLI):
        _missing("varigraph-mi not built")
    d = os.path.join(GOLDEN, "cohort_sv")
    graph = tmp_path / "graph.bin"
    graph.write_bytes(gzip.open(os.path.join(d, "graph.bin.gz"), "rb").read())
    fq = [os.path.join(d, f"reads_{i}.fq.gz") for i in (1, 2)]
    out = {}
    for name, knob in (("device", {}), ("host", {"VGH_HMM_DEVICE": "0"}), ("bound", {"VGH_HMM_DEVICE_GIB": "0"}),
                       ("nomem", {"VGH_HMM_FAKE_NOMEM": "1", "VGH_HMM_EMIT_DEVICE": "0"}), ("host_emissions", {"VGH_HMM_EMIT_DEVICE": "0"})):
        w = tmp_path / name
        w.mkdir()
        (w / "samples.cfg").write_text("sample0 " + " ".join(fq) + "\n")
        env = dict(os.environ, VGH_RANDOM_DEVICE_VALUE="20241022", VGH_TIMING="1", **knob)
        r = _run([CLI, "genotype", "--load-graph", str(graph), "-s", "samples.cfg", "-t", "4", "--gpu", "0", "--granularity", "0.005"],
                 cwd=w, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = (gzip.open(w / "sample0.varigraph.vcf.gz", "rb").read(), r.stderr)
    assert out["device"][0] == out["host"][0] and out["device"][0].count(b"\n") > 50
    parts = [ln for ln in out["device"][1].split("\n") if "HMM part" in ln]
    assert len(parts) == 4, out["device"][1][-1500:]
    assert "HMM part" not in out["host"][1]
    # parts the device has no memory for (here: every second part, after the fact) go back to the host with the haplotypes
    # already drawn: same bytes
    assert out["nomem"][0] == out["host"][0] and "back on the host" in out["nomem"][1]
    # emission scores prepared by the host (round 2's path) and on the device (the default for this panel: all 15 haplotypes selected)
    assert out["host_emissions"][0] == out["host"][0] and "emissions on the device" not in out["host_emissions"][1]
    assert "emissions on the device" in out["device"][1]
    # a sample whose scores exceed the bound stays on the host
    assert out["bound"][0] == out["host"][0] and "HMM part" not in out["bound"][1]
