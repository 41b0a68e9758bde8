import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the log of every run names its slowest tests (the GPU suite has a wall-clock limit: what grows shows up here first)
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 15


_LONG_REFERENCE_TESTS = ("test_more_than_128_genotypes_run_on_the_device_vcf_identical", "test_c3_chr20_scale_12m_pairs_use_depth_vcf_identical",
                         "test_c4_eight_samples_over_the_gpus_present_equal_single_sample_reference_runs")


def pytest_collection_modifyitems(config, items):
    """The tests that wait for the long all-CPU reference runs go to the END of the session (in the order their runs finish): the runs
    start with the session (pytest_collection_finish below) and are done by the time the rest of the suite is."""
    def rank(it):
        name = it.name.split("[")[0]
        return _LONG_REFERENCE_TESTS.index(name) + 1 if name in _LONG_REFERENCE_TESTS else 0
    items.sort(key=rank)      # stable: everything else keeps its order


def pytest_collection_finish(session):
    """The GPU suite's two long all-CPU reference runs (tests/test_gpu_configs.py: C3, C4) start now, on a background thread, when
    their tests are among the selected ones: they then run beside the rest of the suite instead of inside its wall clock."""
    names = {it.name.split("[")[0] for it in session.items}
    which = [n for n, t in (("n20", "test_more_than_128_genotypes_run_on_the_device_vcf_identical"),
                            ("c3", "test_c3_chr20_scale_12m_pairs_use_depth_vcf_identical"),
                            ("c4", "test_c4_eight_samples_over_the_gpus_present_equal_single_sample_reference_runs")) if t in names]
    if not which or session.config.option.collectonly or os.environ.get("VG_TEST_NO_EARLY"):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    from varigraph_amd import build
    build.build_vgmi()
    build.build_host()
    build.build_cli()
    import test_gpu_configs
    if all(os.path.exists(b) for b in (test_gpu_configs.CLI, test_gpu_configs.REF)):
        test_gpu_configs.start_early(which)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Which build of the checker ran: the byte-exact VCF / graph.bin comparisons run the prebuilt, git-ignored binaries of
    oracle/_ref (the unmodified reference compiled by oracle/Makefile) -- their sha256 goes into the log of every run."""
    import hashlib
    lines = []
    for rel in ("oracle/_ref/varigraph_det", "oracle/_ref/varigraph_ref", "oracle/_ref/ref_harness_det", "oracle/_ref/ref_harness", "oracle/_ref/varigraph_hip", "oracle/liboracle.so",
                "varigraph_amd/libvgmi.so", "varigraph_amd/libvghost.so", "varigraph_amd/libvgsynth.so", "varigraph_amd/bin/varigraph-mi"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            h = hashlib.sha256()
            with open(path, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 20), b""):
                    h.update(chunk)
            lines.append(f"sha256 {h.hexdigest()}  {rel}")
        else:
            lines.append(f"absent                                                                   {rel}")
    terminalreporter.write_sep("=", "binaries this run used")      # (a summary section: printed under -q too, which hides the report header)
    for ln in lines:
        terminalreporter.write_line(ln)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native libraries exist (hipcc cross-compiles without a GPU)."""
    from varigraph_amd import build
    build.build_vgmi()
    build.build_host()
    build.build_cli()
    build.build_oracle(with_ref=False)


def read_fastq_seqs(path):
    """Sequence lines of a FASTQ(.gz) file as a list of bytes (4-line records)."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        lines = f.read().split(b"\n")
    return lines[1::4][: len(lines) // 4]


def block_from_seqs(seqs):
    return np.frombuffer(b"".join(s + b"\n" for s in seqs), dtype=np.uint8).copy()


class Cohort:
    def __init__(self, name):
        import graphbin_py
        self.dir = os.path.join(GOLDEN, name)
        self.meta = json.load(open(os.path.join(self.dir, "meta.json")))
        self.k = int(self.meta["k"])
        self.graph = graphbin_py.load(os.path.join(self.dir, "graph.bin.gz"))
        rb, gs, keys, c, f = graphbin_py.load_counts_dump(os.path.join(self.dir, "counts.bin.gz"))
        self.ref_read_base, self.ref_genome_size = rb, gs
        self.ref_keys, self.ref_c, self.ref_f = keys, c, f
        self.ref_nodes = graphbin_py.load_nodes_dump_gz(os.path.join(self.dir, "nodes.bin.gz"))
        self._block = None

    @property
    def n_reads(self):
        return 2 * int(self.meta["n_pairs"])

    def block(self):
        """The sample's read block: FASTQ files if stored (file 1 then file 2, as the reference reads
        them), else regenerated by the seeded generator (interleaved; order is irrelevant to counts)."""
        if self._block is None:
            fq = [os.path.join(self.dir, f"reads_{m}.fq.gz") for m in (1, 2)]
            if all(os.path.exists(p) for p in fq):
                seqs = read_fastq_seqs(fq[0]) + read_fastq_seqs(fq[1])
                self._block = block_from_seqs(seqs)
            else:
                self._block = self.regenerate_block()
        return self._block

    def haplotypes(self):
        from varigraph_amd import synth
        m = self.meta
        ref = synth.make_reference(m["ref_len"], seed=m["ref_seed"])
        variants, gts = synth.make_cohort(ref, m["n_var"], n_samples=m["n_samples"], ploidy=m["ploidy"],
                                          seed=m["cohort_seed"], indel_frac=m["indel_frac"], sv_frac=m["sv_frac"])
        return synth.sample_haplotypes(ref, variants, gts, 0, m["ploidy"])

    def regenerate_block(self):
        from varigraph_amd import vgmi
        import hashlib
        m = self.meta
        b = vgmi.synth_reads_host(m["read_seed"], 0, self.n_reads, m["read_len"], self.haplotypes())
        assert hashlib.md5(b.tobytes()).hexdigest() == m["block_md5"], "synthetic generator drifted"
        return b

    def ref_c_in_graph_order(self):
        """Reference counters re-ordered to graph.bin key order."""
        order = np.argsort(self.ref_keys)
        pos = np.searchsorted(self.ref_keys[order], self.graph.keys)
        assert np.array_equal(self.ref_keys[order][pos], self.graph.keys)
        return self.ref_c[order][pos]


_cohorts = {}


def get_cohort(name):
    if name not in _cohorts:
        _cohorts[name] = Cohort(name)
    return _cohorts[name]


COHORTS = ["cohort_snp", "cohort_sv", "cohort_k22", "cohort_tetra", "c1"]


@pytest.fixture(scope="session", params=COHORTS)
def cohort(request):
    return get_cohort(request.param)
