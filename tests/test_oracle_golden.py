"""CPU tests: the oracle (oracle/vg_oracle.c) against the golden vectors dumped from the real
reference (tests/golden/make_golden.py), i.e. the pinning of the checker itself."""
import json
import os

import numpy as np
import pytest

import oracle_lib as o
from conftest import GOLDEN, get_cohort

KATS = json.load(open(os.path.join(GOLDEN, "kats.json")))
BLOOM = json.load(open(os.path.join(GOLDEN, "bloom.json")))


def test_hash64_kats():
    for case in KATS["hash64"]:
        for a, b in zip(case["in"], case["out"]):
            assert o.hash64(int(a, 16), case["k"]) == int(b, 16)


@pytest.mark.parametrize("case", KATS["sketch"], ids=lambda c: f"k{c['k']}")
def test_sketch_traces(case):
    k = case["k"]
    n_nonempty = 0
    for tr in case["traces"]:
        seq = bytes.fromhex(tr["seq_hex"])
        got = o.sketch(seq, k)
        want = np.array([int(x, 16) for x in tr["keys"]], dtype=np.uint64)
        assert got is not None
        assert np.array_equal(got, want), (k, seq)
        n_nonempty += len(want) > 0
    assert n_nonempty > 5


def test_sketch_asserts_like_reference():
    assert o.sketch(b"", 27) is None          # assert(len > 0)
    assert o.sketch(b"ACGT", 29) is None      # assert(k <= 28)
    assert o.sketch(b"ACGT", 0) is None


def test_even_k_differs_from_simple_model():
    """The palindrome / stale-register rules are observable for even k:
    the fixture must contain traces where 'emit iff last k bases valid' is wrong."""
    case = [c for c in KATS["sketch"] if c["k"] == 6][0]
    diff = 0
    for tr in case["traces"]:
        seq = bytes.fromhex(tr["seq_hex"])
        run = 0
        simple = 0
        for ch in seq:
            run = run + 1 if o.lib().vgo_hash64 and (ch in b"ACGTUacgtu\x00\x01\x02\x03") else 0
            simple += run >= 6
        diff += simple != len(tr["keys"])
    assert diff > 0


def test_bloom_sizes():
    for c in KATS["bloom_size"]:
        m = o.lib().vgo_bloom_size(c["n"], c["p"])
        assert m == c["m"]
        assert o.lib().vgo_bloom_num_hashes(c["n"], m) == c["n_hash"]


def test_murmur_kats():
    for c in KATS["murmur"]:
        assert o.lib().vgo_murmur_sum(int(c["key"], 16), int(c["seed"], 16)) == int(c["sum"], 16)


@pytest.mark.parametrize("case", BLOOM, ids=lambda c: c["name"])
def test_bloom_filter(case):
    want = np.load(os.path.join(GOLDEN, f"bloom_{case['name']}_filter.npy"))
    seeds = np.array([int(s, 16) for s in case["seeds"]], dtype=np.uint64)
    filt = np.zeros(case["m"], dtype=np.uint8)
    for s in case["seqs"]:
        assert o.bloom_add_seq(filt, seeds, np.frombuffer(s.encode(), dtype=np.uint8), case["k"]) >= 0
    assert np.array_equal(filt, want)
    keys = [int(x, 16) for x in case["query_keys"]]
    mn, nz = o.bloom_query(filt, seeds, keys)
    assert mn.tolist() == case["query_count"]
    assert nz.tolist() == case["query_find"]


def test_cohort_counts(cohort):
    """oracle counting == reference FastqKmer::build_fastq_index on the same reads."""
    g = cohort.graph
    t = o.Table(g.keys)
    hits, rb = t.count_block(cohort.block(), cohort.k)
    assert hits >= 0
    assert rb == cohort.ref_read_base == int(cohort.meta["read_base"])
    assert np.array_equal(t.counts(), cohort.ref_c_in_graph_order())
    assert np.array_equal(np.sort(g.keys), np.sort(cohort.ref_keys))


def test_cohort_hist_and_peak(cohort):
    g = cohort.graph
    m = cohort.meta
    c = cohort.ref_c_in_graph_order()
    hist = o.hom_hist(c, g.f, g.bitvec, g.hap_num, g.vcf_ploidy)
    assert hist.tolist() == m["hist"]
    depth = o.lib().vgo_read_depth(cohort.ref_read_base, g.genome_size)
    assert np.float32(depth).view(np.uint32) == int(m["read_depth_bits"], 16)
    mx, hm = o.hom_peak(hist, depth)
    assert (mx, hm) == (int(m["max_cov"]), int(m["hom_cov"]))
    hap = o.lib().vgo_hap_kmer_cov(hm, m["sample_ploidy"], depth)
    assert np.float32(hap).view(np.uint32) == int(m["hap_kmer_cov_bits"], 16)
    hm2 = o.lib().vgo_use_depth_cov(depth)
    assert hm2 == int(m["use_depth_hom_cov"])
    hap2 = o.lib().vgo_hap_kmer_cov(hm2, m["sample_ploidy"], depth)
    assert np.float32(hap2).view(np.uint32) == int(m["use_depth_hap_kmer_cov_bits"], 16)


def test_genome_size_matches_reference(cohort):
    assert cohort.graph.genome_size == cohort.ref_genome_size
