"""The context table's entry algebra (varigraph_amd/csrc/vgmi_ctable.h: what count27c_kernel and the generic kernels' tail lookups
compute per read position) on the host: tests/native/ctable_model.cpp builds the table for seeded key sets -- SNPs near and far,
short deletions, homopolymers, 16-mers that are their own reverse complement, tandem repeats, a repeat-rich reference that fills the
exact overflow table -- walks reads of both strands with errors, N and lower case through the grid, and holds every counter against
a brute-force dictionary count of every 27-mer; the single-k-mer lookup (ct_find) is held against the dictionary on every k-mer.
Round 5: the same for k = 19 .. 25 (flanks of k - 16 bases, a 16-mer looked up every 6 or 4 bases for the windows that end there)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ctable") / "ctable_model")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "varigraph_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "ctable_model.cpp"), "-o", out], check=True)
    return out


@pytest.mark.parametrize("args", [(1, 20000, 400, 2000, 40), (2, 50000, 2000, 5000, 40), (3, 30000, 1500, 3000, 90),
                                  (4, 30000, 300, 3000, 40, 100), (5, 240000, 2000, 3000, 10), (6, 6000, 60, 500, 95)],
                         ids=["dense", "very-dense", "crowded", "repeats", "sparse-table", "tiny-crowded"])
def test_context_table_model_counts_equal_dictionary(exe, args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    d = json.loads(r.stdout)
    assert d["bad"] == 0 and d["hits"] > 1000 and d["entries"] == d["keys"] + 11 * d["unitigs"]
    if len(args) > 5 or args[4] >= 90:
        assert d["over_kmers"] > 0          # the overflow trail was walked


@pytest.mark.parametrize("k", [19, 21, 23, 25, 20, 22, 24, 26, 28])      # (even k: no window bit for a k-mer that is its own reverse complement; 28, round 6: flanks of 11, eleven windows an entry)
@pytest.mark.parametrize("args", [(1, 20000, 400, 2000, 40, 0), (3, 30000, 1500, 3000, 90, 0), (4, 30000, 300, 3000, 40, 100), (6, 6000, 60, 500, 95, 0)],
                         ids=["dense", "crowded", "repeats", "tiny-crowded"])
def test_context_table_model_other_odd_k(exe, args, k):
    r = subprocess.run([exe] + [str(a) for a in args] + [str(k)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    d = json.loads(r.stdout)
    assert d["bad"] == 0 and d["hits"] > 1000 and d["entries"] == d["keys"] + (10 if k == 28 else k - 16) * d["unitigs"]
    assert d["over_kmers"] > 0
