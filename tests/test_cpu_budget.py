"""vgh::CpuBudget, the one budget of running threads the Genotypers of a `varigraph-mi genotype` run share (csrc/host/genotyper.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "varigraph_amd")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from varigraph_amd import build
    build.build_all()
    exe = str(tmp_path_factory.mktemp("cb") / "cpu_budget_check")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-I", os.path.join(PKG, "csrc", "host"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "native", "cpu_budget_check.cpp"), "-L", PKG, "-lvghost", "-lvgmi", f"-Wl,-rpath,{PKG}", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-600:]
    return exe


@pytest.mark.parametrize("limit,threads", [(1, 8), (3, 16), (10, 40)])
def test_never_more_holders_than_tokens_and_nobody_starves(driver, limit, threads):
    r = subprocess.run([driver, str(limit), str(threads)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-300:]
    limited, unlimited, done = map(int, r.stdout.split())
    assert 1 <= limited <= limit
    assert unlimited > limit          # set(0) lifts the limit (the sleeps overlap)
    assert done == threads
