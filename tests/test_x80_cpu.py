"""csrc/vg_x80.h -- the reference's `long double` arithmetic (x87 extended precision) restated in integer operations for the
device -- against the x87 unit itself: products, sums, quotients and 120-term chains over random and edge operands
(tests/native/x80_check.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_x80_arithmetic_is_the_x87_units(tmp_path):
    exe = str(tmp_path / "x80_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "varigraph_amd", "csrc"),
                        os.path.join(ROOT, "tests", "native", "x80_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "4000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ", 0 mismatches" in r.stdout, r.stdout[-2000:]
