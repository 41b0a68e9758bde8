"""csrc/host/par_gunzip.cpp (one ordinary gzip stream decoded by several threads: guessed block starts, 16-bit symbols
with window markers, spans checked against each other) against the serial decoder, under AddressSanitizer + UBSan on the
CPU, on every stream shape of test_fast_inflate_cpu.py plus larger FASTQ-like and multi-member files.  The driver is
tests/native/par_gunzip_check.cpp."""
import gzip
import os
import random
import subprocess
import zlib

import numpy as np
import pytest

from test_fast_inflate_cpu import _gz, make_streams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "varigraph_amd", "csrc", "host")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pargz") / "par_gunzip_check")
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I", HOST,
           os.path.join(ROOT, "tests", "native", "par_gunzip_check.cpp"), os.path.join(HOST, "par_gunzip.cpp"),
           os.path.join(HOST, "fast_inflate.cpp"), "-lz", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return exe


def _fastq(n, seed):
    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    seqs = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, p=[.2499, .2499, .2499, .2499, .0004], size=(n, 150))]
    return b"".join(b"@A00123:45:HXXXXXXX:1:1101:%d:%d 1:N:0:ACGT\n%s\n+\n%s\n" % (1000 + i, 2000 + 7 * i, s.tobytes(),
                                                                                   bytes(rnd.choices(b"FFFFFF:,#", k=150)))
                    for i, s in enumerate(seqs))


def test_par_gunzip_matches_the_serial_decoder(driver, tmp_path):
    files = make_streams()
    fq = _fastq(40_000, 9)                                    # 13 MB of text: several default-size spans at level 1
    for lv in (1, 4, 9):
        files[f"big_l{lv}"] = _gz(fq, lv)
    files["big_multi"] = _gz(fq[:3_000_000], 4) + _gz(fq[3_000_000:3_000_100], 6) + _gz(b"") + _gz(fq[3_000_100:], 2)
    files["big_garbage"] = _gz(fq[:2_000_000], 4) + b"\0" * 5000 + os.urandom(3000)
    big = _gz(fq[:4_000_000], 4)
    rnd = random.Random(77)
    for i in range(12):
        b = bytearray(big)
        b[rnd.randrange(10, len(b))] ^= 1 << rnd.randrange(8)
        files[f"big_flip_{i}"] = bytes(b)
    for i, cut in enumerate([len(big) // 3, len(big) // 2 + 1, len(big) - 8, len(big) - 3]):
        files[f"big_trunc_{i}"] = big[:cut]
    # a stored block and a fixed-code block in the middle of dynamic ones: the seams around them cannot be guessed
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(fq[:1_000_000]) + c.flush(zlib.Z_FULL_FLUSH)
    c0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    rest = fq[1_000_000:1_300_000]
    body2 = c0.compress(rest) + c0.flush(zlib.Z_FULL_FLUSH)
    cf = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    body3 = cf.compress(fq[1_300_000:1_600_000]) + cf.flush(zlib.Z_FULL_FLUSH)
    c2 = zlib.compressobj(6, zlib.DEFLATED, -15)
    body4 = c2.compress(fq[1_600_000:2_500_000]) + c2.flush()
    whole = fq[:2_500_000]
    import struct
    files["mixed_blocks"] = b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + body + body2 + body3 + body4 + struct.pack("<II", zlib.crc32(whole), len(whole))
    paths = []
    for name, payload in files.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(payload)
        paths.append(str(p))
    r = subprocess.run([driver] + paths, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    n_checked = sum(len(v) >= 18 and v[:2] == b"\x1f\x8b" for v in files.values())
    assert f"{n_checked} files, 0 mismatches" in r.stdout
    assert zlib.decompress(files["mixed_blocks"], 31) == whole   # the hand-made stream is a valid one
