"""The byte-source combinators behind the hand-over from the device-side FASTQ parser / inflate to the host reader
(csrc/host/byte_source.hpp: skip, concat, from_memory, open_at), under AddressSanitizer + UBSan, on plain, gzip and
block-gzip copies of one FASTQ text."""
import gzip
import os
import subprocess

import pytest

from varigraph_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "varigraph_amd", "csrc", "host")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bs") / "byte_source_check")
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I", HOST,
           os.path.join(ROOT, "tests", "native", "byte_source_check.cpp"), os.path.join(HOST, "byte_source.cpp"),
           os.path.join(HOST, "fast_inflate.cpp"), os.path.join(HOST, "par_gunzip.cpp"), os.path.join(HOST, "fastx_reader.cpp"), "-lz", "-lpthread", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return exe


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("bsf")
    reads = [bytes(b"ACGT"[(i * 7 + j * 3 + i * j) % 4] for j in range(40 + i % 90)) for i in range(4000)]
    text = b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads))
    plain = d / "t.fq"
    plain.write_bytes(text)
    gz = d / "t.fq.gz"
    with gzip.open(gz, "wb", compresslevel=4) as f:
        f.write(text)
    bgz = d / "t.bgz.gz"
    synth.bgzf_compress_file(str(plain), str(bgz), level=5, block=20000)
    return {"text": text, "reads": reads, "plain": str(plain), "gzip": str(gz), "bgzf": str(bgz)}


def _run(driver, *args):
    r = subprocess.run([driver, *map(str, args)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    return r.stdout


@pytest.mark.parametrize("kind", ["plain", "gzip", "bgzf"])
def test_skip_glue_and_reader_from_an_offset(kind, driver, files):
    text, path = files["text"], files[kind]
    assert _run(driver, path, "cat") == text
    for n in (0, 1, 4095, 65536, 100001, len(text) - 1, len(text), len(text) + 5):
        assert _run(driver, path, "skip", n) == text[n:]
    for n, m in ((0, 0), (10, 10), (70000, 70000), (123, 456)):
        assert _run(driver, path, "glue", n, m) == text[:n] + text[m:]
    # the host reader taking a stream over at a record boundary yields exactly the remaining records
    off = len(b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(files["reads"][:1234])))
    assert _run(driver, path, "seqs", off) == b"".join(r + b"\n" for r in files["reads"][1234:])


def test_open_at_member_boundaries(driver, files):
    text = files["text"]
    # bytes that are not a gzip header are text only at the START of a file (gzopen's transparent mode); behind a compressed
    # stream they are what gzread and the host block-gzip reader ignore (trailing garbage), so a take-over at an offset yields nothing
    assert _run(driver, files["plain"], "at", 5000) == b""
    assert _run(driver, files["plain"], "at", 0) == text
    raw = open(files["bgzf"], "rb").read()
    offs, pos = [], 0
    while pos < len(raw):
        offs.append(pos)
        pos += (raw[pos + 16] | raw[pos + 17] << 8) + 1
    for k in (0, 1, 7, len(offs) - 2):
        assert _run(driver, files["bgzf"], "at", offs[k]) == text[20000 * k:]
