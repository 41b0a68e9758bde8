"""csrc/host/fast_inflate.cpp (the gunzip of the ingest path) against zlib, under AddressSanitizer + UBSan on the CPU:
every block type, header field, flush pattern, concatenated members, trailing garbage, truncation at many offsets and
random bit flips.  The driver is tests/native/inflate_check.cpp."""
import os
import random
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "varigraph_amd", "csrc", "host")


def _gz(d, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, fields=False):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = c.compress(d) + c.flush()
    h = b"\x1f\x8b\x08" + bytes([0x1E if fields else 0]) + b"\0\0\0\0\0\xff"
    if fields:
        h += struct.pack("<H", 7) + b"ab\x03\x00xyz" + b"file.fq\0" + b"a comment\0"
        h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    return h + body + struct.pack("<II", zlib.crc32(d), len(d) & 0xFFFFFFFF)


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("inflate") / "inflate_check")
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I", HOST,
           os.path.join(ROOT, "tests", "native", "inflate_check.cpp"), os.path.join(HOST, "fast_inflate.cpp"), "-lz", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return exe


def make_streams():
    """name -> gzip bytes: every block type, header field, flush pattern, concatenated members, trailing garbage, truncation at many
    offsets and random bit flips"""
    rnd = random.Random(5)
    rng = np.random.default_rng(3)
    seqs = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, p=[.2499, .2499, .2499, .2499, .0004], size=(3000, 150))]
    fq = b"".join(b"@read_%d/1 extra\n%s\n+\n%s\n" % (i, s.tobytes(), bytes(rnd.choices(b"FFFF:,#", k=150))) for i, s in enumerate(seqs))
    binary = rng.integers(0, 256, size=200_000, dtype=np.uint8).tobytes()
    runs = b"".join(bytes([65 + rnd.randrange(4)]) * rnd.randrange(1, 40) for _ in range(8000))
    periodic = b"abcdefg" * 20000 + b"xy" * 30000 + b"q" * 70000 + bytes(range(256)) * 200 + b"0123456789abc" * 5000
    files = {}
    for name, d in (("fq", fq), ("bin", binary), ("runs", runs), ("per", periodic), ("empty", b""), ("one", b"A"), ("tiny", b"hello hello hello")):
        for lv in (0, 1, 4, 6, 9):
            files[f"{name}_l{lv}"] = _gz(d, lv)
        files[f"{name}_fixed"] = _gz(d, 6, zlib.Z_FIXED)
        files[f"{name}_huff"] = _gz(d, 6, zlib.Z_HUFFMAN_ONLY)
        files[f"{name}_rle"] = _gz(d, 6, zlib.Z_RLE)
        files[f"{name}_fields"] = _gz(d, 6, fields=True)
    files["multi"] = _gz(fq[:100000]) + _gz(b"") + _gz(fq[100000:200000], 1) + _gz(binary[:5000], 0)
    files["garbage"] = _gz(fq[:50000]) + b"trailing garbage bytes"
    files["garbage1"] = _gz(fq[:50000]) + b"\x1f"
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    step, body = 777, b""
    flushed = fq[: (200000 // step) * step]
    for i in range(0, len(flushed), step):
        body += c.compress(flushed[i:i + step]) + c.flush(zlib.Z_SYNC_FLUSH if (i // step) % 3 else zlib.Z_FULL_FLUSH)
    files["flushy"] = b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + body + c.flush() + struct.pack("<II", zlib.crc32(flushed), len(flushed))
    base = _gz(fq[:300000], 6)
    for i, cut in enumerate([1, 2, 5, 9, 10, 11, 15, 40, 100, 1000, 5000, len(base) // 2, len(base) - 9, len(base) - 8, len(base) - 4, len(base) - 1]):
        files[f"trunc_{i}"] = base[:cut]
    stored = _gz(binary[:70000], 0)
    for i, cut in enumerate([12, 14, 15, 16, 100, 65000, 65550, len(stored) - 3]):
        files[f"trunc_stored_{i}"] = stored[:cut]
    fixed = _gz(fq[:3000], 6, zlib.Z_FIXED)
    for cut in range(11, len(fixed), 53):
        files[f"trunc_fixed_{cut}"] = fixed[:cut]
    for i in range(120):
        b = bytearray(base)
        b[rnd.randrange(10, len(b))] ^= 1 << rnd.randrange(8)
        files[f"flip_{i}"] = bytes(b)
    for name, off in (("badcrc", -6), ("badisize", -2)):
        b = bytearray(base)
        b[off] ^= 0x55
        files[name] = bytes(b)
    return files


def test_fast_inflate_matches_zlib_on_every_stream_shape(driver, tmp_path):
    files = make_streams()
    paths = []
    for name, payload in files.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(payload)
        paths.append(str(p))
    r = subprocess.run([driver] + paths, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    n_checked = sum(len(v) >= 2 for v in files.values())   # the driver skips what has no gzip magic
    assert f"{n_checked} files, 0 mismatches" in r.stdout


def test_crc32_fast_matches_zlib():
    """the carry-less-multiply CRC behind the decoder's member check (exported by libvghost for this test)"""
    import ctypes as C
    from varigraph_amd import host
    l = host.lib()
    l.vgh_crc32.restype = C.c_uint32
    l.vgh_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    for n in (0, 1, 15, 16, 63, 64, 65, 79, 80, 127, 128, 1000, 4096 + 13, 1 << 20):
        for start in (0, 1, 7):
            buf = np.ascontiguousarray(data[start:start + n])
            seed = 0 if n % 2 else 0x1234ABCD
            assert l.vgh_crc32(seed, buf.ctypes.data, buf.size) == zlib.crc32(buf.tobytes(), seed)
